#!/bin/bash
# round 5, lease 16: diagnostic 2 of the sharing race
O=gpurun_out/r05r; mkdir -p $O
timeout 500 python tools/diag/share_pipeline2.py > $O/diag2.log 2>&1; echo "diag rc=$?"; grep -v "^\[\|amdgpu.ids" $O/diag2.log | tail -30

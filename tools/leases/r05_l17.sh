#!/bin/bash
O=gpurun_out/r05s; mkdir -p $O
timeout 300 python tools/diag/share_forward.py > $O/diag3.log 2>&1; echo "diag rc=$?"; grep -v "^\[\|amdgpu.ids" $O/diag3.log | tail -40 | cut -c1-1500

#!/bin/bash
O=gpurun_out/r05w; mkdir -p $O
timeout 280 python tools/diag/share_pipeline3.py second > $O/diag5.log 2>&1; echo "diag rc=$?"; grep -v "^\[\|amdgpu.ids" $O/diag5.log | tail -12 | cut -c1-600

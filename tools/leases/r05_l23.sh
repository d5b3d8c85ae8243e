#!/bin/bash
O=gpurun_out/r05y; mkdir -p $O
timeout 95 python tools/diag/share_edit_localize.py attribute > $O/diag7.log 2>&1; echo "diag rc=$?"; grep -v "^\[\|amdgpu.ids" $O/diag7.log | tail -12 | cut -c1-400

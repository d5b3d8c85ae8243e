#!/bin/bash
# Round 4, lease 11: the whole -m gpu suite on the current defaults (split-bf16 GEMMs + attention, codec on split-bf16, drain widening)
O=gpurun_out/r04k; mkdir -p $O
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q -x > $O/gpu_suite.log 2>&1; echo "suite rc=$? $(( $(date +%s) - t0 )) s"
tail -45 $O/gpu_suite.log

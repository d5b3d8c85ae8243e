#!/bin/bash
# Round 4, lease 13 / 14 (second run after the tolerance and acceptance-test changes): MX-FP8 experiment after the operand-layout fix (measured with tools/f8_probe.cpp)
O=gpurun_out/r04n; mkdir -p $O
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 400 python -m pytest -m gpu -q -s tests/test_gpu_fp8_experiment.py -k acceptance > $O/tests_fp8.log 2>&1; echo "fp8 tests rc=$? $(date +%T)"
grep -E "\[fp8|passed|failed|Error|^E  " $O/tests_fp8.log | tail -24

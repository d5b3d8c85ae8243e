#!/bin/bash
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_gpu_stable_audio.py -m gpu -q > gpurun_out/sa5_tests.log 2>&1; echo "tests rc=$?"; tail -30 gpurun_out/sa5_tests.log

#!/bin/bash
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
timeout 170 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_loops.py -m gpu -q -x -k "hooks or clip_edit_end or smoke or plan_cache or replay or ddpm_inversion or two_prompt or step_method" > gpurun_out/sanity_tests.log 2>&1; echo "tests rc=$?"; tail -22 gpurun_out/sanity_tests.log

#!/bin/bash
export PYTHONPATH=. TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_unet.py -m gpu -x -q > gpurun_out/c4_tests.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/c4_tests.log
timeout 600 python tools/unet_profile.py 2 > gpurun_out/c4_prof_B2.log 2>&1; echo "prof2 rc=$?"
grep "^\[" gpurun_out/c4_prof_B2.log
timeout 600 python tools/unet_profile.py 40 > gpurun_out/c4_prof_B40.log 2>&1; echo "prof40 rc=$?"
grep "^\[" gpurun_out/c4_prof_B40.log
AED_TILE_OVERRIDE="" timeout 900 python tools/tile_sweep.py 2 > gpurun_out/c4_sweep_B2.log 2>&1; echo "sweep2 rc=$?"
tail -1 gpurun_out/c4_sweep_B2.log
timeout 900 python tools/tile_sweep.py 40 > gpurun_out/c4_sweep_B40.log 2>&1; echo "sweep40 rc=$?"
tail -1 gpurun_out/c4_sweep_B40.log

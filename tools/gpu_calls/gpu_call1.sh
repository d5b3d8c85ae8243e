#!/bin/bash
# GPU call 1 of round 2: correctness of the new kernels, then A/B profiles and the tile sweep.
export PYTHONPATH=. TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q > gpurun_out/c1_kernels.log 2>&1; echo "kernels rc=$?" | tee -a gpurun_out/c1_summary.log
tail -5 gpurun_out/c1_kernels.log
timeout 600 python -m pytest tests/test_gpu_unet.py -m gpu -x -q > gpurun_out/c1_unet.log 2>&1; echo "unet rc=$?" | tee -a gpurun_out/c1_summary.log
tail -5 gpurun_out/c1_unet.log
timeout 600 python tools/unet_profile.py 2 lin=0,geglu=0,two=0 "" attn=2 gn=1 attn=2,gn=1 > gpurun_out/c1_prof_B2.log 2>&1; echo "prof2 rc=$?" | tee -a gpurun_out/c1_summary.log
grep "^\[" gpurun_out/c1_prof_B2.log
timeout 600 python tools/unet_profile.py 40 lin=0,geglu=0,two=0 "" attn=2 > gpurun_out/c1_prof_B40.log 2>&1; echo "prof40 rc=$?" | tee -a gpurun_out/c1_summary.log
grep "^\[" gpurun_out/c1_prof_B40.log
timeout 900 python tools/tile_sweep.py 2 > gpurun_out/c1_sweep_B2.log 2>&1; echo "sweep2 rc=$?" | tee -a gpurun_out/c1_summary.log
tail -3 gpurun_out/c1_sweep_B2.log
timeout 900 python tools/tile_sweep.py 40 > gpurun_out/c1_sweep_B40.log 2>&1; echo "sweep40 rc=$?" | tee -a gpurun_out/c1_summary.log
tail -3 gpurun_out/c1_sweep_B40.log

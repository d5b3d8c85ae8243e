#!/bin/bash
export PYTHONPATH=. TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "lin_gemm or geglu or two_source" > gpurun_out/c2_kernels.log 2>&1; echo "kernels rc=$?"
tail -3 gpurun_out/c2_kernels.log
timeout 600 python tools/unet_profile.py 2 "" attn=2,gn=1 > gpurun_out/c2_prof_B2.log 2>&1; echo "prof2 rc=$?"
grep "^\[" gpurun_out/c2_prof_B2.log
timeout 900 python tools/tile_sweep.py 2 > gpurun_out/c2_sweep_B2.log 2>&1; echo "sweep2 rc=$?"
tail -1 gpurun_out/c2_sweep_B2.log

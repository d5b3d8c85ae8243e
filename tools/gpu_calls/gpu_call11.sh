#!/bin/bash
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "per_batch or lin_gemm_epilogues or fused_layernorm" > gpurun_out/c11_k.log 2>&1; echo "kernel tests rc=$?"; tail -3 gpurun_out/c11_k.log
timeout 900 python -m pytest tests/test_gpu_unet.py -m gpu -x -q > gpurun_out/c11_unet.log 2>&1; echo "unet tests rc=$?"; tail -3 gpurun_out/c11_unet.log; grep "^E " gpurun_out/c11_unet.log | head -5
timeout 600 python tools/unet_profile.py 2 "" fold=0 > gpurun_out/c11_prof_B2.log 2>&1; echo "prof2 rc=$?"; grep "^\[" gpurun_out/c11_prof_B2.log
timeout 900 python tools/tile_sweep.py 200 8 > gpurun_out/c11_sweep_B200.log 2>&1; echo "sweep200 rc=$?"; tail -1 gpurun_out/c11_sweep_B200.log

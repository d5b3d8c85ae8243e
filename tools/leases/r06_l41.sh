#!/bin/bash
# round 6, lease 41: GEGLU epilogue in simple-rows form: GEGLU / U-Net / split-bf16 / stable-audio (SwiGLU) / pipeline tests
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06ao; mkdir -p $O
timeout 1200 python -m pytest -q -m gpu -x tests/test_gpu_kernels.py tests/test_gpu_unet.py tests/test_gpu_zz_split_bf16.py tests/test_gpu_stable_audio.py tests/test_gpu_pipeline.py tests/test_gpu_coresidency.py > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log

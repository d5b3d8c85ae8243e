#!/bin/bash
# Stable Audio path: first GPU validation + register-resident GroupNorm A/B.
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_stable_audio.py -m gpu -q -x > gpurun_out/sa_tests.log 2>&1; echo "sa tests rc=$?"
tail -25 gpurun_out/sa_tests.log
timeout 400 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "groupnorm or geglu or misc or linear_epilogues or layernorm" > gpurun_out/sa_kern.log 2>&1; echo "kernel tests rc=$?"
tail -5 gpurun_out/sa_kern.log
timeout 300 python tools/unet_profile.py 2 "" gnreg=0 > gpurun_out/sa_gnreg.log 2>&1; echo "gnreg rc=$?"
grep -E "^\[|gn1 " gpurun_out/sa_gnreg.log
timeout 500 python tools/sa_profile.py 2 > gpurun_out/sa_prof_B2.log 2>&1; echo "sa prof rc=$?"
head -30 gpurun_out/sa_prof_B2.log

#!/bin/bash
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "attention" > gpurun_out/c7_attn.log 2>&1; echo "attn tests rc=$?"
tail -4 gpurun_out/c7_attn.log
timeout 900 python -m pytest "tests/test_gpu_pc.py::test_pc_clis_extract_pt_apply_on_the_gpu" "tests/test_gpu_unet.py::test_full_audioldm2_unet_matches_oracle" "tests/test_gpu_unet.py::test_tiny_unet_matches_oracle" -m gpu -x -q > gpurun_out/c7_tests.log 2>&1; echo "tests rc=$?"
tail -4 gpurun_out/c7_tests.log
timeout 600 python tools/unet_profile.py 2 "" attn=2 > gpurun_out/c7_prof_B2.log 2>&1; echo "prof2 rc=$?"
grep "^\[" gpurun_out/c7_prof_B2.log
timeout 600 python tools/unet_profile.py 40 "" attn=2 > gpurun_out/c7_prof_B40.log 2>&1; echo "prof40 rc=$?"
grep "^\[" gpurun_out/c7_prof_B40.log
timeout 600 python tools/unet_profile.py 80 "" > gpurun_out/c7_prof_B80.log 2>&1; echo "prof80 rc=$?"
grep "^\[" gpurun_out/c7_prof_B80.log

#!/bin/bash
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "geglu or groupnorm" > gpurun_out/c8_k.log 2>&1; echo "kernel tests rc=$?"; tail -3 gpurun_out/c8_k.log
timeout 900 python -m pytest "tests/test_gpu_pc.py::test_pc_clis_extract_pt_apply_on_the_gpu" -m gpu -x -q > gpurun_out/c8_pc.log 2>&1; echo "pc rc=$?"; tail -3 gpurun_out/c8_pc.log; grep "^E " gpurun_out/c8_pc.log | head -5
timeout 600 python tools/unet_profile.py 2 "" gn=1 > gpurun_out/c8_prof_B2.log 2>&1; echo "prof2 rc=$?"; grep "^\[" gpurun_out/c8_prof_B2.log
timeout 600 python tools/unet_profile.py 100 "" > gpurun_out/c8_prof_B100.log 2>&1; echo "prof100 rc=$?"; grep "^\[" gpurun_out/c8_prof_B100.log
timeout 600 python tools/unet_profile.py 200 "" > gpurun_out/c8_prof_B200.log 2>&1; echo "prof200 rc=$?"; grep "^\[" gpurun_out/c8_prof_B200.log
AED_TILE_OVERRIDE="" timeout 900 python tools/tile_sweep.py 2 > gpurun_out/c8_sweep_B2.log 2>&1; echo "sweep2 rc=$?"; tail -1 gpurun_out/c8_sweep_B2.log
timeout 900 python tools/tile_sweep.py 80 > gpurun_out/c8_sweep_B80.log 2>&1; echo "sweep80 rc=$?"; tail -1 gpurun_out/c8_sweep_B80.log

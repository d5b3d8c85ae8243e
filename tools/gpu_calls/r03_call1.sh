#!/bin/bash
# r03 call 1: CU-partition premise test + mid-batch (config 3 per-rank shape) profiles and tile sweeps.
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/cu_partition.py 100 > gpurun_out/r03_cu_partition.log 2>&1; echo "partition rc=$?"
tail -40 gpurun_out/r03_cu_partition.log
timeout 200 python tools/unet_profile.py 16 "" > gpurun_out/r03_prof_B16.log 2>&1; echo "prof16 rc=$?"; head -30 gpurun_out/r03_prof_B16.log
timeout 300 python tools/tile_sweep.py 16 32 > gpurun_out/r03_sweep_B16.log 2>&1; echo "sweep16 rc=$?"; tail -3 gpurun_out/r03_sweep_B16.log
timeout 400 python tools/tile_sweep.py 192 8 > gpurun_out/r03_sweep_B192.log 2>&1; echo "sweep192 rc=$?"; tail -3 gpurun_out/r03_sweep_B192.log

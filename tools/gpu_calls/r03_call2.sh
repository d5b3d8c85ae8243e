#!/bin/bash
# r03 call 2: per-op profile under CU masks, several edit lanes per partition, masked tile sweep (batch 2 on 128 CUs).
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/cu_partition2.py 100 > gpurun_out/r03_cu_partition2.log 2>&1; echo "partition2 rc=$?"
tail -75 gpurun_out/r03_cu_partition2.log
AED_SWEEP_CUS=128 timeout 400 python tools/tile_sweep.py 2 60 > gpurun_out/r03_sweep_B2_cus128.log 2>&1; echo "sweep2@128 rc=$?"; tail -2 gpurun_out/r03_sweep_B2_cus128.log
AED_SWEEP_CUS=64 timeout 400 python tools/tile_sweep.py 2 60 > gpurun_out/r03_sweep_B2_cus64.log 2>&1; echo "sweep2@64 rc=$?"; tail -2 gpurun_out/r03_sweep_B2_cus64.log

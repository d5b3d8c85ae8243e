#!/bin/bash
# Tile sweep of the Stable Audio DiT's contractions (batch 2 and 40), table generation, re-profile, re-test.
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
timeout 150 python tools/tile_sweep.py 2 60 dit > gpurun_out/sa3_sweep_B2.log 2>&1; echo "sweep B2 rc=$?"
timeout 150 python tools/tile_sweep.py 40 6 dit > gpurun_out/sa3_sweep_B40.log 2>&1; echo "sweep B40 rc=$?"
grep -v amdgpu gpurun_out/sa3_sweep_B2.log | cut -c1-150; grep -v amdgpu gpurun_out/sa3_sweep_B40.log | cut -c1-150
python tools/tile_table_from_sweep.py --out audioeditingcode_amd/tile_table_dit.py gpurun_out/tile_sweep_dit_B2.json gpurun_out/tile_sweep_dit_B40.json
cp audioeditingcode_amd/tile_table_dit.py gpurun_out/tile_table_dit.py
timeout 120 python tools/sa_profile.py 2 > gpurun_out/sa3_prof_B2.log 2>&1; echo "prof rc=$?"; sed -n 2,12p gpurun_out/sa3_prof_B2.log
timeout 120 python -m pytest tests/test_gpu_stable_audio.py -m gpu -q -k "dit_forward or swiglu or device_loops" > gpurun_out/sa3_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/sa3_tests.log

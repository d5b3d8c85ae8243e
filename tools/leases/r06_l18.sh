#!/bin/bash
# round 6, lease 18: chain test again + the files after it, then lease 17's no-split diagnostic
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06r; mkdir -p $O
( time timeout 1500 python -m pytest -x -q -m gpu "tests/test_gpu_pc.py::test_config4_three_consecutive_drift_timesteps_at_full_size" tests/test_gpu_pipeline.py tests/test_gpu_stable_audio.py tests/test_gpu_unet.py tests/test_gpu_zz_split_bf16.py tests/test_gpu_zzz_fullsize_oracle_fixture.py -s ) > $O/tests.log 2>&1; echo "tests rc=$?"; grep -h "config 4 chain\|drifted trajectory\|config 5 at T=200\|passed\|failed\|^real\|Error" $O/tests.log | cut -c1-1500; tail -22 $O/tests.log | grep "s call"
bash tools/leases/r06_l17.sh

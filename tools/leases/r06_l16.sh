#!/bin/bash
# round 6, lease 16: the rest of the GPU suite after the regenerated chain fixture (pc, pipeline, stable audio incl. T=200, unet, zz, zzz),
# then the per-family fp8 budget of the Stable Audio DiT at T=200
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06p; mkdir -p $O
( time timeout 1500 python -m pytest -x -q -m gpu tests/test_gpu_pc.py tests/test_gpu_pipeline.py tests/test_gpu_stable_audio.py tests/test_gpu_unet.py tests/test_gpu_zz_split_bf16.py tests/test_gpu_zzz_fullsize_oracle_fixture.py -s ) > $O/tests.log 2>&1; echo "tests rc=$?"; grep -h "config 4 chain\|drifted trajectory\|config 5 at T=200\|passed\|failed\|^real" $O/tests.log | cut -c1-1200; tail -22 $O/tests.log | grep "s call"
PYTHONPATH=. timeout 900 python tools/fp8_layer_budget.py > $O/fp8_budget.jsonl 2> $O/fp8_budget.log; echo "fp8 budget rc=$?"; cat $O/fp8_budget.jsonl; grep -i "error\|Traceback" -A5 $O/fp8_budget.log | head -20

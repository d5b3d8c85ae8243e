#!/bin/bash
# round 5, lease 14: CFG row sharing on the GPU: the loop / end-to-end / pipeline / full-size-fixture tests, then the bench with
# and without it (K=20)
O=gpurun_out/r05p; mkdir -p $O
timeout 700 python -m pytest -x -q -m gpu tests/test_gpu_loops.py tests/test_gpu_pipeline.py tests/test_gpu_unet.py \
  tests/test_gpu_zzz_fullsize_oracle_fixture.py tests/test_gpu_e2e.py > $O/tests.log 2>&1; echo "tests rc=$?"; tail -25 $O/tests.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batched --no-extras > $O/bench_share.json 2> $O/bench_share.err; echo "share rc=$?"; tail -c 1500 $O/bench_share.json
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batched --no-extras --no-share-cfg-rows > $O/bench_noshare.json 2> $O/bench_noshare.err; echo "noshare rc=$?"; tail -c 1500 $O/bench_noshare.json

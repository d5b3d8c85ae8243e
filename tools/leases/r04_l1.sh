#!/bin/bash
# Round 4, lease 1: observed values of the un-xfailed tests, bf16x6 acceptance subset, pipeline variants, 64/128-CU tile sweeps, RCCL world 1
O=gpurun_out/r04a; mkdir -p $O
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
X=./audioeditingcode_amd/x6_bench
( timeout 150 $X 60 sweep profiles/unet_b2_gemm_ops.txt cus=64 x6 > $O/sweep_B2_cus64_x6.json 2> $O/sweep_B2_cus64_x6.err
  timeout 150 $X 60 sweep profiles/unet_b2_gemm_ops.txt cus=128 x6 > $O/sweep_B2_cus128_x6.json 2> $O/sweep_B2_cus128_x6.err
  timeout 200 $X 12 sweep profiles/unet_b200_gemm_ops.txt cus=128 x6 > $O/sweep_B200_cus128_x6.json 2> $O/sweep_B200_cus128_x6.err
  timeout 200 $X 12 sweep profiles/unet_b200_gemm_ops.txt x6 > $O/sweep_B200_x6.json 2> $O/sweep_B200_x6.err ) 
echo "sweeps done $(date +%T)"
timeout 600 python -m pytest -m gpu -q -s -x tests/test_gpu_zzz_fullsize_oracle_fixture.py tests/test_gpu_zz_split_bf16.py tests/test_gpu_dist.py > $O/tests_f32.log 2>&1; echo "tests f32 rc=$? $(date +%T)"
timeout 500 python -m pytest -m gpu -q -s -x --arith bf16x6 tests/test_gpu_zzz_fullsize_oracle_fixture.py tests/test_gpu_pipeline.py "tests/test_gpu_loops.py::test_full_size_headline_length_batched_vs_sequential" tests/test_gpu_e2e.py::test_clip_edit_end_to_end_vs_oracle > $O/tests_x6.log 2>&1; echo "tests x6 rc=$? $(date +%T)"
B="--steps 10 --warmup 2 --no-extras --no-cpu-baseline --no-batched"
timeout 300 python bench.py $B --arith bf16x6 --edit-lanes 2 > $O/bench_x6_l2.json 2> $O/bench_x6_l2.err; echo "bench x6 l2 rc=$? $(date +%T)"
timeout 300 python bench.py $B --arith bf16x6 --edit-lanes 1 > $O/bench_x6_l1.json 2> $O/bench_x6_l1.err; echo "bench x6 l1 rc=$? $(date +%T)"
timeout 300 python bench.py $B --arith bf16x6 --plan lanes --lanes 4 --lane-cus 64 > $O/bench_x6_lanes4.json 2> $O/bench_x6_lanes4.err; echo "bench lanes rc=$? $(date +%T)"
timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29633 bench.py --gpus 1 --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-batched > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err; echo "torchrun rc=$? $(date +%T)"
tail -3 $O/tests_f32.log $O/tests_x6.log; for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get('roofline') or {}
    print(sys.argv[1], 'value', round(d['value'],4), 'ms', round(d['ms_per_step'],1), 'frac', r.get('frac'), 'path', r.get('path_frac'), (d.get('pipeline') or {}).get('device_ms'))
except Exception as e: print(sys.argv[1], 'unreadable', e)
PY
done

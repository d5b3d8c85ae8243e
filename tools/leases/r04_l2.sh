#!/bin/bash
# Round 4, lease 2: x6 as the default (edit engines too, swept 64/128-CU tables): lane interference, pipeline variants, new tests
O=gpurun_out/r04b; mkdir -p $O
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
X=./audioeditingcode_amd/x6_bench
timeout 150 $X 60 sweep profiles/unet_b2_gemm_ops.txt x6 > $O/sweep_B2_x6.json 2> $O/sweep_B2_x6.err; echo "sweep done $(date +%T)"
timeout 200 python tools/lane_interference.py > $O/interf_default.json 2> $O/interf_default.err; echo "interf rc=$? $(date +%T)"
GPU_MAX_HW_QUEUES=8 timeout 200 python tools/lane_interference.py > $O/interf_q8.json 2> $O/interf_q8.err; echo "interf q8 rc=$? $(date +%T)"
cat $O/interf_default.json $O/interf_q8.json
B="--steps 10 --warmup 2 --no-extras --no-cpu-baseline"
timeout 400 python bench.py $B --serial-clips 2 > $O/bench_l1.json 2> $O/bench_l1.err; echo "bench l1 rc=$? $(date +%T)"
timeout 300 python bench.py $B --no-batched --edit-lanes 2 > $O/bench_l2.json 2> $O/bench_l2.err; echo "bench l2 rc=$? $(date +%T)"
GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py $B --no-batched --edit-lanes 2 > $O/bench_l2_q8.json 2> $O/bench_l2_q8.err; echo "bench l2 q8 rc=$? $(date +%T)"
timeout 300 python bench.py $B --no-batched --plan lanes --lanes 4 --lane-cus 64 > $O/bench_lanes4.json 2> $O/bench_lanes4.err; echo "bench lanes4 rc=$? $(date +%T)"
timeout 600 python -m pytest -m gpu -q -s -x tests/test_gpu_dist.py tests/test_gpu_pc.py::test_full_size_power_iteration_vs_the_oracle_fixture tests/test_gpu_loops.py tests/test_gpu_zzz_fullsize_oracle_fixture.py::test_full_size_headline_length_loops_vs_the_oracle_fixture > $O/tests.log 2>&1; echo "tests rc=$? $(date +%T)"
grep -E "passed|failed|HIP vs oracle|config 4|RCCL|live oracle" $O/tests.log | tail -8
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get('roofline') or {}
    print(sys.argv[1], 'value', round(d['value'],4), 'ms', round(d['ms_per_step'],1), 'frac', r.get('frac'), 'path', r.get('path_frac'), (d.get('pipeline') or {}).get('device_ms'), d.get('value_single_clip_batched'), d.get('value_reference_order'), d.get('schedule_deviation_rel_l2'))
except Exception as e: print(sys.argv[1], 'unreadable', e)
PY
done

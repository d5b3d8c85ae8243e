#!/bin/bash
# Round 4, lease 4: dispatch-pipe separation of the lanes' queues: pipeline variants; updated PC tolerances; pipeline tests
O=gpurun_out/r04d; mkdir -p $O
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
B="--steps 12 --warmup 2 --no-extras --no-cpu-baseline --no-batched"
timeout 300 python bench.py $B --edit-lanes 2 > $O/bench_l2.json 2> $O/bench_l2.err; echo "bench l2 rc=$? $(date +%T)"
timeout 300 python bench.py $B --plan lanes --lanes 4 --lane-cus 64 > $O/bench_lanes4.json 2> $O/bench_lanes4.err; echo "bench lanes4 rc=$? $(date +%T)"
timeout 300 python bench.py $B --edit-lanes 2 --edit-cus 128 --no-overlap-prep > $O/bench_l2_noprep.json 2> $O/bench_l2_noprep.err; echo "bench l2 noprep rc=$? $(date +%T)"
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get('roofline') or {}; p=d.get('pipeline') or {}
    print(sys.argv[1], 'value', round(d['value'],4), 'ms', round(d['ms_per_step'],1), 'path', r.get('path_frac'), p.get('device_ms'), 'lat', p.get('clip_latency_ms_avg'))
    print('   queues', p.get('queue_separation'))
except Exception as e: print(sys.argv[1], 'unreadable', e)
PY
done
timeout 700 python -m pytest -m gpu -q -s -x tests/test_gpu_pipeline.py tests/test_gpu_pc.py > $O/tests.log 2>&1; echo "tests rc=$? $(date +%T)"
grep -E "passed|failed|config 4|PC CLIs|Error" $O/tests.log | tail -8

#!/bin/bash
# Round 4, lease 10: edit lanes widen on drain (test + bench), where an edit lane's step goes (per-op profile on a 64-CU stream),
# attention kernels at batch 2 on a 64-CU lane
O=gpurun_out/r04j; mkdir -p $O
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m pytest -m gpu -q -x -s tests/test_gpu_pipeline.py > $O/tests_pipeline.log 2>&1; echo "pipeline tests rc=$? $(date +%T)"
grep -E "passed|failed|Error|assert" $O/tests_pipeline.log | tail -8
timeout 200 python tools/lane_perop.py 64 > $O/lane_perop_cus64.json 2> $O/lane_perop.err; echo "perop rc=$? $(date +%T)"
python - $O/lane_perop_cus64.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print('per-op sum', d['per_op_sum_ms'], 'graph replay', d['graph_replay_ms'], 'ops', d['ops'])
for c in d['by_category'][:28]: print('  %-28s rows %6d  n %3d  %7.3f ms  %5.1f%%  %s TF/s' % (c['category'], c['rows'], c['launches'], c['ms'], 100*c['share'], c['tflops']))
PY
timeout 200 python tools/attn_x6_ab.py > $O/attn_x6_ab.json 2> $O/attn_x6_ab.err; echo "ab rc=$? $(date +%T)"; cat $O/attn_x6_ab.json
B="--warmup 2 --no-extras --no-cpu-baseline --no-batched"
timeout 280 python bench.py $B --steps 10 > $O/bench_widen.json 2> $O/bench_widen.err; echo "bench rc=$? $(date +%T)"
timeout 280 python bench.py $B --steps 20 > $O/bench_widen20.json 2> $O/bench_widen20.err; echo "bench20 rc=$? $(date +%T)"
for f in $O/bench_widen.json $O/bench_widen20.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get('roofline') or {}; p=d.get('pipeline') or {}
    print(sys.argv[1], 'value', round(d['value'],4), 'ms', round(d['ms_per_step'],1), 'path', r.get('path_frac'), 'lat', p.get('clip_latency_ms_avg'))
    print('   ', {k: round(v['avg'],1) for k, v in (p.get('device_ms') or {}).items()}, 'widened', p.get('widened_on_drain'))
    print('    drain queues', p.get('drain_queue_separation'))
    print('    timeline', p.get('timeline')[-12:])
except Exception as e: print(sys.argv[1], 'unreadable', e)
PY
done

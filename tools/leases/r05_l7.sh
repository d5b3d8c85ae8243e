#!/bin/bash
# round 5, lease 7: narrower edit lanes (3 x 32 CUs beside a 160-CU inversion partition; 4 x 32 beside 128), K = 20
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05g; mkdir -p $O
PYTHONPATH=. timeout 120 python tools/batch_scaling.py 32 2,4 > $O/batch_scaling_cus32.jsonl 2>/dev/null; cat $O/batch_scaling_cus32.jsonl
run() { n=$1; shift
  timeout 420 python bench.py --steps 20 --warmup 2 --no-extras --no-cpu-baseline --no-batched "$@" > $O/$n.json 2> $O/$n.log
  echo "$n rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); p=d.get('pipeline',{}); r=d.get('roofline') or {}
    print(round(d['value'],4), 'ms/clip', round(d['ms_per_step'],1), {k:round(v['avg'],1) for k,v in p.get('device_ms',{}).items()}, 'widened', p.get('widened_on_drain'))
    print('   roofline frac', r.get('frac'), 'achieved', r.get('achieved'), 'fp32eq', r.get('achieved_fp32_equiv'), 'part', (r.get('on_partition') or {}).get('frac'), 'path', (r.get('path') or {}).get('matrix_pipe_frac'))
    print('   queues', p.get('queue_separation'))
except Exception as e: print('ERR', e)
PY
)"; grep -i "error\|Traceback" $O/$n.log | head -3; }
run l3x32 --edit-cus 96 --edit-lanes 3
run l4x32 --edit-cus 128 --edit-lanes 4
run l2x64 --edit-cus 128 --edit-lanes 2

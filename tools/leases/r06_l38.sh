#!/bin/bash
# round 6, lease 38: the driver's bench invocation on the final tree, then the whole GPU suite
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06al; mkdir -p $O
( time timeout 1100 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_k20.json 2> $O/bench_k20.log; echo "k20 bench rc=$? $(date +%T)"
python - <<PY
import json
d=json.loads(open('$O/bench_k20.json').read().strip().splitlines()[-1]); p=d.get('pipeline',{}); r=d.get('roofline') or {}
print(round(d['value'],4), 'clips/s  ms/clip', round(d['ms_per_step'],1), 'steps', d['steps'], {k:round(v['avg'],1) for k,v in p.get('device_ms',{}).items()})
print('   roofline frac', r.get('frac'), 'achieved', r.get('achieved'), 'traffic', r.get('traffic'), 'part', (r.get('on_partition') or {}).get('frac'))
print('   edit_step', {k:(v['ms_per_step_as_graph'], v['launches']) for k,v in (r.get('edit_step') or {}).items()})
print('   single', d.get('value_single_clip_batched'), d.get('value_reference_order'), d.get('pipeline_vs_one_clip_at_a_time'))
print('   cpu', {k: d['cpu_baseline'].get(k) for k in ('value','cores','kind')} if d.get('cpu_baseline') else None)
print('   subs', {k: (v.get('value') if isinstance(v, dict) else v) for k, v in d.items() if k.startswith('config')})
PY
tail -3 $O/bench_k20.log
timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=25 > $O/suite.log 2>&1; echo "suite rc=$?"; tail -30 $O/suite.log

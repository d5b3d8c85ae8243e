#!/bin/bash
# round 6, lease 21: the driver's two steps on the final tree: the whole GPU suite (timed against its 20-minute limit) and its bench invocation
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06u; mkdir -p $O
( time timeout 1500 python -m pytest tests/ -x -q -m gpu ) > $O/suite.log 2>&1; echo "suite rc=$? $(date +%T)"; tail -24 $O/suite.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_k20.json 2> $O/bench_k20.log; echo "bench rc=$? $(date +%T)"
python - <<PY
import json
d=json.loads(open('$O/bench_k20.json').read().strip().splitlines()[-1]); p=d.get('pipeline',{}); r=d.get('roofline') or {}
print(round(d['value'],4), 'clips/s  ms/clip', round(d['ms_per_step'],1), 'steps', d['steps'], {k:round(v['avg'],1) for k,v in p.get('device_ms',{}).items()})
print('   roofline frac', r.get('frac'), 'achieved', r.get('achieved'), 'traffic', r.get('traffic'), 'part', (r.get('on_partition') or {}).get('frac'), 'path', (r.get('path') or {}).get('matrix_pipe_frac'))
print('   edit_step', {k:(v['ms_per_step_as_graph'], v['launches']) for k,v in (r.get('edit_step') or {}).items()})
print('   fwd', {k:(round(v['ms'],2), v.get('frac')) for k,v in r['forward']['families'].items()})
print('   single', d.get('value_single_clip_batched'), d.get('value_reference_order'), d.get('pipeline_vs_one_clip_at_a_time'))
print('   cpu', d.get('cpu_baseline'))
print('   subs', {k: (v.get('value') if isinstance(v, dict) else v) for k, v in d.items() if k.startswith('config')})
PY
tail -3 $O/bench_k20.log | cut -c1-600

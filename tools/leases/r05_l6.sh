#!/bin/bash
# round 5, lease 6: tuned group tables (160|96 g8) vs the two-lane default at K = 20, with the new roofline leg
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05f; mkdir -p $O
run() { n=$1; shift
  timeout 420 python bench.py --steps 20 --warmup 2 --no-extras --no-cpu-baseline "$@" > $O/$n.json 2> $O/$n.log
  echo "$n rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); p=d.get('pipeline',{}); r=d.get('roofline') or {}
    print(round(d['value'],4), 'ms/clip', round(d['ms_per_step'],1), 'groups', p.get('groups_formed'), {k:round(v['avg'],1) for k,v in p.get('device_ms',{}).items()})
    print('   roofline frac', r.get('frac'), 'achieved', r.get('achieved'), 'fp32eq', r.get('achieved_fp32_equiv'), 'part', (r.get('on_partition') or {}).get('frac'), 'path', r.get('path'))
    print('   edit_step', {k:(v['ms_per_step_as_graph'], v['ms_per_clip_step']) for k,v in (r.get('edit_step') or {}).items()})
    print('   single', d.get('value_single_clip_batched'), d.get('value_reference_order'), d.get('pipeline_vs_one_clip_at_a_time'))
except Exception as e: print('ERR', e)
PY
)"; tail -3 $O/$n.log | cut -c1-300; }
run g8_cus96_tuned --edit-cus 96 --edit-lanes 1 --edit-group 8 --serial-clips 2
run base_2lanes --no-batched
timeout 100 python -m pytest -q -m gpu tests/test_gpu_zz_split_bf16.py -k "harness" 2>&1 | tail -3

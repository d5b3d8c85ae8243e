#!/bin/bash
# round 5, lease 2: group plan sweep (edit_cus x edit_group), 16 clips each, headline only
mkdir -p gpurun_out/r05b
cd "$GRAFT_REPO_ROOT"
run() { # name, args...
  n=$1; shift
  timeout 300 python bench.py --steps 16 --warmup 2 --no-extras --no-cpu-baseline --no-batched "$@" > gpurun_out/r05b/$n.json 2> gpurun_out/r05b/$n.log
  echo "$n rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r05b/$n.json').read().strip().splitlines()[-1])
    p=d.get('pipeline',{})
    print(round(d['value'],4), 'ms/clip', round(d['ms_per_step'],1), 'groups', p.get('groups_formed'), {k:round(v['avg'],1) for k,v in p.get('device_ms',{}).items()})
except Exception as e: print('ERR', e)
PY
)"
}
run base_128x2 --edit-cus 128 --edit-lanes 2
run g4_cus128 --edit-cus 128 --edit-lanes 1 --edit-group 4
run g8_cus96 --edit-cus 96 --edit-lanes 1 --edit-group 8
run g4_cus96 --edit-cus 96 --edit-lanes 1 --edit-group 4
run g8_cus128 --edit-cus 128 --edit-lanes 1 --edit-group 8
tail -5 gpurun_out/r05b/g8_cus96.log

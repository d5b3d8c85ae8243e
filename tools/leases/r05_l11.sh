#!/bin/bash
# round 5, lease 11: work stealing with the GPU-aware trigger (K = 20 and K = 40), GPU tests of the stealing path
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05k; mkdir -p $O
run() { n=$1; k=$2; shift; shift
  timeout 500 python bench.py --steps $k --warmup 2 --no-extras --no-cpu-baseline --no-batched "$@" > $O/$n.json 2> $O/$n.log
  echo "$n rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); p=d.get('pipeline',{})
    print(round(d['value'],4), 'ms/clip', round(d['ms_per_step'],1), {k:round(v['avg'],1) for k,v in p.get('device_ms',{}).items()}, 'stolen', p.get('clips_inverted_by_edit_lanes'), 'lat', round(p.get('clip_latency_ms_avg') or 0))
except Exception as e: print('ERR', e)
PY
)"; grep -i "error\|Traceback" $O/$n.log | head -3; }
run steal_k20 20 --steal
run nosteal_k20 20
run steal_k40 40 --steal
timeout 300 python -m pytest -q -m gpu -x -s tests/test_gpu_pipeline.py -k "stealing" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log

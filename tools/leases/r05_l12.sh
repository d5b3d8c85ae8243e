#!/bin/bash
# round 5, lease 12: codec on the edit lanes + set-up on a side stream (the inversion queue keeps the two U-Net calls)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05l; mkdir -p $O
run() { n=$1; k=$2; shift; shift
  timeout 500 python bench.py --steps $k --warmup 2 --no-extras --no-cpu-baseline --no-batched "$@" > $O/$n.json 2> $O/$n.log
  echo "$n rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); p=d.get('pipeline',{})
    print(round(d['value'],4), 'ms/clip', round(d['ms_per_step'],1), {k:round(v['avg'],1) for k,v in p.get('device_ms',{}).items()}, 'lat', round(p.get('clip_latency_ms_avg') or 0))
except Exception as e: print('ERR', e)
PY
)"; grep -i "error\|Traceback" $O/$n.log | head -3; }
run codec_lane 20 --codec-queue lane
run codec_lane_noprep 20 --codec-queue lane --no-overlap-prep
run codec_front 20

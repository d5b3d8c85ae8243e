#!/bin/bash
# round 5, lease 10: work stealing (edit lanes invert clips when idle), deep-prefetch sweep on a 64-CU stream
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05j; mkdir -p $O
X=./audioeditingcode_amd/x6_bench
timeout 120 $X 1 wide > $O/wide_deep_cases.jsonl 2> $O/wide_deep_cases.err; echo "wide+deep cases rc=$?"; tail -2 $O/wide_deep_cases.err; grep '"pass": false' $O/wide_deep_cases.jsonl | head -5
run() { n=$1; shift
  timeout 420 python bench.py --steps 20 --warmup 2 --no-extras --no-cpu-baseline --no-batched "$@" > $O/$n.json 2> $O/$n.log
  echo "$n rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); p=d.get('pipeline',{})
    print(round(d['value'],4), 'ms/clip', round(d['ms_per_step'],1), {k:round(v['avg'],1) for k,v in p.get('device_ms',{}).items()}, 'stolen', p.get('clips_inverted_by_edit_lanes'), 'lat', round(p.get('clip_latency_ms_avg') or 0))
except Exception as e: print('ERR', e)
PY
)"; grep -i "error\|Traceback" $O/$n.log | head -3; }
run steal --steal
run steal_codec_chip --steal --codec-queue chip
timeout 300 $X 30 sweep profiles/unet_b2_gemm_ops.txt cus=64 x6 > $O/sweep_B2_cus64_x6_deep.json 2> $O/sweep_B2_cus64_x6_deep.err; echo "sweep deep cus64 rc=$? $(date +%T)"; tail -1 $O/sweep_B2_cus64_x6_deep.err

#!/bin/bash
# round 5, lease 5: group plan at K = 20 (192|64 and 160|96, groups of up to 8), tile sweeps of the lockstep batch shapes on a 64-CU stream
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05e; mkdir -p $O
X=./audioeditingcode_amd/x6_bench
run() { n=$1; shift
  timeout 300 python bench.py --steps 20 --warmup 2 --no-extras --no-cpu-baseline --no-batched "$@" > $O/$n.json 2> $O/$n.log
  echo "$n rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); p=d.get('pipeline',{})
    print(round(d['value'],4), 'ms/clip', round(d['ms_per_step'],1), 'groups', p.get('groups_formed'), {k:round(v['avg'],1) for k,v in p.get('device_ms',{}).items()}, 'lat', round(p.get('clip_latency_ms_avg') or 0))
except Exception as e: print('ERR', e)
PY
)"; }
run g8_cus64 --edit-cus 64 --edit-lanes 1 --edit-group 8
run g8_cus96 --edit-cus 96 --edit-lanes 1 --edit-group 8
for b in 16 8 4; do
  timeout 200 $X 12 sweep profiles/unet_b${b}_gemm_ops.txt cus=64 x6 > $O/sweep_B${b}_cus64_x6.json 2> $O/sweep_B${b}_cus64_x6.err; echo "sweep B$b cus64 rc=$? $(date +%T)"
done
timeout 200 $X 12 sweep profiles/unet_b16_gemm_ops.txt cus=96 x6 > $O/sweep_B16_cus96_x6.json 2> $O/sweep_B16_cus96_x6.err; echo "sweep B16 cus96 rc=$? $(date +%T)"
ls -la $O

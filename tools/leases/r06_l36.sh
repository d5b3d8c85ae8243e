#!/bin/bash
# round 6, lease 36: the head of the edit loop on the front stage (edit_head_steps): tiny test, then bench A/B k = 0 / 2 / 3 / 4 / 6
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06aj; mkdir -p $O
timeout 600 python -m pytest -q -m gpu -x tests/test_gpu_pipeline.py -k "edit_head or tiny" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for k in 0 3 2 4 6; do
  timeout 420 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline --no-batched --edit-head-steps $k > $O/bench_k$k.json 2> $O/bench_k$k.log; echo "bench k=$k rc=$?"
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench_k$k.json').read().strip().splitlines()[-1]); p=d.get('pipeline',{})
    print('k=$k', round(d['value'],4), 'clips/s  ms/clip', round(d['ms_per_step'],1), {k:round(v['avg'],1) for k,v in p.get('device_ms',{}).items()}, d.get('pipeline_vs_one_clip_at_a_time',{}).get('bit_identical_to_same_engines_alone'), 'parity', (d.get('parity') or {}).get('rel_l2_latent'))
except Exception as e: print('ERR', e)
PY
done

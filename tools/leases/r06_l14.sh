#!/bin/bash
# round 6, lease 14: sharing back on in the edit loop, side stream masked: co-residency tests, pipeline tests, kernel + U-Net parity,
# then the bench A/B: shipped / sharing in the inversion only / unmasked set-up stream
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06n; mkdir -p $O
timeout 1500 python -m pytest -q -m gpu -x tests/test_gpu_coresidency.py tests/test_gpu_pipeline.py tests/test_gpu_kernels.py tests/test_gpu_unet.py tests/test_gpu_zz_split_bf16.py > $O/tests.log 2>&1; echo "tests rc=$?"; tail -25 $O/tests.log
for tag in shipped "noshare_edit --no-share-in-edit-loop" "unmasked_prep --unmasked-prep"; do set -- $tag; t=$1; shift
  timeout 420 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline --no-batched "$@" > $O/bench_$t.json 2> $O/bench_$t.log; echo "bench $t rc=$?"
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$t.json').read().strip().splitlines()[-1]); p=d.get('pipeline',{}); r=d.get('roofline') or {}
    print('$t', round(d['value'],4), 'clips/s  ms/clip', round(d['ms_per_step'],1), {k:round(v['avg'],1) for k,v in p.get('device_ms',{}).items()})
    print('   edit_step', {k:(v['ms_per_step_as_graph'], v['launches']) for k,v in (r.get('edit_step') or {}).items()}, 'frac', r.get('frac'), d.get('pipeline_vs_one_clip_at_a_time'))
except Exception as e: print('ERR', e)
PY
done
grep -i "error\|Traceback" $O/bench_*.log | head -5

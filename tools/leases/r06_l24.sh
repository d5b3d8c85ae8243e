#!/bin/bash
# round 6, lease 24: swept tiles for the batch-1 head of the CFG-shared edit engine: parity / pipeline / co-residency tests, bench A/B
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06x; mkdir -p $O
timeout 1500 python -m pytest -q -m gpu -x tests/test_gpu_coresidency.py tests/test_gpu_pipeline.py tests/test_gpu_unet.py tests/test_gpu_loops.py tests/test_gpu_zzz_fullsize_oracle_fixture.py > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log
for tag in shipped "noshare_edit --no-share-in-edit-loop" "shipped_again"; do set -- $tag; t=$1; shift
  timeout 420 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline --no-batched "$@" > $O/bench_$t.json 2> $O/bench_$t.log; echo "bench $t rc=$?"
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$t.json').read().strip().splitlines()[-1]); p=d.get('pipeline',{}); r=d.get('roofline') or {}
    print('$t', round(d['value'],4), 'clips/s  ms/clip', round(d['ms_per_step'],1), {k:round(v['avg'],1) for k,v in p.get('device_ms',{}).items()})
    print('   edit_step', {k:(v['ms_per_step_as_graph'], v['launches']) for k,v in (r.get('edit_step') or {}).items()}, d.get('pipeline_vs_one_clip_at_a_time',{}).get('bit_identical_to_same_engines_alone'))
except Exception as e: print('ERR', e)
PY
done

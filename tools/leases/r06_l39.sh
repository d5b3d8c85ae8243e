#!/bin/bash
# round 6, lease 39: gn_apply grid of 8 blocks per CU (was 2): GroupNorm / U-Net / codec / pipeline tests, bench
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06am; mkdir -p $O
timeout 1200 python -m pytest -q -m gpu -x tests/test_gpu_kernels.py tests/test_gpu_unet.py tests/test_gpu_codec.py tests/test_gpu_pipeline.py tests/test_gpu_stable_audio.py > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
for t in a b; do
timeout 420 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline --no-batched > $O/bench_$t.json 2> $O/bench_$t.log; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open('$O/bench_$t.json').read().strip().splitlines()[-1]); p=d.get('pipeline',{}); r=d.get('roofline') or {}
print(round(d['value'],4), 'clips/s  ms/clip', round(d['ms_per_step'],1), {k:round(v['avg'],1) for k,v in p.get('device_ms',{}).items()})
f=r['forward']; print('   forward ms as graph', f['ms_as_graph'], 'gn', {k:round(v,3) for k,v in f['families']['groupnorm'].items() if isinstance(v,float)}, 'part', r['on_partition']['ms_as_graph'], r['on_partition']['families']['groupnorm']['ms'])
PY
done

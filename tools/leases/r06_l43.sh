#!/bin/bash
# round 6, lease 43: REPEAT of the driver's bench invocation on the final tree (lease 42 measured 1.159 clips/s there against 1.195 on the
# pre-GEGLU tree in lease 38; the only stage time that moved was the drain, back_chip 1666 vs ~1100 ms).  Same command, nothing else changed.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06aq; mkdir -p $O
( time timeout 1100 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_k20_repeat.json 2> $O/bench_k20_repeat.log; echo "k20 repeat rc=$? $(date +%T)"
python - <<PY
import json
d=json.loads(open('$O/bench_k20_repeat.json').read().strip().splitlines()[-1]); p=d['pipeline']; r=d['roofline']
print(round(d['value'],4), 'clips/s  ms/clip', round(d['ms_per_step'],1), {k:round(v['avg'],1) for k,v in p['device_ms'].items()}, 'widened', p.get('widened_on_drain'))
print('   frac', r['frac'], 'part', r['on_partition']['frac'], 'hash', r['csrc_hash'], 'tail', p['timeline'][-4:])
PY

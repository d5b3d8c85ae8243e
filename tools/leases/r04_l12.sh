#!/bin/bash
# Round 4, lease 12: MX-FP8 experiment (kernel vs CPU emulation, DiT deviation, Stable Audio clip in fp8); headline bench with the
# split-bf16 attention on the edit lanes
O=gpurun_out/r04l; mkdir -p $O
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 400 python -m pytest -m gpu -q -x -s tests/test_gpu_fp8_experiment.py > $O/tests_fp8.log 2>&1; echo "fp8 tests rc=$? $(date +%T)"
grep -E "\[fp8|passed|failed|Error|assert|^E " $O/tests_fp8.log | tail -20
timeout 300 python tools/bench_stable_audio.py --arith fp8 --steps 1 --warmup 1 > $O/bench_sa_fp8.json 2> $O/bench_sa_fp8.err; echo "sa fp8 rc=$? $(date +%T)"
tail -c 1800 $O/bench_sa_fp8.json; echo; tail -3 $O/bench_sa_fp8.err
timeout 300 python tools/bench_stable_audio.py --steps 1 --warmup 1 > $O/bench_sa_x6.json 2> $O/bench_sa_x6.err; echo "sa x6 rc=$? $(date +%T)"
python - $O/bench_sa_fp8.json $O/bench_sa_x6.json <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, 's/clip', round(d['ms_per_step']/1e3,3), d.get('parity_T200'), d.get('phases_s_one_clip'))
    except Exception as e: print(f, 'unreadable', e)
PY
B="--warmup 2 --no-extras --no-cpu-baseline --no-batched"
timeout 280 python bench.py $B --steps 20 > $O/bench_k20.json 2> $O/bench_k20.err; echo "bench20 rc=$? $(date +%T)"
python - "$O/bench_k20.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get('roofline') or {}; p=d.get('pipeline') or {}
    print(sys.argv[1], 'value', round(d['value'],4), 'ms', round(d['ms_per_step'],1), 'path', r.get('path_frac'), 'lat', p.get('clip_latency_ms_avg'))
    print('   ', {k: round(v['avg'],1) for k, v in (p.get('device_ms') or {}).items()}, 'widened', p.get('widened_on_drain'))
    print('    b2', (r.get('by_batch') or {}).get('unet_batch_2'))
    print('    timeline', p.get('timeline')[-14:])
except Exception as e: print(sys.argv[1], 'unreadable', e)
PY

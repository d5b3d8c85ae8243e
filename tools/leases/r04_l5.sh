#!/bin/bash
# Round 4, lease 5: codec stage (three-stage partition pipeline), config 3 / config 5 with the split-bf16 arithmetic
O=gpurun_out/r04e; mkdir -p $O
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
B="--steps 12 --warmup 2 --no-extras --no-cpu-baseline --no-batched"
timeout 300 python bench.py $B > $O/bench_l2_codec.json 2> $O/bench_l2_codec.err; echo "bench l2 codec rc=$? $(date +%T)"
timeout 300 python bench.py $B --no-overlap-prep > $O/bench_l2_codec_noprep.json 2> $O/bench_l2_codec_noprep.err; echo "bench l2 codec noprep rc=$? $(date +%T)"
timeout 300 python bench.py --clips-per-gpu 8 --steps 1 --warmup 1 --lanes 1 --no-extras --no-cpu-baseline --no-batched > $O/bench_config3.json 2> $O/bench_config3.err; echo "config3 rc=$? $(date +%T)"
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get('roofline') or {}; p=d.get('pipeline') or {}
    print(sys.argv[1], 'value', round(d['value'],4), 'ms', round(d['ms_per_step'],1), 'path', r.get('path_frac'), p.get('device_ms'), 'lat', p.get('clip_latency_ms_avg'))
    print('   queues', p.get('queue_separation'))
except Exception as e: print(sys.argv[1], 'unreadable', e)
PY
done
timeout 300 python tools/bench_stable_audio.py --arith bf16x6 --steps 1 --warmup 1 > $O/bench_sa_x6.json 2> $O/bench_sa_x6.err; echo "sa x6 rc=$? $(date +%T)"; tail -c 1500 $O/bench_sa_x6.json
timeout 400 python -m pytest -m gpu -q -s -x tests/test_gpu_pipeline.py > $O/tests.log 2>&1; echo "tests rc=$? $(date +%T)"
grep -E "passed|failed|Error" $O/tests.log | tail -5

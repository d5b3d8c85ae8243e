#!/bin/bash
# Round 4, lease 8: front-queue relief (codec engines on split-bf16 GEMMs), edit lanes sharing the edit partition (own dispatch
# pipes), per-job device timeline of the pipeline; codec / e2e parity tests under --codec-arith bf16x6
O=gpurun_out/r04h; mkdir -p $O
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
B="--warmup 2 --no-extras --no-cpu-baseline --no-batched"
run() { tag=$1; shift; timeout 280 python bench.py $B "$@" > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "bench $tag rc=$? $(date +%T)"; }
run base --steps 10
run codecx6 --steps 10 --codec-arith bf16x6
run lanes3_shared --steps 9 --edit-lanes 3
run lanes2_shared --steps 8 --edit-lanes 2 --edit-share
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get('roofline') or {}; p=d.get('pipeline') or {}
    print(sys.argv[1], 'value', round(d['value'],4), 'ms', round(d['ms_per_step'],1), 'path', r.get('path_frac'), 'lat', p.get('clip_latency_ms_avg'))
    print('   ', {k: round(v['avg'],1) for k, v in (p.get('device_ms') or {}).items()})
    print('    queues', [(q['cus'], q['attempt'], q['delay']) for q in p.get('queue_separation') or []])
    print('    timeline', p.get('timeline'))
except Exception as e: print(sys.argv[1], 'unreadable', e)
PY
done
timeout 400 python -m pytest -m gpu -q -x --codec-arith bf16x6 tests/test_gpu_codec.py tests/test_gpu_e2e.py::test_clip_edit_end_to_end_vs_oracle > $O/tests_codecx6.log 2>&1; echo "tests codec x6 rc=$? $(date +%T)"
grep -E "passed|failed|Error|assert" $O/tests_codecx6.log | tail -8

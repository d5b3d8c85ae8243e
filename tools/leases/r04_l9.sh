#!/bin/bash
# Round 4, lease 9: split-bf16 self-attention (csrc/attention_x6.hip): kernel parity, A/B timing, full-size U-Net / loop parity, bench
O=gpurun_out/r04i; mkdir -p $O
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m pytest -m gpu -q -x -s tests/test_gpu_zz_split_bf16.py -k "attention" > $O/tests_attn.log 2>&1; echo "attention x6 tests rc=$? $(date +%T)"
grep -E "attention x6|passed|failed|Error|assert" $O/tests_attn.log | tail -12
timeout 200 python tools/attn_x6_ab.py > $O/attn_x6_ab.json 2> $O/attn_x6_ab.err; echo "ab rc=$? $(date +%T)"; cat $O/attn_x6_ab.json; tail -3 $O/attn_x6_ab.err
timeout 500 python -m pytest -m gpu -q -x -s tests/test_gpu_kernels.py -k attention tests/test_gpu_unet.py tests/test_gpu_zzz_fullsize_oracle_fixture.py::test_full_size_headline_length_loops_vs_the_oracle_fixture > $O/tests_unet.log 2>&1; echo "unet/loop tests rc=$? $(date +%T)"
grep -E "passed|failed|Error|HIP vs oracle|rel" $O/tests_unet.log | tail -10
B="--warmup 2 --no-extras --no-cpu-baseline --no-batched"
timeout 280 python bench.py $B --steps 10 > $O/bench_attnx6.json 2> $O/bench_attnx6.err; echo "bench rc=$? $(date +%T)"
python - "$O/bench_attnx6.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get('roofline') or {}; p=d.get('pipeline') or {}
    print(sys.argv[1], 'value', round(d['value'],4), 'ms', round(d['ms_per_step'],1), 'path', r.get('path_frac'), 'lat', p.get('clip_latency_ms_avg'))
    print('   ', {k: round(v['avg'],1) for k, v in (p.get('device_ms') or {}).items()})
    print('    b200', (r.get('by_batch') or {}).get('unet_batch_200'))
    print('    timeline', p.get('timeline'))
except Exception as e: print(sys.argv[1], 'unreadable', e)
PY

#!/bin/bash
# r03 call 7: pipeline tests with the 128-CU tile regime + noise prefetch; one complete default bench line (extras included).
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_pipeline.py -x -q > gpurun_out/r03_c7_tests.log 2>&1; echo "tests rc=$?"; tail -12 gpurun_out/r03_c7_tests.log | cut -c1-200
timeout 1500 python bench.py --steps 10 --warmup 3 > gpurun_out/r03_bench_full.json 2> gpurun_out/r03_bench_full.err; echo "bench rc=$?"; grep "^\[bench\]" gpurun_out/r03_bench_full.err | cut -c1-900; grep -v "^\[bench\]\|WARNING\|amdgpu" gpurun_out/r03_bench_full.err | tail -8
python - <<'PY'
import json
try:
    d = json.load(open('gpurun_out/r03_bench_full.json'))
    print({k: d.get(k) for k in ('value', 'ms_per_step', 'value_reference_order', 'value_single_clip_batched', 'parity')})
    r = d['roofline']; print({k: r[k] for k in ('achieved', 'frac', 'path_frac', 'path_frac_executed')})
    for k, v in r['by_batch'].items(): print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()})
    for k in ('config3_per_rank', 'config4_pc_extract_apply', 'config5_stable_audio_fp32'): print(k, json.dumps(d.get(k))[:500])
except Exception as e:
    print('no bench json', e)
PY

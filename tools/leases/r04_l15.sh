#!/bin/bash
# Round 4, lease 15: regression subset after the CGParams change (every GEMM kernel recompiled) + the driver's bench invocation, timed
O=gpurun_out/r04p; mkdir -p $O
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
t0=$(date +%s)
timeout 600 python -m pytest -m gpu -q -x tests/test_gpu_kernels.py tests/test_gpu_codec.py tests/test_gpu_unet.py tests/test_gpu_zz_split_bf16.py tests/test_gpu_stable_audio.py tests/test_gpu_e2e.py::test_graft_entry_smoke > $O/tests.log 2>&1; echo "tests rc=$? $(( $(date +%s) - t0 )) s"
grep -E "passed|failed|Error|^E  " $O/tests.log | tail -6
t0=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$? $(( $(date +%s) - t0 )) s"
python - "$O/bench_driver.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get('roofline') or {}; p=d.get('pipeline') or {}
    print('value', round(d['value'],4), 'ms', round(d['ms_per_step'],1), 'frac', r.get('frac'), 'path', r.get('path_frac'), 'serial', r.get('frac_whole_chip_serial'), 'traffic', r.get('traffic'))
    print('single', d.get('value_single_clip_batched'), 'ref order', d.get('value_reference_order'), 'sched dev', d.get('schedule_deviation_rel_l2'), 'pipe vs alone', d.get('pipeline_vs_one_clip_at_a_time'))
    print('parity', d.get('parity'))
    print('cpu', d.get('cpu_baseline'))
    ex=d.get('extras') or {}
    for k,v in ex.items():
        if isinstance(v, dict): print('  extra', k, {kk: v[kk] for kk in ('value','ms_per_step','seconds','failed','skipped','parity_T200') if kk in v}, (v.get('roofline') or {}).get('path_frac'))
    print('keys', list(d.keys()))
except Exception as e: print(sys.argv[1], 'unreadable', e)
PY
tail -5 $O/bench_driver.err

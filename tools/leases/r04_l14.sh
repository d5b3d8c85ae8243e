#!/bin/bash
# Round 4, lease 14: fp8 experiment with pre-quantised weights: tests, Stable Audio clip
O=gpurun_out/r04o; mkdir -p $O
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 400 python -m pytest -m gpu -q -s tests/test_gpu_fp8_experiment.py > $O/tests_fp8.log 2>&1; echo "fp8 tests rc=$? $(date +%T)"
grep -E "\[fp8|passed|failed|Error|^E  " $O/tests_fp8.log | tail -24
timeout 300 python tools/bench_stable_audio.py --arith fp8 --steps 1 --warmup 1 > $O/bench_sa_fp8.json 2> $O/bench_sa_fp8.err; echo "sa fp8 rc=$? $(date +%T)"
python - $O/bench_sa_fp8.json <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, 's/clip', round(d['ms_per_step']/1e3,3), d.get('parity_T200'), d.get('phases_s_one_clip'), d.get('roofline'))
    except Exception as e: print(f, 'unreadable', e)
PY
tail -3 $O/bench_sa_fp8.err

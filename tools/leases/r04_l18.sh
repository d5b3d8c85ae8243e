#!/bin/bash
# Round 4, lease 18: pipeline tests after the last edits; per-op profile of an edit lane with the split-bf16 attention in place
O=gpurun_out/r04s; mkdir -p $O
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m pytest -m gpu -q -x tests/test_gpu_pipeline.py tests/test_gpu_zz_split_bf16.py -k "pipeline or attention or widen or lanes or census" > $O/tests.log 2>&1; echo "tests rc=$? $(date +%T)"
grep -E "passed|failed|Error|^E  " $O/tests.log | tail -4
timeout 200 python tools/lane_perop.py 64 > $O/lane_perop_cus64.json 2> $O/lane_perop.err; echo "perop rc=$? $(date +%T)"
python - $O/lane_perop_cus64.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print('per-op sum', d['per_op_sum_ms'], 'graph replay', d['graph_replay_ms'], 'ops', d['ops'])
for c in d['by_category'][:12]: print('  %-28s rows %6d  n %3d  %7.3f ms  %5.1f%%  %s TF/s' % (c['category'], c['rows'], c['launches'], c['ms'], 100*c['share'], c['tflops']))
PY

#!/bin/bash
# r03 call 3: lanes vs hardware queues (synthetic), pipeline GPU tests, first lanes bench line, config 4 at full size.
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
for q in default 8 16; do
  if [ "$q" = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  timeout 300 python tools/lanes.py 12 > gpurun_out/r03_lanes_q$q.log 2>&1; echo "lanes q=$q rc=$?"; grep "^queues" gpurun_out/r03_lanes_q$q.log
done
unset GPU_MAX_HW_QUEUES
timeout 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_e2e.py::test_text_encoders_on_the_gpu_match_the_reference_encode_text tests/test_gpu_e2e.py::test_graft_entry_smoke -x -q > gpurun_out/r03_c3_tests.log 2>&1; echo "tests rc=$?"; tail -12 gpurun_out/r03_c3_tests.log
timeout 600 python bench.py --steps 8 --warmup 4 --lanes 4 --no-extras --no-cpu-baseline > gpurun_out/r03_bench_lanes4.json 2> gpurun_out/r03_bench_lanes4.err; echo "bench rc=$?"; grep "^\[bench\]" gpurun_out/r03_bench_lanes4.err | tail -12
python - <<'PY'
import json
try:
    d = json.load(open('gpurun_out/r03_bench_lanes4.json'))
    print({k: d.get(k) for k in ('value', 'ms_per_step', 'value_reference_order', 'value_single_clip_batched', 'schedule_deviation_rel_l2', 'lanes_vs_serial', 'pipeline')})
    r = d['roofline']; print({k: r[k] for k in ('achieved', 'frac', 'path_frac', 'path_frac_executed', 'clip_unet_tflop', 'clip_unet_tflop_executed')})
    for k, v in r['by_batch'].items(): print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()})
except Exception as e:
    print('no bench json', e)
PY
timeout 400 python tools/bench_config4.py > gpurun_out/r03_config4.json 2> gpurun_out/r03_config4.err; echo "config4 rc=$?"; cat gpurun_out/r03_config4.json; tail -3 gpurun_out/r03_config4.err

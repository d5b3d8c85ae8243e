#!/bin/bash
# r03 call 6: pipeline tests with process-lifetime streams; split sweep with the codec off the edit partition; bench line.
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_pipeline.py -x -q > gpurun_out/r03_c6_tests.log 2>&1; echo "tests rc=$?"; tail -12 gpurun_out/r03_c6_tests.log | cut -c1-200
for cfg in 128:1 120:1 112:1; do
timeout 300 python tools/pipeline_sweep.py 8 $cfg > gpurun_out/r03_pipeline_sweep_$cfg.log 2>&1; echo "sweep $cfg rc=$?"; grep "^{" gpurun_out/r03_pipeline_sweep_$cfg.log | cut -c1-600
done
timeout 600 python bench.py --steps 8 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r03_bench_partition.json 2> gpurun_out/r03_bench_partition.err; echo "bench rc=$?"; grep "^\[bench\]" gpurun_out/r03_bench_partition.err | tail -10 | cut -c1-700; grep -v "^\[bench\]\|WARNING\|amdgpu" gpurun_out/r03_bench_partition.err | tail -8
python - <<'PY'
import json
try:
    d = json.load(open('gpurun_out/r03_bench_partition.json'))
    print({k: d.get(k) for k in ('value', 'ms_per_step', 'value_reference_order', 'value_single_clip_batched')})
    r = d['roofline']; print({k: r[k] for k in ('achieved', 'frac', 'path_frac', 'path_frac_executed')})
    for k, v in r['by_batch'].items(): print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()})
except Exception as e:
    print('no bench json', e)
PY

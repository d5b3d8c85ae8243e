#!/bin/bash
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest "tests/test_gpu_pc.py::test_pc_clis_extract_pt_apply_on_the_gpu" -m gpu -x -q > gpurun_out/c9_pc.log 2>&1; echo "pc rc=$?"; tail -2 gpurun_out/c9_pc.log; grep "^E " gpurun_out/c9_pc.log | head -5
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench_r02b.json 2> gpurun_out/bench_r02b.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r02b.json'))
print({k:d[k] for k in ('value','ms_per_step','value_reference_order','ms_per_step_reference_order') if k in d})
r=d['roofline']; print(r['achieved'], r['frac'], r['path_tflops'], r['path_frac'])
for k,v in r['by_batch'].items(): print(k, round(v['forward_ms'],3), round(v['forward_tflops'],1), round(v['conv_gemm_tflops'],1))
PY
tail -3 gpurun_out/bench_r02b.err
rocm-smi --showmeminfo vram 2>/dev/null | head -8

#!/bin/bash
# Full GPU suite + default bench line on the ABI-v3 library.
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/f3_tests.log 2>&1; echo "tests rc=$?"
tail -6 gpurun_out/f3_tests.log
timeout 900 python bench.py > gpurun_out/bench_r02_final3.json 2> gpurun_out/bench_r02_final3.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r02_final3.json'))
print({k:d[k] for k in ('value','ms_per_step','value_reference_order') if k in d})
r=d['roofline']; print('roofline', r['achieved'], r['frac'], 'traffic', r['traffic'], 'path', r['path_frac'], r['csrc_hash'])
for k,v in r['by_batch'].items(): print(k, round(v['forward_ms'],3), round(v['forward_tflops'],1), round(v['conv_gemm_tflops'],1))
PY

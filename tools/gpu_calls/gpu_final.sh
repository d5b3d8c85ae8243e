#!/bin/bash
# Round-2 final measurement pass: full GPU suite, default bench line, headline kernel trace, PMC passes.
export PYTHONPATH=$PWD TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/f_tests.log 2>&1; echo "tests rc=$?"
tail -6 gpurun_out/f_tests.log
timeout 1200 python bench.py > gpurun_out/bench_r02_final.json 2> gpurun_out/bench_r02_final.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r02_final.json'))
print({k:d[k] for k in ('value','ms_per_step','value_reference_order') if k in d})
r=d['roofline']; print('roofline', r['achieved'], r['frac'], 'path', r['path_tflops'], r['path_frac'])
for k,v in r['by_batch'].items(): print(k, round(v['forward_ms'],3), round(v['forward_tflops'],1), round(v['conv_gemm_tflops'],1))
print(d['phases_ms_one_clip']); print(d['cpu_baseline'])
PY
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_r02f -o kt --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-batched > $R/gpurun_out/f_kt.log 2>&1; echo "kt rc=$?"
for pass in "f FETCH_SIZE" "w WRITE_SIZE" "s SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  set -- $pass; tag=$1; shift
  timeout 400 rocprofv3 --pmc "$@" -d $R/gpurun_out/pmc_r02f/$tag -o $tag --output-format csv -- python $R/tools/pmc_forward.py 200 2 > $R/gpurun_out/f_pmc_$tag.log 2>&1; echo "pmc $tag rc=$?"
done
cd $R; grep "forward done" gpurun_out/f_pmc_f.log; du -sh gpurun_out/kt_r02f gpurun_out/pmc_r02f

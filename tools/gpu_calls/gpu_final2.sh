#!/bin/bash
# Round-2 closing pass on the final binary: new/changed tests, default bench line, headline kernel trace, PMC passes.
export PYTHONPATH=$PWD TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_unet.py "tests/test_gpu_loops.py::test_ddpm_inversion_and_edit_match_oracle" "tests/test_gpu_e2e.py::test_clip_edit_end_to_end_vs_oracle" "tests/test_gpu_e2e.py::test_graft_entry_smoke" -m gpu -q > gpurun_out/f2_tests.log 2>&1; echo "tests rc=$?"
tail -4 gpurun_out/f2_tests.log
timeout 1200 python bench.py > gpurun_out/bench_r02_final2.json 2> gpurun_out/bench_r02_final2.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r02_final2.json'))
print({k:d[k] for k in ('value','ms_per_step','value_reference_order') if k in d})
r=d['roofline']; print('roofline', r['achieved'], r['frac'], 'path', r['path_tflops'], r['path_frac'], r['csrc_hash'])
for k,v in r['by_batch'].items(): print(k, round(v['forward_ms'],3), round(v['forward_tflops'],1), round(v['conv_gemm_tflops'],1))
print(d['phases_ms_one_clip'])
PY
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_r02g -o kt --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-batched > $R/gpurun_out/f2_kt.log 2>&1; echo "kt rc=$?"
for pass in "f FETCH_SIZE" "w WRITE_SIZE" "s SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  set -- $pass; tag=$1; shift
  timeout 400 rocprofv3 --pmc "$@" -d $R/gpurun_out/pmc_r02g/$tag -o $tag --output-format csv -- python $R/tools/pmc_forward.py 200 2 > $R/gpurun_out/f2_pmc_$tag.log 2>&1; echo "pmc $tag rc=$?"
done
cd $R; grep "forward done" gpurun_out/f2_pmc_f.log

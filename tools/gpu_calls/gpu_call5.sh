#!/bin/bash
# GPU call 5: full GPU suite, default bench line, headline kernel trace, PMC passes.
export PYTHONPATH=$PWD TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/c5_tests.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/c5_tests.log
timeout 900 python bench.py > gpurun_out/bench_r02a.json 2> gpurun_out/bench_r02a.err; echo "bench rc=$?"
cat gpurun_out/bench_r02a.json | head -c 1500; echo
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_r02 -o kt --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-batched > $R/gpurun_out/c5_kt.log 2>&1; echo "kt rc=$?"
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_r02/fetch -o f --output-format csv -- python $R/tools/pmc_forward.py 40 2 > $R/gpurun_out/c5_pmc_f.log 2>&1; echo "pmc fetch rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_r02/write -o w --output-format csv -- python $R/tools/pmc_forward.py 40 2 > $R/gpurun_out/c5_pmc_w.log 2>&1; echo "pmc write rc=$?"
timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_r02/sq -o s --output-format csv -- python $R/tools/pmc_forward.py 40 2 > $R/gpurun_out/c5_pmc_s.log 2>&1; echo "pmc sq rc=$?"
cd $R
grep "forward done" gpurun_out/c5_pmc_f.log
find gpurun_out/kt_r02 gpurun_out/pmc_r02 -name "*.csv" | head; du -sh gpurun_out/kt_r02 gpurun_out/pmc_r02

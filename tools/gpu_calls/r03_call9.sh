#!/bin/bash
# r03 call 9: the round's evidence run -- full GPU suite, complete bench line, PMC passes on the final GEMM sources, kernel trace
# of the pipelined bench.
export PYTHONPATH=$PWD TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
timeout 1500 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_r03.json 2> gpurun_out/bench_r03.err; echo "bench rc=$?"; grep "^\[bench\]" gpurun_out/bench_r03.err | cut -c1-500; grep -v "^\[bench\]\|WARNING\|amdgpu" gpurun_out/bench_r03.err | tail -6
bash tools/gpu_pmc.sh pmc_r03 2>&1 | tail -5
ALG=$(grep "forward done" gpurun_out/pmc_r03_f.log | awk '{s+=$NF} END {print s}')
python tools/pmc_summary.py gpurun_out/pmc_r03 --json gpurun_out/r03_pmc_forward.json --alg-total-bytes $ALG > gpurun_out/r03_pmc_forward.md 2> gpurun_out/r03_pmc_summary.err; echo "pmc summary rc=$? alg=$ALG"; tail -4 gpurun_out/r03_pmc_forward.md | cut -c1-300; cat gpurun_out/r03_pmc_forward.json | head -12
find gpurun_out/pmc_r03 -name "*.csv" -size +2M -delete
rm -rf gpurun_out/kt_r03; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_r03 -o kt --output-format csv -- python $R/bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline --no-batched > $R/gpurun_out/r03_kt_bench.json 2> $R/gpurun_out/r03_kt_bench.err; echo "rocprof rc=$? (139 = the profiler's teardown crash after the line was printed)"
cd $R
TR=$(find gpurun_out/kt_r03 -name "*kernel_trace.csv" | head -1); ST=$(find gpurun_out/kt_r03 -name "*kernel_stats.csv" | head -1)
python tools/trace_overlap.py $TR > gpurun_out/r03_kernel_trace_pipeline.md 2> gpurun_out/r03_trace_overlap.err; echo "overlap rc=$?"; tail -6 gpurun_out/r03_kernel_trace_pipeline.md | cut -c1-400
cp $ST gpurun_out/r03_rocprofv3_kernel_stats_pipeline.csv; rm -f $TR
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r03_gpu_suite.log 2>&1; echo "suite rc=$?"; tail -25 gpurun_out/r03_gpu_suite.log | cut -c1-200

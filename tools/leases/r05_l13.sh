#!/bin/bash
# round 5, lease 13: rocprofv3 evidence on the final binary: kernel trace of one clip alone (serial), kernel trace of the
# pipelined bench (per-queue busy time), counter passes (FETCH / WRITE / MFMA busy) over one forward per batch shape
O=gpurun_out/r05o; mkdir -p $O
R=$PWD
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
cd /tmp; timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_r05s -o kt --output-format csv -- python $R/bench.py --plan serial --steps 1 --warmup 1 --no-cpu-baseline --no-batched --no-extras > $R/$O/kt_serial.json 2> $R/$O/kt_serial.err; echo "serial kernel trace rc=$? $(date +%T)"; cd $R
KT=$(find gpurun_out/kt_r05s -name "kt_kernel_trace.csv" | head -1); ST=$(find gpurun_out/kt_r05s -name "kt_kernel_stats.csv" | head -1)
python tools/trace_segments.py $KT > $O/kernel_trace_serial.md 2> $O/trace_segments.err; cp $ST $O/rocprofv3_kernel_stats_serial.csv; head -30 $O/kernel_trace_serial.md
rm -rf gpurun_out/kt_r05s
cd /tmp; timeout 500 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_r05p -o kt --output-format csv -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-batched --no-extras > $R/$O/kt_pipeline.json 2> $R/$O/kt_pipeline.err; echo "pipeline kernel trace rc=$? $(date +%T)"; cd $R
KT=$(find gpurun_out/kt_r05p -name "kt_kernel_trace.csv" | head -1); ST=$(find gpurun_out/kt_r05p -name "kt_kernel_stats.csv" | head -1)
python tools/trace_overlap.py $KT > $O/kernel_trace_pipeline.md 2> $O/trace_overlap.err; cp $ST $O/rocprofv3_kernel_stats_pipeline.csv; head -30 $O/kernel_trace_pipeline.md; tail -c 600 $O/kt_pipeline.json
rm -rf gpurun_out/kt_r05p
bash tools/gpu_pmc.sh pmc_r05; echo "pmc done $(date +%T)"
python tools/pmc_summary.py gpurun_out/pmc_r05 > $O/pmc_summary_raw.md 2> $O/pmc_summary.err; tail -5 $O/pmc_summary_raw.md; grep "forward done\|arith" gpurun_out/pmc_r05_f.log

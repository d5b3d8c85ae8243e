#!/bin/bash
# round 5, lease 19: kernel trace of the PIPELINED bench (edit lanes launched eagerly: rocprofv3's hipGraphLaunch interception
# segfaulted in the multi-threaded graph run, lease 13), then the whole GPU suite on the final tree
O=gpurun_out/r05u; mkdir -p $O
R=$PWD
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
cd /tmp; timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_r05p -o kt --output-format csv -- python $R/bench.py --steps 6 --warmup 2 --lane-launch eager --no-cpu-baseline --no-batched --no-extras > $R/$O/kt_pipeline.json 2> $R/$O/kt_pipeline.err; echo "pipeline kernel trace rc=$? $(date +%T)"; cd $R
KT=$(find gpurun_out/kt_r05p -name "kt_kernel_trace.csv" | head -1); ST=$(find gpurun_out/kt_r05p -name "kt_kernel_stats.csv" | head -1)
if [ -n "$KT" ]; then python tools/trace_overlap.py $KT > $O/kernel_trace_pipeline.md 2> $O/trace_overlap.err; cp $ST $O/rocprofv3_kernel_stats_pipeline.csv; head -40 $O/kernel_trace_pipeline.md; fi
tail -c 300 $O/kt_pipeline.json
rm -rf gpurun_out/kt_r05p
timeout 900 python -m pytest -q -m gpu tests > $O/gpu_suite.log 2>&1; echo "suite rc=$? $(date +%T)"; tail -25 $O/gpu_suite.log

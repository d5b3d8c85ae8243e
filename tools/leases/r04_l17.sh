#!/bin/bash
# Round 4, lease 17: rocprofv3 kernel trace of one clip alone (--plan serial) on the FINAL binary (split-bf16 attention included)
O=gpurun_out/r04r; mkdir -p $O
R=$PWD
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
cd /tmp; timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_r04f -o kt --output-format csv -- python $R/bench.py --plan serial --steps 1 --warmup 1 --no-cpu-baseline --no-batched --no-extras > $R/$O/kt_serial.json 2> $R/$O/kt_serial.err; echo "kernel trace rc=$? $(date +%T)"; cd $R
KT=$(find gpurun_out/kt_r04f -name "kt_kernel_trace.csv" | head -1); ST=$(find gpurun_out/kt_r04f -name "kt_kernel_stats.csv" | head -1)
python tools/trace_segments.py $KT > $O/kernel_trace_serial.md 2> $O/trace_segments.err; cp $ST $O/rocprofv3_kernel_stats_serial.csv; head -36 $O/kernel_trace_serial.md
rm -rf gpurun_out/kt_r04f

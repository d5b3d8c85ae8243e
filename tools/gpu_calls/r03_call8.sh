#!/bin/bash
# r03 call 8: codec on a partition (why 4x), kernel trace of the pipelined bench (both kernel classes resident at once), config 5.
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
timeout 400 python tools/codec_partition.py > gpurun_out/r03_codec_partition.log 2>&1; echo "codec rc=$?"; grep -v "amdgpu.ids\|WARNING" gpurun_out/r03_codec_partition.log | tail -12 | cut -c1-400
rm -rf gpurun_out/kt_r03; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/kt_r03 -o kt --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-extras --no-cpu-baseline --no-batched > $GRAFT_REPO_ROOT/gpurun_out/r03_kt_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r03_kt_bench.err; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
grep "^\[bench\]" gpurun_out/r03_kt_bench.err | tail -3 | cut -c1-400
TR=$(find gpurun_out/kt_r03 -name "*kernel_trace.csv" | head -1); ST=$(find gpurun_out/kt_r03 -name "*kernel_stats.csv" | head -1)
python tools/trace_overlap.py $TR > gpurun_out/r03_kernel_trace_pipeline.md 2> gpurun_out/r03_trace_overlap.err; echo "overlap rc=$?"; tail -12 gpurun_out/r03_kernel_trace_pipeline.md | cut -c1-500
cp $ST gpurun_out/r03_rocprofv3_kernel_stats_pipeline.csv; head -8 $ST | cut -c1-200
rm -f $TR   # the raw trace is tens of MB; the summaries travel back
timeout 400 python tools/bench_stable_audio.py --steps 1 --warmup 1 > gpurun_out/r03_config5.json 2> gpurun_out/r03_config5.err; echo "config5 rc=$?"; cut -c1-600 gpurun_out/r03_config5.json; tail -2 gpurun_out/r03_config5.err

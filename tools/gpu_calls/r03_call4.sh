#!/bin/bash
# r03 call 4: lanes measured with wall clock + execution counters (graph vs eager, 1 host thread vs one per lane); pipeline tests
# on the pruned library; lanes bench with eager and graph steps.
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/lanes2.py 8 40 > gpurun_out/r03_lanes2.log 2>&1; echo "lanes2 rc=$?"; grep -v "amdgpu.ids" gpurun_out/r03_lanes2.log | tail -60
timeout 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_kernels.py tests/test_gpu_unet.py -x -q > gpurun_out/r03_c4_tests.log 2>&1; echo "tests rc=$?"; tail -8 gpurun_out/r03_c4_tests.log
for mode in eager graph; do
timeout 400 python bench.py --steps 8 --warmup 4 --lanes 4 --lane-launch $mode --no-extras --no-cpu-baseline > gpurun_out/r03_bench_lanes4_$mode.json 2> gpurun_out/r03_bench_lanes4_$mode.err; echo "bench $mode rc=$?"; grep "^\[bench\]" gpurun_out/r03_bench_lanes4_$mode.err | tail -8
done

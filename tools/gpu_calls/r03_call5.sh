#!/bin/bash
# r03 call 5: partition-pipeline GPU tests, (edit_cus, edit_lanes) sweep on the headline workload, one bench line.
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_pipeline.py -x -q > gpurun_out/r03_c5_tests.log 2>&1; echo "tests rc=$?"; tail -12 gpurun_out/r03_c5_tests.log
timeout 900 python tools/pipeline_sweep.py 8 128:1 112:1 120:2 112:2 104:2 > gpurun_out/r03_pipeline_sweep.log 2>&1; echo "sweep rc=$?"; grep "^{" gpurun_out/r03_pipeline_sweep.log; grep -i "error\|Traceback" -A8 gpurun_out/r03_pipeline_sweep.log | head -30
timeout 600 python bench.py --steps 8 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r03_bench_partition.json 2> gpurun_out/r03_bench_partition.err; echo "bench rc=$?"; grep "^\[bench\]" gpurun_out/r03_bench_partition.err | tail -10

#!/bin/bash
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_loops.py tests/test_gpu_e2e.py tests/test_gpu_pc.py "tests/test_gpu_unet.py::test_full_audioldm_s_unet_matches_oracle" "tests/test_gpu_codec.py::test_vocoder_ten_second_clip_matches_oracle" "tests/test_gpu_kernels.py::test_lin_gemm_epilogues" -m gpu -x -q > gpurun_out/c6_tests.log 2>&1; echo "tests rc=$?"
tail -4 gpurun_out/c6_tests.log
timeout 600 python tools/unet_profile.py 2 "" late=1 merge=0 > gpurun_out/c6_prof_B2.log 2>&1; echo "prof2 rc=$?"
grep "^\[" gpurun_out/c6_prof_B2.log
timeout 600 python tools/unet_profile.py 40 "" merge=0 > gpurun_out/c6_prof_B40.log 2>&1; echo "prof40 rc=$?"
grep "^\[" gpurun_out/c6_prof_B40.log
timeout 600 python bench.py --clips-per-gpu 8 --steps 1 --warmup 1 --no-cpu-baseline --no-batched > gpurun_out/bench_r02_cpg8.json 2> gpurun_out/bench_r02_cpg8.err; echo "bench cpg8 rc=$?"
head -c 600 gpurun_out/bench_r02_cpg8.json; echo; tail -3 gpurun_out/bench_r02_cpg8.err

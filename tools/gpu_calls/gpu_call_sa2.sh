#!/bin/bash
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_stable_audio.py -m gpu -q > gpurun_out/sa2_tests.log 2>&1; echo "sa tests rc=$?"
tail -25 gpurun_out/sa2_tests.log
timeout 900 python tools/bench_stable_audio.py --steps 1 --warmup 1 > gpurun_out/bench_sa_r02.json 2> gpurun_out/bench_sa_r02.err; echo "sa bench rc=$?"
tail -3 gpurun_out/bench_sa_r02.err; cat gpurun_out/bench_sa_r02.json
timeout 400 python tools/sa_profile.py 40 > gpurun_out/sa_prof_B40.log 2>&1; echo "prof B40 rc=$?"; head -14 gpurun_out/sa_prof_B40.log

#!/bin/bash
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python tools/bench_stable_audio.py --steps 1 --warmup 1 > gpurun_out/bench_sa_r02b.json 2> gpurun_out/bench_sa_r02b.err; echo "sa bench rc=$?"
cat gpurun_out/bench_sa_r02b.json

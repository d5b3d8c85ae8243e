#!/bin/bash
# r03 call 10: the round's evidence, part 1 -- complete bench line and the PMC passes on the final GEMM sources.
export PYTHONPATH=$PWD TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
timeout 1200 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_r03.json 2> gpurun_out/bench_r03.err; echo "bench rc=$?"; grep "^\[bench\]" gpurun_out/bench_r03.err | cut -c1-420; grep -v "^\[bench\]\|WARNING\|amdgpu" gpurun_out/bench_r03.err | tail -6
bash tools/gpu_pmc.sh pmc_r03 2>&1 | tail -5
ALG=$(grep "forward done" gpurun_out/pmc_r03_f.log | awk '{s+=$NF} END {print s}')
python tools/pmc_summary.py gpurun_out/pmc_r03 --json gpurun_out/r03_pmc_forward.json --alg-total-bytes $ALG > gpurun_out/r03_pmc_forward.md 2> gpurun_out/r03_pmc_summary.err; echo "pmc summary rc=$? alg=$ALG"; tail -3 gpurun_out/r03_pmc_forward.md | cut -c1-300; head -12 gpurun_out/r03_pmc_forward.json
find gpurun_out/pmc_r03 -name "*.csv" -size +2M -delete

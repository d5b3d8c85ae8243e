#!/bin/bash
# r03 call 11: the round's evidence, part 2 -- the full GPU suite on the final tree.
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1400 python -m pytest tests -m gpu -q > gpurun_out/r03_gpu_suite.log 2>&1; echo "suite rc=$?"; tail -25 gpurun_out/r03_gpu_suite.log | cut -c1-200

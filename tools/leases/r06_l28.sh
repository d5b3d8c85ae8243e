#!/bin/bash
# round 6, lease 28: whole GPU suite on the grouped tile order (every split-bf16 GEMM takes it), with the 100 slowest tests listed
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06ab; mkdir -p $O
timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=100 > $O/suite.log 2>&1; echo "suite rc=$?"; tail -3 $O/suite.log

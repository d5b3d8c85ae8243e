#!/bin/bash
# round 6, lease 25: per-op profile of the batch-2 forward as a 64-CU edit lane builds it (share=2 like the edit loop) -- the
# baseline for the launch-cutting work (GroupNorm / split-K reduce)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06y; mkdir -p $O
PYTHONPATH=. timeout 300 python tools/lane_perop.py 64 bf16x6 share=2 > $O/lane_perop_cus64_share2.json 2> $O/lane_perop.err; echo "rc=$?"; tail -3 $O/lane_perop.err
PYTHONPATH=. timeout 300 python tools/lane_perop.py 64 bf16x6 > $O/lane_perop_cus64.json 2>> $O/lane_perop.err; echo "rc=$?"

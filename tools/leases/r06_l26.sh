#!/bin/bash
# round 6, lease 26: split-K finished by the last-arriving block (flag bit 9): kernel tests (bit-identity with the reduce launch,
# header reuse, co-residency), the engine-level co-residency test, and the lane step A/B (fuse=0 / fuse=1 in one lease)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06z; mkdir -p $O
timeout 900 python -m pytest -q -m gpu -x tests/test_gpu_splitk_fused.py tests/test_gpu_coresidency.py > $O/tests.log 2>&1; echo "tests rc=$?"; tail -25 $O/tests.log
for f in 0 1 0 1; do
  PYTHONPATH=. timeout 300 python tools/lane_perop.py 64 bf16x6 share=2 fuse=$f > $O/lane_fuse${f}_$RANDOM.json 2>> $O/lane.err; echo "fuse=$f rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06z/lane_fuse*.json')):
    d=json.load(open(f)); print(f, d['fuse_splitk'], d['ops'], 'sum', d['per_op_sum_ms'], 'graph', d['graph_replay_ms'])
PY

#!/bin/bash
# round 6, lease 4: source bisect of the gather loader (no mask multiply / no loader activation / no timeline stamps / mask pinned)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06d; mkdir -p $O
for v in 20 21 22 24; do echo "=== v$v"; PYTHONPATH=. timeout 200 python tools/diag/lin_gather_stress.py cases=one stress=x6 R=60 lib=scratch/libaed_v$v.so > $O/v$v.log 2>&1; grep -v "WARNING\|amdgpu.ids" $O/v$v.log | cut -c1-300 | head; done

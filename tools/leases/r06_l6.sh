#!/bin/bash
# round 6, lease 6: ds_write_b128 data registers rewritten by the next VALU op: s_nop 1 / s_nop 7 / lgkmcnt(0) after each A-tile write
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06f; mkdir -p $O
for v in 40 41 42; do echo "=== v$v"; PYTHONPATH=. timeout 200 python tools/diag/lin_gather_stress.py cases=one stress=x6 R=400 lib=scratch/libaed_v$v.so 2>&1 | grep -v "WARNING\|amdgpu.ids" | cut -c1-300 | tee $O/v$v.log; done

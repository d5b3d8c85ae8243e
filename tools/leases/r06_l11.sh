#!/bin/bash
# round 6, lease 11: v_pk_mul_f32 written in place over the register pair whose LOW half it broadcasts (75) against a separate destination (76)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06k; mkdir -p $O
for v in 75 76; do echo "=== v$v"; PYTHONPATH=. timeout 200 python tools/diag/lin_gather_stress.py cases=one stress=x6 R=400 lib=scratch/libaed_v$v.so 2>&1 | grep -v "WARNING\|amdgpu.ids" | cut -c1-300 | tee $O/v$v.log; done

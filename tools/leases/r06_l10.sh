#!/bin/bash
# round 6, lease 10: same code position, same scheduling fence: v_pk_mul_f32 (with / without op_sel_hi) against four v_mul_f32
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06j; mkdir -p $O
for v in 72 73 74; do echo "=== v$v"; PYTHONPATH=. timeout 200 python tools/diag/lin_gather_stress.py cases=one stress=x6 R=400 lib=scratch/libaed_v$v.so 2>&1 | grep -v "WARNING\|amdgpu.ids" | cut -c1-300 | tee $O/v$v.log; done

#!/bin/bash
# round 6, lease 9: without v_pk_mul_f32 in the loader (bitwise zeroing, with and without the activation branches; four scalar v_mul_f32)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06i; mkdir -p $O
for v in 70 71 72; do echo "=== v$v"; PYTHONPATH=. timeout 200 python tools/diag/lin_gather_stress.py cases=one stress=x6 R=400 lib=scratch/libaed_v$v.so 2>&1 | grep -v "WARNING\|amdgpu.ids" | cut -c1-300 | tee $O/v$v.log; done

#!/bin/bash
# round 6, lease 12: a 32-bit VALU write consumed by v_pk_mul_f32 0 / 1 / 2 wait states later (inline asm, nothing else changed)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06l; mkdir -p $O
for v in 77 78 79; do echo "=== v$v"; PYTHONPATH=. timeout 200 python tools/diag/lin_gather_stress.py cases=one stress=x6 R=400 lib=scratch/libaed_v$v.so 2>&1 | grep -v "WARNING\|amdgpu.ids" | cut -c1-300 | tee $O/v$v.log; done

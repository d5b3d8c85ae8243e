#!/bin/bash
# round 6, lease 7: do the MFMA fragment registers reused by the next VALU op matter?  v50: 1000 cycles of s_nop after every MFMA
# chunk; v51: the fragment registers kept allocated across the next stage write
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06g; mkdir -p $O
for v in 50 51; do echo "=== v$v"; PYTHONPATH=. timeout 200 python tools/diag/lin_gather_stress.py cases=one stress=x6 R=400 lib=scratch/libaed_v$v.so 2>&1 | grep -v "WARNING\|amdgpu.ids" | cut -c1-300 | tee $O/v$v.log; done

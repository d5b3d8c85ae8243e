#!/bin/bash
# round 6, lease 3: which partial tile values does the finishing thread read differently under co-residency (LIN_DIAG=11 dump)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06c; mkdir -p $O
PYTHONPATH=. timeout 300 python tools/diag/lin_gather_stress.py cases=head stress=x6 R=40 dump=1 lib=scratch/libaed_v11.so > $O/dump.log 2>&1; grep -v "WARNING\|amdgpu.ids" $O/dump.log | cut -c1-700 | head -80

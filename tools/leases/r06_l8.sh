#!/bin/bash
# round 6, lease 8: per lane / loader row checksums of raw loads, masks and staged values (LIN_DIAG=60) solo vs under the stressor
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06h; mkdir -p $O
PYTHONPATH=. timeout 200 python tools/diag/lin_gather_stress.py cases=one stress=x6 R=60 dump=2 lib=scratch/libaed_v60.so 2>&1 | grep -v "WARNING\|amdgpu.ids" | cut -c1-400 | tee $O/v60.log

#!/bin/bash
# round 6, lease 22: census of the OTHER kernel families as victims of the split-bf16 stressor (LDS-staged GEMMs in both arithmetics,
# GEGLU lin tiles, attention variants, GroupNorm kernels): does anything else depend on co-resident workgroups?
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06v; mkdir -p $O
PYTHONPATH=. timeout 600 python tools/diag/lin_gather_stress.py cases=victims R=200 2>&1 | grep -v "WARNING\|amdgpu.ids" | cut -c1-300 | tee $O/victims.log

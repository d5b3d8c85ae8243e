#!/bin/bash
# round 6, lease 5: the VGPR-only gather mask: standalone census under the split-bf16 stressor (old form beside it), then the
# CFG-shared batch-2 edit engine stepwise under VAE encodes (tools/diag/share_edit_bisect.py), graph and eager
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06e; mkdir -p $O
echo "=== old mask form (LIN_DIAG=30)"; PYTHONPATH=. timeout 200 python tools/diag/lin_gather_stress.py cases=one stress=x6 R=100 lib=scratch/libaed_v30.so 2>&1 | grep -v "WARNING\|amdgpu.ids" | cut -c1-300 | tee $O/old.log
echo "=== new mask form"; PYTHONPATH=. timeout 200 python tools/diag/lin_gather_stress.py cases=one stress=x6 R=400 2>&1 | grep -v "WARNING\|amdgpu.ids" | cut -c1-300 | tee $O/new_one.log
PYTHONPATH=. timeout 300 python tools/diag/lin_gather_stress.py cases=census stress=x6 R=100 2>&1 | grep -v "WARNING\|amdgpu.ids" | cut -c1-300 | tee $O/new_census.log | grep -c "perturbed launches 0 of"
grep -v "perturbed launches 0 of" $O/new_census.log
PYTHONPATH=. timeout 200 python tools/diag/lin_gather_stress.py cases=head stress=x6 R=200 2>&1 | grep -v "WARNING\|amdgpu.ids" | cut -c1-300 | tee $O/new_head.log
echo "=== engine, graph"; PYTHONPATH=. timeout 300 python tools/diag/share_edit_bisect.py N=16 2>&1 | grep -v "WARNING\|amdgpu.ids" | cut -c1-400 | tail -12 | tee $O/engine_graph.log
echo "=== engine, eager"; PYTHONPATH=. timeout 300 python tools/diag/share_edit_bisect.py N=8 launch=eager 2>&1 | grep -v "WARNING\|amdgpu.ids" | cut -c1-400 | tail -8 | tee $O/engine_eager.log

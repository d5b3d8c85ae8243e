#!/bin/bash
# round 6, lease 13: the shipped gather loader (fenced bitwise zero padding) against the round-5 form: standalone census under the
# split-bf16 stressor, then the CFG-shared batch-2 edit engine stepwise under VAE encodes (graph and eager)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06m; mkdir -p $O
echo "=== round-5 form (LIN_GATHER_R5_FORM)"; PYTHONPATH=. timeout 200 python tools/diag/lin_gather_stress.py cases=one stress=x6 R=200 lib=scratch/libaed_r5form.so 2>&1 | grep -v "WARNING\|amdgpu.ids" | cut -c1-300 | tee $O/r5form.log
echo "=== shipped"; PYTHONPATH=. timeout 300 python tools/diag/lin_gather_stress.py cases=one stress=x6 R=1000 2>&1 | grep -v "WARNING\|amdgpu.ids" | cut -c1-300 | tee $O/one.log
PYTHONPATH=. timeout 300 python tools/diag/lin_gather_stress.py cases=head stress=x6 R=400 2>&1 | grep -v "WARNING\|amdgpu.ids" | cut -c1-300 | tee $O/head.log
PYTHONPATH=. timeout 600 python tools/diag/lin_gather_stress.py cases=census stress=x6 R=200 2>&1 | grep -v "WARNING\|amdgpu.ids" | cut -c1-300 > $O/census.log; echo "census: clean cases $(grep -c 'perturbed launches 0 of' $O/census.log) of $(grep -c 'perturbed launches' $O/census.log)"; grep -v "perturbed launches 0 of" $O/census.log
echo "=== engine, graph"; PYTHONPATH=. timeout 400 python tools/diag/share_edit_bisect.py N=24 2>&1 | grep -v "WARNING\|amdgpu.ids" | cut -c1-400 | tail -8 | tee $O/engine_graph.log
echo "=== engine, eager"; PYTHONPATH=. timeout 400 python tools/diag/share_edit_bisect.py N=12 launch=eager 2>&1 | grep -v "WARNING\|amdgpu.ids" | cut -c1-400 | tail -5 | tee $O/engine_eager.log

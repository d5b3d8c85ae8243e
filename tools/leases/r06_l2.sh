#!/bin/bash
# round 6, lease 2: lin_gemm under CU co-residency outside any engine: the head's shapes under three stressors, four kernel
# variants (s_nop padding before the partial write / extra barrier / vmcnt(0) before every stage write / padded LDS), census
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06b; mkdir -p $O
run() { tag=$1; shift; echo "=== $tag"; PYTHONPATH=. timeout 300 python tools/diag/lin_gather_stress.py "$@" > $O/$tag.log 2>&1; grep -v "WARNING\|amdgpu.ids" $O/$tag.log | cut -c1-400 | tail -60; }
run head cases=head stress=x6,f32,copy,none R=60
for v in 1 2 3 5; do run head_v$v cases=head stress=x6 R=60 lib=scratch/libaed_v$v.so; done
run census cases=census stress=x6 R=40

#!/bin/bash
# round 6, lease 1: name the first perturbed node of the CFG-shared batch-2 edit engine (stepwise replays, per-buffer diffs) and
# discriminate kernel vs runtime: graph packet capture off, shared head's lin_gemm gather convs on x6 tiles, eager, masked stressor
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06a; mkdir -p $O
run() { tag=$1; shift; echo "=== $tag"; env "${ENVV[@]}" PYTHONPATH=. timeout 300 python tools/diag/share_edit_bisect.py "$@" > $O/$tag.log 2>&1; grep -v "WARNING\|amdgpu.ids" $O/$tag.log | cut -c1-600 | tail -40; }
ENVV=(); run base N=8
ENVV=(DEBUG_CLR_GRAPH_PACKET_CAPTURE=0); run nocapture N=8
ENVV=(); run head_x6 N=8 head=x6
run eager N=8 launch=eager
run masked N=8 side=masked
run noshare N=6 share=0

#!/bin/bash
# round 6, lease 31: persistent tile walk of conv_gemm_x6 (one tile per workgroup = flag 0x4000 vs the walk) on the batch-200 records,
# whole chip and 128-CU partition; the feature cases; the K sweep of the short-K Linears
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06ae; mkdir -p $O
X=./audioeditingcode_amd/x6_bench
timeout 120 $X 1 cases > $O/cases.log 2>&1; echo "feature cases rc=$?"; tail -1 $O/cases.log
timeout 300 $X 5 replay profiles/unet_b200_share2_gemm_ops.txt ab=16384:0 > $O/persist_chip.jsonl 2> $O/persist_chip.err; echo "chip rc=$? $(tail -1 $O/persist_chip.jsonl | cut -c1-250)"
timeout 300 $X 5 replay profiles/unet_b200_cus128_share2_gemm_ops.txt cus=128 ab=16384:0 > $O/persist_cus128.jsonl 2> $O/persist_cus128.err; echo "cus128 rc=$? $(tail -1 $O/persist_cus128.jsonl | cut -c1-250)"
timeout 300 $X 5 replay profiles/r06_ksweep_ops.txt ab=16384:0 > $O/ksweep.jsonl 2> $O/ksweep.err; grep '"op"' $O/ksweep.jsonl | cut -c1-200

#!/bin/bash
# round 6, lease 23: tile sweep of the batch-1 shapes (the CFG-shared head of the batch-2 edit engine runs at batch 1: M = 4096 / 1024
# rows took rule-based lin tiles) on 64- and 128-CU streams, split-bf16 + split-K candidates
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06w; mkdir -p $O
X=./audioeditingcode_amd/x6_bench
for c in 64 128; do
  timeout 400 $X 12 sweep profiles/unet_b1_gemm_ops.txt cus=$c x6 > $O/sweep_B1_cus${c}_x6.json 2> $O/sweep_B1_cus${c}_x6.err; echo "sweep B1 cus$c rc=$? $(date +%T)"; tail -2 $O/sweep_B1_cus${c}_x6.err
done
ls -la $O

#!/bin/bash
# round 6, lease 40: re-sweep of the edit lane's GEMM shapes (batch 2 and the batch-1 head, 64-CU stream) on the round's final kernels
# (the simple-rows epilogue made the unsplit form cheaper: split-K choices may move)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06an; mkdir -p $O
X=./audioeditingcode_amd/x6_bench
timeout 500 $X 20 sweep profiles/unet_b2_gemm_ops.txt cus=64 x6 > $O/sweep_B2_cus64_x6.json 2> $O/sweep_B2.err; echo "sweep B2 rc=$? $(date +%T)"; tail -1 $O/sweep_B2.err
timeout 500 $X 20 sweep profiles/unet_b1_gemm_ops.txt cus=64 x6 > $O/sweep_B1_cus64_x6.json 2> $O/sweep_B1.err; echo "sweep B1 rc=$? $(date +%T)"; tail -1 $O/sweep_B1.err
ls -la $O

#!/bin/bash
# round 6, lease 33: persistent walk sized for the stream's CU budget (flag bits 16-17) on the 128-CU partition; both changes on the lane
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06ag; mkdir -p $O
X=./audioeditingcode_amd/x6_bench
timeout 300 $X 5 replay profiles/unet_b200_cus128_share2_gemm_ops.txt cus=128 ab=16384:0 > $O/persist_cus128.jsonl 2> $O/persist_cus128.err; echo "cus128 walk A/B rc=$? $(tail -1 $O/persist_cus128.jsonl | cut -c1-250)"
timeout 300 $X 5 replay profiles/unet_b200_cus128_share2_gemm_ops.txt cus=128 ab=49152:0 > $O/both_cus128.jsonl 2> $O/both_cus128.err; echo "cus128 both rc=$? $(tail -1 $O/both_cus128.jsonl | cut -c1-250)"
timeout 300 $X 5 replay profiles/unet_b200_share2_gemm_ops.txt ab=49152:0 > $O/both_chip.jsonl 2> $O/both_chip.err; echo "chip both rc=$? $(tail -1 $O/both_chip.jsonl | cut -c1-250)"

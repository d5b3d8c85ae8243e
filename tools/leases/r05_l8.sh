#!/bin/bash
# round 5, lease 8: split-K candidates in the batch-2 sweeps (64-CU lane, whole chip), the vectorised split-K reduce
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05h; mkdir -p $O
X=./audioeditingcode_amd/x6_bench
timeout 60 $X 2 cases > $O/x6_cases.log 2>&1; echo "x6 cases rc=$? (the split-K cases go through the new reduce)"; grep -c '"pass": true' $O/x6_cases.log
timeout 300 $X 40 sweep profiles/unet_b2_gemm_ops.txt cus=64 x6 > $O/sweep_B2_cus64_x6_ks.json 2> $O/sweep_B2_cus64_x6_ks.err; echo "sweep cus64 rc=$? $(date +%T)"; tail -1 $O/sweep_B2_cus64_x6_ks.err
timeout 300 $X 40 sweep profiles/unet_b2_gemm_ops.txt x6 > $O/sweep_B2_x6_ks.json 2> $O/sweep_B2_x6_ks.err; echo "sweep whole chip rc=$? $(date +%T)"; tail -1 $O/sweep_B2_x6_ks.err
grep " 128 " $O/sweep_B2_cus64_x6_ks.err | head -40

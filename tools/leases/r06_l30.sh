#!/bin/bash
# round 6, lease 30: fixed cost per output tile of conv_gemm_x6 on the short-K Linears of the batch-200 forward: the same records with
# K = 32 ... 1024 (time = fixed + K x slope)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06ad; mkdir -p $O
timeout 300 ./audioeditingcode_amd/x6_bench 5 replay profiles/r06_ksweep_ops.txt x6only > $O/ksweep.jsonl 2> $O/ksweep.err; echo "rc=$?"; tail -3 $O/ksweep.err
grep '"op"' $O/ksweep.jsonl | cut -c1-200

#!/bin/bash
# round 5, lease 4: wide chunks (BK = 32) of conv_gemm_x6: parity cases + A/B on the batch-200 shapes
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05d
timeout 120 ./audioeditingcode_amd/x6_bench 1 wide > gpurun_out/r05d/wide_cases.jsonl 2> gpurun_out/r05d/wide_cases.err; echo "wide cases rc=$?"; tail -2 gpurun_out/r05d/wide_cases.err
grep -c '"pass": true' gpurun_out/r05d/wide_cases.jsonl; grep '"pass": false' gpurun_out/r05d/wide_cases.jsonl | head
timeout 300 ./audioeditingcode_amd/x6_bench 10 quick > gpurun_out/r05d/quick.jsonl 2> gpurun_out/r05d/quick.err; echo "quick rc=$?"
grep '"shape"' gpurun_out/r05d/quick.jsonl | python3 -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['shape'][:44].ljust(44), d['variant'].ljust(26), d['us'], d['tflops'], d.get('rel_l2_vs_fp32_kernel'))"

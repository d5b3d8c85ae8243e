#!/bin/bash
# round 6, lease 32: simple-rows epilogue of conv_gemm_x6 (flag 0x8000 = general epilogue) on the batch-200 records, the batch-2
# lane records and the K sweep; feature cases
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06af; mkdir -p $O
X=./audioeditingcode_amd/x6_bench
timeout 120 $X 1 cases > $O/cases.log 2>&1; echo "feature cases rc=$?"; tail -1 $O/cases.log
timeout 300 $X 5 replay profiles/unet_b200_share2_gemm_ops.txt ab=32768:0 > $O/epi_chip.jsonl 2> $O/epi_chip.err; echo "chip rc=$? $(tail -1 $O/epi_chip.jsonl | cut -c1-250)"
timeout 300 $X 5 replay profiles/unet_b200_share2_gemm_ops.txt ab=49152:0 > $O/epi_persist_chip.jsonl 2> $O/epi_persist_chip.err; echo "chip (both off vs both on) rc=$? $(tail -1 $O/epi_persist_chip.jsonl | cut -c1-250)"
timeout 300 $X 5 replay profiles/unet_b200_cus128_share2_gemm_ops.txt cus=128 ab=49152:16384 > $O/epi_cus128.jsonl 2> $O/epi_cus128.err; echo "cus128 (no walk; epilogue A/B) rc=$? $(tail -1 $O/epi_cus128.jsonl | cut -c1-250)"
timeout 300 $X 20 replay profiles/unet_b2_gemm_ops.txt cus=64 ab=32768:0 > $O/epi_b2_cus64.jsonl 2> $O/epi_b2.err; echo "b2 cus64 rc=$? $(tail -1 $O/epi_b2_cus64.jsonl | cut -c1-250)"
timeout 300 $X 5 replay profiles/r06_ksweep_ops.txt ab=32768:0 > $O/ksweep.jsonl 2> $O/ksweep.err; grep '"op"' $O/ksweep.jsonl | grep "K32\|K256" | cut -c1-200

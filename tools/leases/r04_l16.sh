#!/bin/bash
# Round 4, lease 16: counter passes on the FINAL GEMM sources (bench.py prints roofline.traffic only for a matching csrc_hash) + the
# tap-major K-order A/B that lease 6/7 lost to a script bug
O=gpurun_out/r04q; mkdir -p $O
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
bash tools/gpu_pmc.sh pmc_r04b ab; echo "pmc done $(date +%T)"
python tools/pmc_summary.py gpurun_out/pmc_r04b > $O/pmc_summary_raw.md 2> $O/pmc_summary.err
python tools/pmc_summary.py gpurun_out/pmc_r04b_tapmajor > $O/pmc_summary_tapmajor_raw.md 2>> $O/pmc_summary.err
tail -3 $O/pmc_summary_raw.md; tail -2 $O/pmc_summary_tapmajor_raw.md; grep "forward done\|arith" gpurun_out/pmc_r04b_f.log gpurun_out/pmc_r04b_tapmajor_f.log

#!/bin/bash
# Two rocprofv3 --pmc passes over the split-bf16 GEMM harness (two shapes: fp32 tile 1, x6 tiles 1 / 8, three-term diagnostic).
R=$PWD
TAG=${1:-x6_pmc}
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
cd /tmp
for pass in "s SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
            "l SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM"; do
  set -- $pass; tag=$1; shift
  timeout 60 rocprofv3 --pmc "$@" -d $R/gpurun_out/$TAG/$tag -o $tag --output-format csv -- $R/audioeditingcode_amd/x6_bench 3 pmc > $R/gpurun_out/${TAG}_$tag.log 2>&1; echo "pmc $tag rc=$?"
done
ls -R $R/gpurun_out/$TAG | head -20

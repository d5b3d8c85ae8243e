#!/bin/bash
# Three separate rocprofv3 --pmc passes over one U-Net forward at batch 200 and one at batch 2 (tools/pmc_forward.py).
export PYTHONPATH=$PWD TMPDIR=/tmp
R=$PWD
TAG=${1:-pmc_r02h}
mkdir -p gpurun_out
cd /tmp
for pass in "f FETCH_SIZE" "w WRITE_SIZE" "s SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  set -- $pass; tag=$1; shift
  timeout 300 rocprofv3 --pmc "$@" -d $R/gpurun_out/$TAG/$tag -o $tag --output-format csv -- python $R/tools/pmc_forward.py 200 2 > $R/gpurun_out/${TAG}_$tag.log 2>&1; echo "pmc $tag rc=$?"
done
cd $R; grep "forward done" gpurun_out/${TAG}_f.log

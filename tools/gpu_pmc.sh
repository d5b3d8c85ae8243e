#!/bin/bash
# Separate rocprofv3 --pmc passes over one U-Net forward at batch 200 and one at batch 2 (tools/pmc_forward.py), the product's
# arithmetic and K traversal; with a second argument "ab" also FETCH_SIZE of the batch-200 forward in the tap-major K order.
export PYTHONPATH=$PWD TMPDIR=/tmp
R=$PWD
TAG=${1:-pmc_r04}
mkdir -p gpurun_out
cd /tmp
AB=$2
for pass in "f FETCH_SIZE" "w WRITE_SIZE" "s SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  set -- $pass; tag=$1; shift
  timeout 300 rocprofv3 --pmc "$@" -d $R/gpurun_out/$TAG/$tag -o $tag --output-format csv -- python $R/tools/pmc_forward.py 200 2 > $R/gpurun_out/${TAG}_$tag.log 2>&1; echo "pmc $tag rc=$?"
done
if [ "$AB" = "ab" ]; then
  AED_PMC_TAPMAJOR=1 timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_tapmajor/f -o f --output-format csv -- python $R/tools/pmc_forward.py 200 > $R/gpurun_out/${TAG}_tapmajor_f.log 2>&1; echo "pmc tap-major f rc=$?"
fi
if [ "$AB" = "order" ]; then
  AED_PMC_NFASTEST=1 timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_nfastest/f -o f --output-format csv -- python $R/tools/pmc_forward.py 200 > $R/gpurun_out/${TAG}_nfastest_f.log 2>&1; echo "pmc n-fastest f rc=$?"
fi
cd $R; grep "forward done" gpurun_out/${TAG}_f.log

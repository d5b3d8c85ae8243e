#!/bin/bash
# r03 call 12: Stable Audio DiT forward (batch 2 and 40) under rocprofv3 -- kernel trace + stats, MFMA-busy and FETCH counters.
export PYTHONPATH=$PWD TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/sa_kt -o kt --output-format csv -- python $R/tools/sa_forward.py 2 40 > $R/gpurun_out/r03_sa_kt.log 2>&1; echo "kt rc=$?"; grep "forward done" $R/gpurun_out/r03_sa_kt.log
export AED_ONE=1
timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $R/gpurun_out/sa_pmc/s -o s --output-format csv -- python $R/tools/sa_forward.py 2 40 > $R/gpurun_out/r03_sa_pmc_s.log 2>&1; echo "pmc s rc=$?"
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/sa_pmc/f -o f --output-format csv -- python $R/tools/sa_forward.py 2 40 > $R/gpurun_out/r03_sa_pmc_f.log 2>&1; echo "pmc f rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/sa_pmc/w -o w --output-format csv -- python $R/tools/sa_forward.py 2 40 > $R/gpurun_out/r03_sa_pmc_w.log 2>&1; echo "pmc w rc=$?"
cd $R
ST=$(find gpurun_out/sa_kt -name "*kernel_stats.csv" | head -1); cp $ST gpurun_out/r03_sa_rocprofv3_kernel_stats.csv; head -14 $ST | cut -c1-160
python tools/pmc_summary.py gpurun_out/sa_pmc > gpurun_out/r03_sa_pmc.md 2> gpurun_out/r03_sa_pmc.err; echo "summary rc=$?"; head -16 gpurun_out/r03_sa_pmc.md | cut -c1-220
find gpurun_out/sa_kt gpurun_out/sa_pmc -name "*.csv" -size +2M -delete

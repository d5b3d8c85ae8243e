#!/bin/bash
# Round 4, lease 7 (re-run of lease 6, whose outputs were lost with the container): codec stage on the inversion queue; PMC passes (traffic, K-order A/B); serial kernel trace
O=gpurun_out/r04g; mkdir -p $O
R=$PWD
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
B="--steps 12 --warmup 2 --no-extras --no-cpu-baseline --no-batched"
timeout 300 python bench.py $B > $O/bench_l2_codecq.json 2> $O/bench_l2_codecq.err; echo "bench l2 codec-on-front-queue rc=$? $(date +%T)"
python - "$O/bench_l2_codecq.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get('roofline') or {}; p=d.get('pipeline') or {}
    print(sys.argv[1], 'value', round(d['value'],4), 'ms', round(d['ms_per_step'],1), 'path', r.get('path_frac'), p.get('device_ms'), 'lat', p.get('clip_latency_ms_avg'))
    print('   queues', p.get('queue_separation'))
except Exception as e: print(sys.argv[1], 'unreadable', e)
PY
bash tools/gpu_pmc.sh pmc_r04 ab; echo "pmc done $(date +%T)"
python tools/pmc_summary.py gpurun_out/pmc_r04 > $O/pmc_summary_raw.md 2> $O/pmc_summary.err
python tools/pmc_summary.py gpurun_out/pmc_r04_tapmajor > $O/pmc_summary_tapmajor_raw.md 2>> $O/pmc_summary.err
tail -4 $O/pmc_summary_raw.md; tail -3 $O/pmc_summary_tapmajor_raw.md; grep "forward done\|arith" gpurun_out/pmc_r04_f.log gpurun_out/pmc_r04_tapmajor_f.log
mkdir -p $O/pmc; cp -r gpurun_out/pmc_r04 gpurun_out/pmc_r04_tapmajor $O/pmc/ 2>/dev/null; du -sh $O/pmc
cd /tmp; timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_r04s -o kt --output-format csv -- python $R/bench.py --plan serial --steps 1 --warmup 1 --no-cpu-baseline --no-batched --no-extras > $R/$O/kt_serial.json 2> $R/$O/kt_serial.err; echo "kernel trace rc=$? $(date +%T)"; cd $R
KT=$(find gpurun_out/kt_r04s -name "kt_kernel_trace.csv" | head -1); ST=$(find gpurun_out/kt_r04s -name "kt_kernel_stats.csv" | head -1)
python tools/trace_segments.py $KT > $O/kernel_trace_serial.md 2> $O/trace_segments.err; cp $ST $O/rocprofv3_kernel_stats_serial.csv; head -30 $O/kernel_trace_serial.md
rm -rf gpurun_out/kt_r04s

#!/bin/bash
# round 6, lease 42: evidence for the FINAL tree (after the GEGLU epilogue and gn_apply grid changes): rocprofv3 kernel trace of one clip (serial plan), PMC passes over the batch-200 / batch-2
# forwards, then the driver's own bench invocation (defaults)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06ap; mkdir -p $O
R=$PWD
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
cd /tmp; timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_r06s -o kt --output-format csv -- python $R/bench.py --plan serial --steps 1 --warmup 1 --no-cpu-baseline --no-batched --no-extras > $R/$O/kt_serial.json 2> $R/$O/kt_serial.err; echo "serial kernel trace rc=$? $(date +%T)"; cd $R
KT=$(find gpurun_out/kt_r06s -name "kt_kernel_trace.csv" | head -1); ST=$(find gpurun_out/kt_r06s -name "kt_kernel_stats.csv" | head -1)
python tools/trace_segments.py $KT > $O/kernel_trace_serial.md 2> $O/trace_segments.err; cp $ST $O/rocprofv3_kernel_stats_serial.csv; head -24 $O/kernel_trace_serial.md
rm -rf gpurun_out/kt_r06s
bash tools/gpu_pmc.sh pmc_r06 order; echo "pmc done $(date +%T)"
ALG=$(grep "forward done" gpurun_out/pmc_r06_f.log | awk '{s += $NF} END {printf "%.0f", s}')
python tools/pmc_summary.py gpurun_out/pmc_r06 --json $O/r06_pmc_forward.json --alg-total-bytes $ALG > $O/r06_pmc_forward.md 2> $O/pmc_summary.err; tail -4 $O/r06_pmc_forward.md; cat $O/r06_pmc_forward.json; grep "forward done\|arith" gpurun_out/pmc_r06_f.log
cp $O/r06_pmc_forward.json profiles/r06_pmc_forward.json      # (so that the bench below finds the traffic of THIS tree; copied back by hand after the lease)
python - <<'PY' | tee $O/fetch_tile_order.txt
import csv,glob,collections
for tag in ('pmc_r06','pmc_r06_nfastest'):
    tot=collections.defaultdict(float); n=collections.Counter()
    for f in glob.glob(f'gpurun_out/{tag}/f/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            if row['Counter_Name']=='FETCH_SIZE':
                k=row['Kernel_Name'].split('(')[0][:40]; tot[k]+=float(row['Counter_Value']); n[k]+=1
    x6=sum(v for k,v in tot.items() if 'conv_gemm_x6' in k)
    print(tag, 'FETCH_SIZE of conv_gemm_x6 kernels: %.1f GB (x2-corrected %.1f GB) over %d launches' % (x6*1024/1e9, 2*x6*1024/1e9, sum(c for k,c in n.items() if 'conv_gemm_x6' in k)))
PY
find gpurun_out/pmc_r06 gpurun_out/pmc_r06_nfastest -name "*.csv" -size +3M -delete
( time timeout 1100 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.log; echo "k20 bench rc=$? $(date +%T)"
python - <<PY
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); p=d.get('pipeline',{}); r=d.get('roofline') or {}
print(round(d['value'],4), 'clips/s  ms/clip', round(d['ms_per_step'],1), 'steps', d['steps'], {k:round(v['avg'],1) for k,v in p.get('device_ms',{}).items()})
print('   roofline frac', r.get('frac'), 'achieved', r.get('achieved'), 'traffic', r.get('traffic'), 'part', (r.get('on_partition') or {}).get('frac'))
print('   edit_step', {k:(v['ms_per_step_as_graph'], v['launches']) for k,v in (r.get('edit_step') or {}).items()})
print('   single', d.get('value_single_clip_batched'), d.get('value_reference_order'), d.get('pipeline_vs_one_clip_at_a_time'))
print('   parity', d.get('parity'), 'cpu', {k: d['cpu_baseline'].get(k) for k in ('value','cores','kind')} if d.get('cpu_baseline') else None)
print('   subs', {k: (v.get('value') if isinstance(v, dict) else v) for k, v in d.items() if k.startswith('config')})
PY
tail -3 $O/bench_default.log

timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=25 > $O/suite.log 2>&1; echo "suite rc=$?"; tail -30 $O/suite.log

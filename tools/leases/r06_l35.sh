#!/bin/bash
# round 6, lease 35: branch-free erf in the GEGLU epilogue + division-free Linear prologue of conv_gemm_x6: K sweep, whole-forward
# replays, kernel / U-Net tests, bench A/B of the codec placement now that the edit lanes are the critical stage
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06ai; mkdir -p $O
X=./audioeditingcode_amd/x6_bench
timeout 120 $X 1 cases > $O/cases.log 2>&1; echo "feature cases rc=$?"; tail -1 $O/cases.log
timeout 300 $X 5 replay profiles/r06_ksweep_ops.txt x6only > $O/ksweep.jsonl 2> $O/ksweep.err; grep '"op"' $O/ksweep.jsonl | grep "K32\"\|K256\"" | cut -c1-140
timeout 300 $X 5 replay profiles/unet_b200_share2_gemm_ops.txt x6only > $O/chip.jsonl 2> $O/chip.err; echo "chip rc=$? $(tail -1 $O/chip.jsonl | cut -c60-200)"
timeout 300 $X 5 replay profiles/unet_b200_cus128_share2_gemm_ops.txt cus=128 x6only > $O/cus128.jsonl 2> $O/cus128.err; echo "cus128 rc=$? $(tail -1 $O/cus128.jsonl | cut -c60-200)"
timeout 900 python -m pytest -q -m gpu -x tests/test_gpu_kernels.py tests/test_gpu_unet.py tests/test_gpu_zz_split_bf16.py > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
for tag in "lane" "front --codec-queue front" "lane2"; do set -- $tag; t=$1; shift
  timeout 420 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline --no-batched "$@" > $O/bench_$t.json 2> $O/bench_$t.log; echo "bench $t rc=$?"
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$t.json').read().strip().splitlines()[-1]); p=d.get('pipeline',{}); r=d.get('roofline') or {}
    print('$t', round(d['value'],4), 'clips/s  ms/clip', round(d['ms_per_step'],1), {k:round(v['avg'],1) for k,v in p.get('device_ms',{}).items()})
    print('   part', (r.get('on_partition') or {}).get('frac'), 'edit_step', {k:(v['ms_per_step_as_graph'], v['launches']) for k,v in (r.get('edit_step') or {}).items()})
except Exception as e: print('ERR', e)
PY
done

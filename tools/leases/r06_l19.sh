#!/bin/bash
# round 6, lease 19: chain test + the files after it; no-split diagnostic again (private copies); bench A/B of the original's vocoding placement
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06s; mkdir -p $O
( time timeout 1500 python -m pytest -x -q -m gpu "tests/test_gpu_pc.py::test_config4_three_consecutive_drift_timesteps_at_full_size" tests/test_gpu_pipeline.py tests/test_gpu_stable_audio.py tests/test_gpu_unet.py tests/test_gpu_zz_split_bf16.py tests/test_gpu_zzz_fullsize_oracle_fixture.py -s ) > $O/tests.log 2>&1; echo "tests rc=$?"; grep -h "config 4 chain\|drifted trajectory\|config 5 at T=200\|passed\|failed\|^real\|Error" $O/tests.log | cut -c1-1500; tail -22 $O/tests.log | grep "s call"
bash tools/leases/r06_l17.sh
for tag in shipped "orig_on_lane --orig-on-lane"; do set -- $tag; t=$1; shift
  timeout 420 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline --no-batched "$@" > $O/bench_$t.json 2> $O/bench_$t.log; echo "bench $t rc=$?"
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$t.json').read().strip().splitlines()[-1]); p=d.get('pipeline',{}); r=d.get('roofline') or {}
    print('$t', round(d['value'],4), 'clips/s  ms/clip', round(d['ms_per_step'],1), {k:round(v['avg'],1) for k,v in p.get('device_ms',{}).items()})
except Exception as e: print('ERR', e)
PY
done
grep -i "error\|Traceback" $O/bench_*.log | head -5

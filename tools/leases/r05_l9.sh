#!/bin/bash
# round 5, lease 9: split-K tables in the engines: parity (full-size U-Net, pipeline vs alone), lane forward time, bench K = 20
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05i; mkdir -p $O
PYTHONPATH=. timeout 120 python tools/batch_scaling.py 64,256 2 > $O/batch_scaling.jsonl 2>/dev/null; cat $O/batch_scaling.jsonl
timeout 600 python -m pytest -q -m gpu -x tests/test_gpu_unet.py tests/test_gpu_pipeline.py::test_partition_pipeline_full_size_audioldm2_bit_identical_and_finite tests/test_gpu_zz_split_bf16.py::test_full_audioldm2_unet_in_split_bf16_matches_the_fp32_engine_and_the_oracle > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout 420 python bench.py --steps 20 --warmup 2 --no-extras --no-cpu-baseline > $O/bench.json 2> $O/bench.log; echo "bench rc=$?"
python - <<PY
import json
try:
    d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); p=d.get('pipeline',{}); r=d.get('roofline') or {}
    print(round(d['value'],4), 'ms/clip', round(d['ms_per_step'],1), {k:round(v['avg'],1) for k,v in p.get('device_ms',{}).items()})
    print('   roofline frac', r.get('frac'), 'achieved', r.get('achieved'), 'fp32eq', r.get('achieved_fp32_equiv'), 'part', (r.get('on_partition') or {}).get('frac'), 'path', (r.get('path') or {}))
    print('   edit_step', {k:(v['ms_per_step_as_graph'], v['ms_per_clip_step']) for k,v in (r.get('edit_step') or {}).items()})
    print('   single', d.get('value_single_clip_batched'), d.get('value_reference_order'), d.get('pipeline_vs_one_clip_at_a_time'))
    print('   fwd', {k:(round(v['ms'],2), v.get('frac')) for k,v in r['forward']['families'].items()})
except Exception as e: print('ERR', e)
PY
grep -i "error\|Traceback\|ROOF" $O/bench.log | head -5
timeout 420 python bench.py --steps 20 --warmup 2 --no-extras --no-cpu-baseline --no-batched --codec-queue chip > $O/bench_codec_chip.json 2> $O/bench_codec_chip.log; echo "bench codec chip rc=$?"
python - <<PY
import json
try:
    d=json.loads(open('$O/bench_codec_chip.json').read().strip().splitlines()[-1]); p=d.get('pipeline',{})
    print('codec on its own unmasked queue:', round(d['value'],4), 'ms/clip', round(d['ms_per_step'],1), {k:round(v['avg'],1) for k,v in p.get('device_ms',{}).items()})
except Exception as e: print('ERR', e)
PY

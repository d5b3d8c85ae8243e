#!/bin/bash
# round 6, lease 34: the simple-rows epilogue + grouped tile order tree (persistent walk removed): harness A/B on chip / partition / lane
# records, the kernel / U-Net / split-bf16 / codec / co-residency / pipeline tests, the lane step, a short bench
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06ah; mkdir -p $O
X=./audioeditingcode_amd/x6_bench
timeout 120 $X 1 cases > $O/cases.log 2>&1; echo "feature cases rc=$?"; tail -1 $O/cases.log
timeout 300 $X 5 replay profiles/unet_b200_share2_gemm_ops.txt ab=32768:0 > $O/epi_chip.jsonl 2> $O/epi_chip.err; echo "chip rc=$? $(tail -1 $O/epi_chip.jsonl | cut -c1-250)"
timeout 300 $X 5 replay profiles/unet_b200_cus128_share2_gemm_ops.txt cus=128 ab=32768:0 > $O/epi_cus128.jsonl 2> $O/epi_cus128.err; echo "cus128 rc=$? $(tail -1 $O/epi_cus128.jsonl | cut -c1-250)"
timeout 1500 python -m pytest -q -m gpu -x tests/test_gpu_kernels.py tests/test_gpu_unet.py tests/test_gpu_zz_split_bf16.py tests/test_gpu_codec.py tests/test_gpu_coresidency.py tests/test_gpu_pipeline.py tests/test_gpu_stable_audio.py > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
PYTHONPATH=. timeout 300 python tools/lane_perop.py 64 bf16x6 share=2 > $O/lane_perop_cus64_share2.json 2> $O/lane.err; python -c "
import json; d=json.load(open('$O/lane_perop_cus64_share2.json')); print('lane step', d['ops'], d['per_op_sum_ms'], d['graph_replay_ms'])"
timeout 420 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline --no-batched > $O/bench.json 2> $O/bench.log; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); p=d.get('pipeline',{}); r=d.get('roofline') or {}
print(round(d['value'],4), 'clips/s  ms/clip', round(d['ms_per_step'],1), {k:round(v['avg'],1) for k,v in p.get('device_ms',{}).items()})
print('   roofline frac', r.get('frac'), 'achieved', r.get('achieved'), 'part', (r.get('on_partition') or {}).get('frac'))
print('   edit_step', {k:(v['ms_per_step_as_graph'], v['launches']) for k,v in (r.get('edit_step') or {}).items()}, d.get('pipeline_vs_one_clip_at_a_time',{}).get('bit_identical_to_same_engines_alone'))
PY

#!/bin/bash
# Round 4, lease 3: lane interference (fixed tool, clocks), grouped-K-order A/B (time + correctness), conv / PC tests
O=gpurun_out/r04c; mkdir -p $O
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
X=./audioeditingcode_amd/x6_bench
timeout 120 $X 3 replay profiles/unet_b200_gemm_ops.txt korder > $O/korder_b200.jsonl 2> $O/korder_b200.err; echo "korder whole chip rc=$? $(date +%T)"; tail -1 $O/korder_b200.jsonl
timeout 120 $X 3 replay profiles/unet_b200_gemm_ops.txt cus=128 korder > $O/korder_b200_cus128.jsonl 2> $O/korder_b200_cus128.err; echo "korder cus128 rc=$? $(date +%T)"; tail -1 $O/korder_b200_cus128.jsonl
timeout 60 $X 2 cases > $O/x6_cases.log 2>&1; echo "x6 cases rc=$? $(date +%T)"
timeout 200 python tools/lane_interference.py > $O/interf.json 2> $O/interf.err; echo "interf rc=$? $(date +%T)"
timeout 200 python tools/lane_interference.py --front-arith f32 > $O/interf_frontf32.json 2> $O/interf_frontf32.err; echo "interf front f32 rc=$? $(date +%T)"
timeout 200 python tools/lane_interference.py --edit-lanes 1 > $O/interf_l1.json 2> $O/interf_l1.err; echo "interf l1 rc=$? $(date +%T)"
cat $O/interf.json $O/interf_frontf32.json $O/interf_l1.json
timeout 700 python -m pytest -m gpu -q -s -x tests/test_gpu_kernels.py tests/test_gpu_codec.py tests/test_gpu_unet.py tests/test_gpu_pc.py::test_full_size_power_iteration_vs_the_oracle_fixture tests/test_gpu_zz_split_bf16.py tests/test_gpu_pc.py::test_pc_clis_extract_pt_apply_on_the_gpu > $O/tests.log 2>&1; echo "tests rc=$? $(date +%T)"
grep -E "passed|failed|config 4|PC CLIs" $O/tests.log | tail -8

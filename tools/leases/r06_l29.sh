#!/bin/bash
# round 6, lease 29: write the committed oracle runs of the GPU suite's slowest CPU-oracle legs (tests/conftest.py oracle_run): the tests
# run their oracle live with AED_WRITE_ORACLE_RUNS set and store its result; then the same tests again from the stored runs
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06ac; mkdir -p $O
SEL="tests/test_gpu_pc.py::test_pc_clis_extract_pt_apply_on_the_gpu tests/test_gpu_loops.py::test_ddpm_inversion_and_edit_match_oracle tests/test_gpu_loops.py::test_ddim_baseline_matches_oracle tests/test_gpu_unet.py::test_full_tango_unet_matches_oracle tests/test_gpu_unet.py::test_full_audioldm_s_unet_matches_oracle tests/test_gpu_zz_split_bf16.py::test_full_audioldm2_unet_in_split_bf16_matches_the_fp32_engine_and_the_oracle tests/test_gpu_e2e.py::test_clip_edit_end_to_end_vs_oracle tests/test_gpu_e2e.py::test_two_prompt_segments_equal_and_unequal_tstart_on_the_gpu"
AED_WRITE_ORACLE_RUNS=$PWD/gpurun_out/oracle_runs timeout 900 python -m pytest -q -m gpu -x $SEL > $O/write.log 2>&1; echo "write rc=$?"; tail -14 $O/write.log
ls -la gpurun_out/oracle_runs
mkdir -p tests/golden/oracle_runs && cp gpurun_out/oracle_runs/*.pt tests/golden/oracle_runs/
timeout 900 python -m pytest -q -m gpu -x $SEL > $O/read.log 2>&1; echo "read rc=$?"; tail -14 $O/read.log

#!/bin/bash
# round 5, lease 15: why the full-size pipeline test differs from the clip alone with CFG row sharing; rest of the pipeline tests
O=gpurun_out/r05q; mkdir -p $O
timeout 400 python tools/diag/share_pipeline.py > $O/diag.log 2>&1; echo "diag rc=$?"; grep -v "^\[" $O/diag.log | tail -30
timeout 400 python -m pytest -q -m gpu tests/test_gpu_pipeline.py -k "group_plan or stealing or codec_on" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -12 $O/tests.log

#!/bin/bash
# round 5, lease 18: CFG row sharing in the inversion only: repeat-determinism tests, the full-size pipeline tests, bench K=20
O=gpurun_out/r05t; mkdir -p $O
timeout 600 python -m pytest -x -q -m gpu tests/test_gpu_pipeline.py -k "full_size" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -15 $O/tests.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batched --no-extras > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05t/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], {k: d["roofline"].get(k) for k in ("frac","achieved","peak","frac_fp32_equiv","launches_per_forward","avg_launch_us")})
PY

#!/bin/bash
# round 5, lease 22: last sanity run of bench.py on the final tree (short)
O=gpurun_out/r05x; mkdir -p $O
timeout 150 python bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline --no-batched > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05x/bench.json").read().strip().splitlines()[-1])
r=d["roofline"]
print(d["value"], d["ms_per_step"], {k: r.get(k) for k in ("frac","achieved","traffic","fp32_equiv_over_fp32_mfma_peak","failed")}, r["path"].get("fp32_equiv_over_fp32_mfma_peak"), r["path"].get("matrix_pipe_frac"))
PY
grep -n "FAILED\|Traceback" $O/bench.err | head -5

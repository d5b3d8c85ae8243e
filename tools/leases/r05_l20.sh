#!/bin/bash
# round 5, lease 20: the driver's invocation on the final tree
O=gpurun_out/r05v; mkdir -p $O
S=$(date +%s); timeout 560 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err; echo "bench rc=$? $(( $(date +%s) - S )) s"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05v/bench_k20.json").read().strip().splitlines()[-1])
r=d["roofline"]
print(d["value"], d["ms_per_step"], {k: r.get(k) for k in ("frac","achieved","peak","traffic","frac_fp32_equiv","launches_per_forward","avg_launch_us","failed")})
print({k: d.get(k) for k in ("cpu_baseline","single_clip","vs_baseline")})
print(json.dumps(d.get("parity"))[:600])
PY
grep -n "FAILED\|Traceback" $O/bench_k20.err | head

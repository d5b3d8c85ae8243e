"""Per-op profile of one Stable Audio Open DiT forward (full width, 24 layers, 1025-token sequence, 130-token context)
at batch B: hipGraph replay time, HIP-event pair per op, grouped by op name.

    PYTHONPATH=. python tools/sa_profile.py <B> [layers]"""
import collections
import json
import os
import sys
import time

import torch

from audioeditingcode_amd import configs, weights
from audioeditingcode_amd.stable_audio import DiTEngine, PackedDiTWeights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfg = dict(configs.FAMILIES["stable_audio"]["dit"])
if len(sys.argv) > 2:
    cfg["num_layers"] = int(sys.argv[2])
t0 = time.time()
sd = weights.random_state_dict(weights.dit_param_shapes(cfg), seed=0)
packed = PackedDiTWeights(sd, cfg, "cuda:0")
del sd
print(f"weights: {packed.nbytes() / 1e9:.2f} GB packed in {time.time() - t0:.1f} s", flush=True)
S = 130
eng = DiTEngine(cfg, packed, "cuda:0", B, S)
g = torch.Generator().manual_seed(1)
ctx = torch.randn(B, S, cfg["cross_attention_input_dim"], generator=g)
ctx[::2] = 0
eng.set_conditioning(ctx, torch.randn(B, cfg["global_states_input_dim"], generator=g))
eng.set_timestep(0.37)
eng.x_in.copy_(torch.randn(B, cfg["sample_size"], cfg["in_channels"], generator=g))
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for _ in range(2):
        eng.forward()
    st.synchronize()
    assert torch.isfinite(eng.v).all()
    eng.tape.capture()
    eng.tape.replay()
    st.synchronize()
    n = 10 if B <= 4 else 3
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(st)
    for _ in range(n):
        eng.tape.replay()
    ev1.record(st)
    st.synchronize()
    ms = ev0.elapsed_time(ev1) / n
    per = eng.tape.profile()
fl = eng.tape.flops
print(f"B={B} layers={cfg['num_layers']} ops={len(eng.tape.ops)} graph {ms:.3f} ms/forward -> {fl / ms / 1e9:.1f} TF/s "
      f"({fl / 1e12:.2f} TFLOP algorithmic, {fl / B / 1e12:.2f} per sample) = {fl / ms / 1e9 / 157.3:.3f} of the fp32 MFMA peak")
agg = collections.OrderedDict()
for m, t in zip(eng.tape.meta, per):
    key = m["name"].split(".", 1)[-1] if m["name"].startswith("b") else m["name"]
    a = agg.setdefault(key, [0, 0.0, 0.0])
    a[0] += 1
    a[1] += t
    a[2] += m["flops"]
print(f"  sum of per-op ms {sum(per):.3f}")
for k, (c, t, f) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"    {k:32s} n={c:4d} {t:9.3f} ms {f / max(t, 1e-9) / 1e9:8.1f} TF/s")
os.makedirs("gpurun_out", exist_ok=True)
json.dump(dict(B=B, layers=cfg["num_layers"], graph_ms=ms, tflops=fl / ms / 1e9,
               ops=[dict(name=m["name"], ms=t, flops=m["flops"]) for m, t in zip(eng.tape.meta, per)]),
          open(f"gpurun_out/sa_perop_B{B}.json", "w"))

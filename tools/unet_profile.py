"""Per-op profile of one AudioLDM2 U-Net forward at batch B for a list of build variants, in ONE process (shared
weights): eager time, hipGraph replay time, HIP-event pair per op, and a JSON dump of every op's shape + time.

    PYTHONPATH=. python tools/unet_profile.py <B> [variant ...]
variant = comma list of lin=0|1, two=0|1, late=0|1, merge=0|1, fold=0|1, attn=0|1, gn=0|1, gnreg=0|1  (default: the package
defaults; the tool flips the module constants of tape.py / unet.py), e.g.
    python tools/unet_profile.py 2 lin=0,two=0 "" attn=1 gn=1"""
import collections
import json
import os
import sys
import time

import torch

from audioeditingcode_amd import configs, tape as tape_mod, unet as unet_mod, weights
from audioeditingcode_amd.unet import PackedUNetWeights, UNetEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
variants = sys.argv[2:] or [""]
fam = configs.FAMILIES["audioldm2"]
sd = weights.random_state_dict(weights.unet_param_shapes(fam["unet"]), seed=0)
packed = PackedUNetWeights(sd, "cuda:0")
DEFAULTS = dict(lin=tape_mod.LIN_MODE, two=int(unet_mod.TWO_SOURCE),
                attn=tape_mod.ATTN_VARIANT, gn=tape_mod.GN_FORCE_SMALL, late=tape_mod.LATE_EPILOGUE,
                merge=int(unet_mod.MERGE_FF2_PROJ), fold=int(unet_mod.FOLD_XATTN), gnreg=1 - tape_mod.GN_VARIANT)
st = torch.cuda.Stream()
os.makedirs("gpurun_out", exist_ok=True)
for spec in variants:
    v = dict(DEFAULTS)
    v.update({k: int(x) for k, x in (kv.split("=") for kv in spec.split(",") if kv)})
    tape_mod.LIN_MODE, tape_mod.ATTN_VARIANT, tape_mod.GN_FORCE_SMALL = v["lin"], v["attn"], v["gn"]
    tape_mod.LATE_EPILOGUE, unet_mod.MERGE_FF2_PROJ, unet_mod.FOLD_XATTN = v["late"], bool(v["merge"]), bool(v["fold"])
    tape_mod.GN_VARIANT = 1 - v["gnreg"]
    tag = "_".join(f"{k}{v[k]}" for k in ("lin", "two", "late", "merge", "attn", "gn", "fold", "gnreg"))
    eng = UNetEngine(fam["unet"], packed, "cuda:0", B, 256, 16, ctx_len0=8, ctx_len1=16, two_source=bool(v["two"]))
    g = torch.Generator().manual_seed(1)
    eng.set_conditioning(ehs0=torch.randn(B, 8, 768, generator=g), ehs1=torch.randn(B, 16, 1024, generator=g),
                         bias1=torch.zeros(B, 16))
    eng.x_in.copy_(torch.randn(B, 256, 16, 8, generator=g))
    eng.set_timestep(500)
    with torch.cuda.stream(st):
        for _ in range(3):
            eng.forward()
        st.synchronize()
        ref_out = eng.eps.clone()
        t0 = time.time()
        for _ in range(10):
            eng.forward()
        st.synchronize()
        dt_e = (time.time() - t0) / 10
        eng.tape.capture()
        for _ in range(3):
            eng.tape.replay()
        st.synchronize()
        n = 30 if B <= 4 else 8
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(st)
        for _ in range(n):
            eng.tape.replay()
        ev1.record(st)
        ev1.synchronize()
        dt = ev0.elapsed_time(ev1) * 1e-3 / n
        print(f"[{tag}] B={B} ops={len(eng.tape.ops)} eager {dt_e * 1e3:.3f} ms | graph {dt * 1e3:.3f} ms -> "
              f"{eng.tape.flops / dt / 1e12:.1f} TF/s  finite={bool(torch.isfinite(ref_out).all())} "
              f"|eps|={ref_out.norm().item():.6e}", flush=True)
        ms = eng.tape.profile()
        ms = eng.tape.profile()
    agg = collections.defaultdict(lambda: [0, 0.0, 0])
    for m, t in zip(eng.tape.meta, ms):
        key = L_NAME = m["name"].split(".")[-1] if m["code"] != 1 else "conv_gemm:" + m["name"].split(".")[-1]
        a = agg[key]
        a[0] += 1
        a[1] += t
        a[2] += m["flops"]
    print(f"  sum of per-op ms {sum(ms):.3f}")
    for k, (cnt, t, f) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
        print(f"    {k:30s} n={cnt:4d} {t:8.3f} ms  {f / max(t, 1e-9) / 1e9:8.1f} TF/s")
    rows = [dict(name=m["name"], code=m["code"], ms=t, flops=m["flops"], i=list(op.i), flags=op.flags)
            for m, t, op in zip(eng.tape.meta, ms, eng.tape.ops)]
    json.dump(rows, open(f"gpurun_out/perop_B{B}_{tag}.json", "w"))
    del eng
    torch.cuda.empty_cache()

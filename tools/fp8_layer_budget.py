"""BASELINE config 5 names an "fp8 MFMA path"; the path's stated tolerance is 5e-3 on the edited latent.  Which GEMMs of the Stable
Audio DiT can go to the MX-FP8 matrix cores inside that budget?  Per GEMM family: build the full-depth model (24 layers, 1.06 B
seeded-random parameters) with ONLY that family on csrc/conv_gemm_f8.hip (tape.FP8_ONLY) and everything else on the fp32-exact
split-bf16 kernels, run the T = 200 / tstart = 100 inversion + edit of tests/golden/sa_parity_T200.npz, and report the deviation of
the edited latent / x_T / noise maps from the all-split-bf16 run of the same loops (and from the CPU oracle's fixture).

    PYTHONPATH=. python tools/fp8_layer_budget.py [families=ff1,ff2,qkv,attn1.to_out,attn2.q,attn2.to_out,all] > profiles/r06_fp8_budget.jsonl"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audioeditingcode_amd import models, tape as tape_mod                   # noqa: E402
from audioeditingcode_amd.ddm_inversion.inversion_utils import inversion_forward_process, inversion_reverse_process  # noqa: E402

DEV = "cuda:0"
FAMILIES = {"ff1": ".ff1", "ff2": ".ff2", "qkv": ".qkv", "attn1.to_out": ".attn1.to_out", "attn2.q": ".attn2.q",
            "attn2.to_out": ".attn2.to_out"}
fams = "ff1,ff2,qkv,attn1.to_out,attn2.q,attn2.to_out,attn(all four),all"
for a in sys.argv[1:]:
    if a.startswith("families="):
        fams = a.split("=", 1)[1]
fx = np.load(os.path.join(ROOT, "tests", "golden", "sa_parity_T200.npz"))
T, tstart = int(fx["T"]), int(fx["tstart"])
psrc, ptgt, pneg = (str(p) for p in fx["prompts"])
dur, (cs, ct) = float(fx["duration"]), (float(v) for v in fx["cfg"])
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())         # noqa: E731


def run(arith, only):
    """The loops with `only(name)` choosing the fp8 records; returns (edited latent, x_T, zs, share of GEMM flops on fp8, seconds)."""
    tape_mod.FP8_ONLY = only
    m = models.load_model("stabilityai/stable-audio-open-1.0", DEV, T, allow_synthetic=True)
    m.arith = arith
    t0 = time.time()
    with torch.inference_mode():
        torch.manual_seed(int(fx["seed"]))
        w_in = torch.from_numpy(fx["w0"]).to(DEV)
        _, zs, wts, extra = inversion_forward_process(m, w_in, etas=1.0, prompts=[psrc], cfg_scales=[cs], num_inference_steps=T,
                                                      numerical_fix=True, schedule="sequential", duration=dur)
        w_e, _ = inversion_reverse_process(m, xT=wts, tstart=torch.tensor([tstart]), etas=1.0, prompts=[ptgt], neg_prompts=[pneg],
                                           cfg_scales=[ct], zs=zs[:tstart], duration=dur, extra_info=extra)
        torch.cuda.synchronize()
    secs = time.time() - t0
    f8 = tot = 0.0
    for plan in m.editor()._plans.values():
        tp = plan["eng"].tape
        for op, meta in zip(tp.ops, tp.meta):
            if op.code == 1:
                tot += meta["flops"]
                f8 += meta["flops"] if op.flags & 64 else 0.0
        break
    out = (w_e.cpu().clone(), wts[-1].cpu().clone(), zs.cpu().clone(), f8 / max(tot, 1.0), secs)
    del m
    torch.cuda.empty_cache()
    tape_mod.FP8_ONLY = None
    return out


ref = run("bf16x6", None)
print(json.dumps(dict(family="none (all split-bf16)", vs_oracle_fixture_latent=rel(ref[0].reshape(fx["w_edit"].shape), torch.from_numpy(fx["w_edit"])),
                      seconds=round(ref[4], 1))), flush=True)
for fam in fams.split(","):
    if fam == "all":
        only = None
    elif fam.startswith("attn(all"):
        only = lambda n: any(k in n for k in (".qkv", ".attn1.to_out", ".attn2.q", ".attn2.to_out"))      # noqa: E731
    else:
        key = FAMILIES[fam]
        only = lambda n, key=key: key in n                                                                # noqa: E731
    w_e, xT, zs, share, secs = run("fp8", only)
    print(json.dumps(dict(family=fam, fp8_share_of_gemm_flops=round(share, 4),
                          edited_latent_rel_l2_vs_split_bf16=rel(w_e, ref[0]), xT_rel_l2=rel(xT, ref[1]),
                          zs_rel_l2_max=max(rel(zs[k], ref[2][k]) for k in (1, 100, 199)),
                          edited_latent_rel_l2_vs_oracle_fixture=rel(w_e.reshape(fx["w_edit"].shape), torch.from_numpy(fx["w_edit"])),
                          finite=bool(torch.isfinite(w_e).all()), seconds=round(secs, 1))), flush=True)

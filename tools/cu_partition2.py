"""Second premise test for the clip pipeline: (1) per-op profile of the batch-2 U-Net forward on a stream masked to
128 / 64 CUs (where does the CU time go once the step is no longer launch-bound?), (2) SEVERAL edit lanes (independent
batch-2 forwards of different clips, each its own engine and stream) sharing one CU partition next to the inversion on the
remaining CUs.

    PYTHONPATH=. python tools/cu_partition2.py -> gpurun_out/cu_partition2.json + lines on stdout"""
import collections
import json
import os
import sys
import time

import torch

from audioeditingcode_amd import configs, weights
from audioeditingcode_amd.streams import PartitionStream
from audioeditingcode_amd.unet import PackedUNetWeights, UNetEngine

G = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = "cuda:0"
TOTAL = 256
fam = configs.FAMILIES["audioldm2"]
sd = weights.random_state_dict(weights.unet_param_shapes(fam["unet"]), seed=0)
pw = PackedUNetWeights(sd, dev)
gen = torch.Generator().manual_seed(1)
out = {"G": G}


def mk(B):
    eng = UNetEngine(fam["unet"], pw, dev, B, 256, 16, ctx_len0=8, ctx_len1=16)
    eng.set_conditioning(ehs0=torch.randn(B, 8, 768, generator=gen), ehs1=torch.randn(B, 16, 1024, generator=gen),
                         bias1=torch.zeros(B, 16))
    eng.x_in.copy_(torch.randn(B, 256, 16, 8, generator=gen))
    eng.set_timestep(500)
    return eng


lanes = [mk(2) for _ in range(4)]
eb = mk(2 * G)
s0 = torch.cuda.Stream()
with torch.cuda.stream(s0):
    for e in lanes + [eb]:
        e.forward()
        s0.synchronize()
        e.tape.capture()
        e.tape.replay()
    s0.synchronize()


def replay_on(eng, ps, n):
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(ps.stream):
        ev0.record(ps.stream)
        for _ in range(n):
            eng.tape.replay()
        ev1.record(ps.stream)
    return ev0, ev1


# ---------------------------------------------------------------- (1) per-op profile under a mask
for c in (256, 128, 64):
    ps = PartitionStream(dev, cus=range(c))
    with torch.cuda.stream(ps.stream):
        lanes[0].tape.profile()
        ms = [lanes[0].tape.profile() for _ in range(3)]
    ms = [sum(x) / len(x) for x in zip(*ms)]
    agg = collections.defaultdict(lambda: [0, 0.0, 0])
    for m, t in zip(lanes[0].tape.meta, ms):
        key = m["name"].split(".")[-1] if m["code"] != 1 else "conv_gemm:" + m["name"].split(".")[-1]
        a = agg[key]
        a[0] += 1
        a[1] += t
        a[2] += m["flops"]
    a0, a1 = replay_on(lanes[0], ps, 20)
    a1.synchronize()
    print(f"--- batch 2 on {c} CUs: graph {a0.elapsed_time(a1) / 20:.3f} ms, sum of per-op event times {sum(ms):.3f} ms")
    for k, (cnt, t, f) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
        print(f"    {k:30s} n={cnt:4d} {t:8.3f} ms  {f / max(t, 1e-9) / 1e9:8.1f} TF/s", flush=True)
    rows = [dict(name=m["name"], code=m["code"], ms=t, flops=m["flops"], i=list(op.i))
            for m, t, op in zip(lanes[0].tape.meta, ms, lanes[0].tape.ops)]
    json.dump(rows, open(f"gpurun_out/perop_B2_cus{c}.json", "w"))
    ps.close(destroy=True)


# ---------------------------------------------------------------- (2) k edit lanes on one partition || inversion
def lanes_case(x, k, with_inv=True, n_edit=60):
    """k edit lanes share CUs [0, x); the inversion runs on [x, 256).  Returns per-lane edit forward ms under load,
    inversion forward ms under load."""
    pes = [PartitionStream(dev, cus=range(x)) if x < TOTAL else PartitionStream(dev) for _ in range(k)]
    pi = PartitionStream(dev, cus=range(x, TOTAL)) if (with_inv and x < TOTAL) else None
    torch.cuda.synchronize()
    inv_ms = None
    if pi is not None:
        a_i, b_i = replay_on(eb, pi, 4)
        time.sleep(0.05)
    evs = [replay_on(lanes[j], pes[j], n_edit) for j in range(k)]
    for a, b in evs:
        b.synchronize()
    edit_ms = [a.elapsed_time(b) / n_edit for a, b in evs]
    if pi is not None:
        b_i.synchronize()
        # inversion under continuously busy edit lanes
        torch.cuda.synchronize()
        evs = [replay_on(lanes[j], pes[j], 6 * n_edit) for j in range(k)]
        time.sleep(0.05)
        a_i, b_i = replay_on(eb, pi, 2)
        b_i.synchronize()
        inv_ms = a_i.elapsed_time(b_i) / 2
        torch.cuda.synchronize()
    per_clip_edit = 100 * max(edit_ms) / k
    r = dict(edit_cus=x, lanes=k, edit_fwd_ms_per_lane=edit_ms, inv_fwd_ms=inv_ms, edit_ms_per_clip=per_clip_edit,
             inv_ms_per_clip=None if inv_ms is None else (200 // G) * inv_ms,
             steady_state_ms_per_clip=per_clip_edit if inv_ms is None else max(per_clip_edit, (200 // G) * inv_ms))
    out[f"lanes_x{x}_k{k}_{'inv' if pi is not None else 'noinv'}"] = r
    print(f"edit partition {x} CUs, {k} lane(s){'' if pi is None else f', inversion on {TOTAL - x}'}: edit fwd/lane "
          f"{max(edit_ms):.3f} ms -> {per_clip_edit:.0f} ms per clip"
          + ("" if inv_ms is None else f"; inversion fwd {inv_ms:.1f} ms -> {(200 // G) * inv_ms:.0f} ms per clip")
          + f"; steady state {r['steady_state_ms_per_clip']:.0f} ms per clip", flush=True)
    for p in pes + ([pi] if pi is not None else []):
        p.close(destroy=True)


for k in (1, 2, 3, 4):
    lanes_case(TOTAL, k, with_inv=False)
for x, k in ((128, 1), (128, 2), (128, 3), (112, 2), (96, 2), (96, 3), (144, 2), (160, 3)):
    lanes_case(x, k)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/cu_partition2.json", "w"), indent=1)
print("wrote gpurun_out/cu_partition2.json")

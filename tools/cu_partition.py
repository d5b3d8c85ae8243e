"""CU-partition premise test for the two-clip pipeline.

One AudioLDM2 U-Net forward at batch 2 (the edit loop's shape, latency-bound) and one at batch 2G (the timestep-batched
inversion, throughput-bound) as hipGraphs; timed (a) alone on streams masked to c CUs, (b) together on DISJOINT masks
(edit: x CUs, inversion: 256-x), (c) together on unmasked streams with / without priorities.  Also records which physical
CUs a mask lands on (aed_cu_census).

    PYTHONPATH=. python tools/cu_partition.py [G] -> gpurun_out/cu_partition.json + lines on stdout"""
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


e2, eb = mk(2), mk(2 * G)
s0 = torch.cuda.Stream()
with torch.cuda.stream(s0):
    for e in (e2, eb):
        e.forward()
        s0.synchronize()
        e.tape.capture()
        e.tape.replay()
    s0.synchronize()
ref2, refb = e2.eps.clone(), eb.eps.clone()
TOTAL = 256


def replay_on(eng, ps, n):
    """n graph replays on the partition stream; returns (ev0, ev1)."""
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(ps.stream):
        ev0.record(ps.stream)
        for _ in range(n):
            eng.tape.replay()
        ev1.record(ps.stream)
    return ev0, ev1


def alone(eng, ps, n):
    replay_on(eng, ps, 1)[1].synchronize()
    a, b = replay_on(eng, ps, n)
    b.synchronize()
    return a.elapsed_time(b) / n


def census_summary(ps):
    cs = ps.census()
    per_xcc = {}
    for x, se, sh, cu in cs:
        per_xcc[x] = per_xcc.get(x, 0) + 1
    return dict(n_cus=len(cs), per_xcc=per_xcc)


# ---------------------------------------------------------------- (0) what a mask means physically
un = PartitionStream(dev)
out["census_unmasked"] = census_summary(un)
print("census unmasked:", out["census_unmasked"], flush=True)
for name, bits in (("low64", range(64)), ("high192", range(64, 256)), ("low8", range(8)), ("xcd01", [b for b in range(256) if b % 8 < 2])):
    ps = PartitionStream(dev, cus=bits)
    out["census_" + name] = census_summary(ps)
    print(f"census {name}:", out["census_" + name], flush=True)
    ps.close(destroy=True)

# ---------------------------------------------------------------- (a) alone, masked
out["alone_b2_ms"], out["alone_big_ms"] = {}, {}
out["alone_b2_ms"]["unmasked"] = alone(e2, un, 30)
out["alone_big_ms"]["unmasked"] = alone(eb, un, 2)
print(f"alone unmasked: batch 2 {out['alone_b2_ms']['unmasked']:.3f} ms, batch {2 * G} {out['alone_big_ms']['unmasked']:.2f} ms",
      flush=True)
for c in (256, 192, 160, 128, 96, 64, 48, 32):
    ps = PartitionStream(dev, cus=range(c))
    out["alone_b2_ms"][str(c)] = alone(e2, ps, 30)
    print(f"alone batch 2 on {c} CUs: {out['alone_b2_ms'][str(c)]:.3f} ms", flush=True)
    ps.close(destroy=True)
for c in (256, 224, 192, 160, 128):
    ps = PartitionStream(dev, cus=range(TOTAL - c, TOTAL))
    out["alone_big_ms"][str(c)] = alone(eb, ps, 2)
    print(f"alone batch {2 * G} on {c} CUs: {out['alone_big_ms'][str(c)]:.2f} ms", flush=True)
    ps.close(destroy=True)
assert torch.equal(ref2, e2.eps) and torch.equal(refb, eb.eps), "masked replays changed the results"


# ---------------------------------------------------------------- (b), (c) together
def together(ps_edit, ps_inv, label):
    # edit loop under a continuously busy inversion stream
    torch.cuda.synchronize()
    a_i, b_i = replay_on(eb, ps_inv, 4)
    time.sleep(0.05)
    a_e, b_e = replay_on(e2, ps_edit, 60)
    b_e.synchronize()
    edit_loaded = a_e.elapsed_time(b_e) / 60
    b_i.synchronize()
    # inversion under a continuously busy edit stream
    torch.cuda.synchronize()
    a_e, b_e = replay_on(e2, ps_edit, 400)
    time.sleep(0.05)
    a_i, b_i = replay_on(eb, ps_inv, 2)
    b_i.synchronize()
    inv_loaded = a_i.elapsed_time(b_i) / 2
    b_e.synchronize()
    # one "clip pair": 100 edit forwards || T/G inversion forwards, wall time of both
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    replay_on(eb, ps_inv, 200 // G)
    replay_on(e2, ps_edit, 100)
    torch.cuda.synchronize()
    pair = (time.perf_counter() - t0) * 1e3
    r = dict(edit_fwd_ms_under_inversion=edit_loaded, inv_fwd_ms_under_edit=inv_loaded, pair_wall_ms=pair)
    out["together_" + label] = r
    print(f"together {label}: edit fwd {edit_loaded:.3f} ms, inversion fwd {inv_loaded:.2f} ms, "
          f"[100 edit || {200 // G} inversion] wall {pair:.1f} ms", flush=True)


for x in (48, 64, 80, 96, 128):
    pe, pi = PartitionStream(dev, cus=range(x)), PartitionStream(dev, cus=range(x, TOTAL))
    together(pe, pi, f"edit{x}_inv{TOTAL - x}")
    pe.close(destroy=True)
    pi.close(destroy=True)
for nx in (2, 3):     # whole XCDs for the edit loop (its own L2s)
    pe = PartitionStream(dev, cus=[b for b in range(TOTAL) if b % 8 < nx])
    pi = PartitionStream(dev, cus=[b for b in range(TOTAL) if b % 8 >= nx])
    together(pe, pi, f"edit_{nx}xcd_inv_{8 - nx}xcd")
    pe.close(destroy=True)
    pi.close(destroy=True)
# overlapping masks: the edit stream may use every CU, the inversion leaves x free
for x in (64, 96):
    pe, pi = PartitionStream(dev, cus=range(TOTAL)), PartitionStream(dev, cus=range(x, TOTAL))
    together(pe, pi, f"edit256_inv{TOTAL - x}")
    pe.close(destroy=True)
    pi.close(destroy=True)
hi, lo = PartitionStream(dev, priority=-1), PartitionStream(dev, priority=0)
together(hi, lo, "unmasked_edit_high_priority")
together(un, lo, "unmasked_equal_priority")
assert torch.equal(ref2, e2.eps) and torch.equal(refb, eb.eps), "concurrent replays changed the results"
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/cu_partition.json", "w"), indent=1)
print("wrote gpurun_out/cu_partition.json")

"""Tile autotuning sweep for AED_OP_CONV_GEMM on the MI355X.

For every distinct contraction of the AudioLDM2 U-Net forward at U-Net batch B, time each candidate tile config the way
the op runs inside the edit / inversion loops: R dependent launches captured in one hipGraph, activations warm (just
written), WEIGHTS COLD (every launch reads its weight matrix from a different slice of a 768 MB pool, so nothing is
served from the 256 MB Infinity Cache -- a forward streams 1.39 GB of weights, i.e. they are always cold in the loop).

    PYTHONPATH=. python tools/tile_sweep.py <B> [R]      -> gpurun_out/tile_sweep_B<B>.json + a table on stdout
    PYTHONPATH=. python tools/tile_sweep.py <B> [R] dit  -> the contractions of the Stable Audio DiT forward instead
                                                            (gpurun_out/tile_sweep_dit_B<B>.json)

    AED_SWEEP_CUS=64 ...                                 -> the same on a stream masked to 64 CUs (one lane of a pipeline)
    AED_SWEEP_ARITH=bf16x6 ...                           -> the split-bf16 kernel's tiles compete too (tile codes 100 + tile:
                                                            101, 102, 103, 104, 108, 109; csrc/conv_gemm_x6.hip)

The winners are pasted into audioeditingcode_amd/tile_table.py (tools/tile_table_from_sweep.py does it; `--x6` writes the
per-shape arithmetic + tile table of engines built under tape.arith_mode("bf16x6"))."""
import collections
import ctypes
import json
import os
import sys

import torch

from audioeditingcode_amd import _lib as L, configs, weights
from audioeditingcode_amd.tape import Tape
from audioeditingcode_amd.unet import UNetEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
R = int(sys.argv[2]) if len(sys.argv) > 2 else (120 if B <= 4 else 16)
POOL_BYTES = 768 << 20
dev = "cuda:0"
DIT = len(sys.argv) > 3 and sys.argv[3] == "dit"
if DIT:
    from audioeditingcode_amd.stable_audio import DiTEngine
    dcfg = dict(configs.FAMILIES["stable_audio"]["dit"], num_layers=1)      # every layer has the same contractions
    sd = weights.random_state_dict(weights.dit_param_shapes(dcfg), seed=0)
    eng = DiTEngine(dcfg, sd, dev, B, 130)
    g = torch.Generator().manual_seed(1)
    eng.set_conditioning(torch.randn(B, 130, dcfg["cross_attention_input_dim"], generator=g),
                         torch.randn(B, dcfg["global_states_input_dim"], generator=g))
    eng.set_timestep(0.4)
    eng.x_in.copy_(torch.randn(B, dcfg["sample_size"], dcfg["in_channels"], generator=g))
else:
    fam = configs.FAMILIES["audioldm2"]
    sd = weights.random_state_dict(weights.unet_param_shapes(fam["unet"]), seed=0)
    eng = UNetEngine(fam["unet"], sd, dev, B, 256, 16, ctx_len0=8, ctx_len1=16)
    g = torch.Generator().manual_seed(1)
    eng.set_conditioning(ehs0=torch.randn(B, 8, 768, generator=g), ehs1=torch.randn(B, 16, 1024, generator=g),
                         bias1=torch.zeros(B, 16))
    eng.x_in.copy_(torch.randn(B, 256, 16, 8, generator=g))
    eng.set_timestep(500)
# AED_SWEEP_CUS=n: run the sweep on a stream masked to n CUs (tile choices for one partition of the clip pipeline)
SWEEP_ARITH = os.environ.get("AED_SWEEP_ARITH", "f32")
SWEEP_CUS = int(os.environ.get("AED_SWEEP_CUS", "0"))
if SWEEP_CUS:
    from audioeditingcode_amd.streams import PartitionStream
    _ps = PartitionStream(dev, cus=range(SWEEP_CUS))
    st = _ps.stream
else:
    st = torch.cuda.Stream()
pool = torch.empty(POOL_BYTES // 4, device=dev, dtype=torch.float32).normal_(0, 0.02)
ws = torch.empty((512 if DIT else 64) << 20, device=dev, dtype=torch.float32)          # split-K workspace for legacy configs

with torch.cuda.stream(st):
    eng.forward()           # every activation buffer holds realistic values
st.synchronize()

# distinct contractions: (M, N, K, taps, geglu, ln, two-source, stride, up) -> representative op + count
reps = collections.OrderedDict()
for op, meta in zip(eng.tape.ops, eng.tape.meta):
    if op.code != L.OP_CONV_GEMM:
        continue
    i = op.i
    key = (i[0], i[1], i[2], i[12] * i[13], i[35], i[31], int(i[32] > 0), i[14], i[19])
    if key not in reps:
        reps[key] = [op, 0, meta["name"], meta["flops"]]
    reps[key][1] += 1


def candidates(M, N, K, taps, geglu, generic):
    if generic:
        return [(0, 0)]
    if geglu:
        c = [(13, 1), (14, 1), (15, 1), (17, 1), (1, 1), (3, 1)]
    else:
        c = [(10, 1), (11, 1), (12, 1), (18, 1), (19, 1), (13, 1), (15, 1), (16, 1), (17, 1), (4, 1), (1, 1), (2, 1), (3, 1)]
        t4 = -(-M // 64) * -(-N // 64)
        if t4 < 256 and K >= 256:          # legacy split-K + reduce
            nch = -(-K // 32)
            c.append((4, max(1, min(-(-512 // t4), nch // 4, 32))))
    big = M * N >= 4096 * 1024
    if big:                                 # throughput regime: the 32x32 lin tiles only add L2 traffic
        c = [x for x in c if x[0] in (1, 2, 3, 4, 15, 17)]
    if SWEEP_ARITH == "bf16x6":             # the split-bf16 kernel (flags 4|8), unsplit K
        c += [(100 + t, 1) for t in ((1, 3, 8, 9) if geglu else (1, 2, 3, 4, 8, 9))]
    return c


def time_cfg(op, tile, ksplit):
    tp = Tape(dev)
    wbytes = op.i[1] * op.i[2] * 4
    step = -(-wbytes // 4096) * 4096
    span = POOL_BYTES - wbytes - 4096
    for r in range(R):
        o = L.aed_op()
        ctypes.memmove(ctypes.byref(o), ctypes.byref(op), ctypes.sizeof(L.aed_op))
        o.i[29] = tile % 100
        o.i[28] = ksplit
        if tile >= 100:
            o.flags |= 12
        o.p[1] = pool.data_ptr() + (r * step) % span
        if ksplit > 1:
            o.p[6] = ws.data_ptr()
            assert ksplit * op.i[0] * op.i[1] <= ws.numel()
        tp.ops.append(o)
        tp.meta.append(dict(name="x", code=1, flops=0, bytes=0))
    tp._arr = None
    with torch.cuda.stream(st):
        try:
            tp.run()
            st.synchronize()
        except L.AedError as e:
            return None, str(e)[-80:]
        gph = tp.capture()
        tp.replay()
        st.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            ev0.record(st)
            tp.replay()
            ev1.record(st)
            ev1.synchronize()
            best = min(best, ev0.elapsed_time(ev1) * 1e3 / R)
    return best, ""


rows = []
tot_auto = tot_best = 0.0
for key, (op, count, name, flops) in reps.items():
    M, N, K, taps, geglu, ln, two, stride, up = key
    generic = (op.i[11] % 32 != 0) or (op.i[3] % 4 != 0)
    res = {}
    auto = (op.i[29], op.i[28])
    cands = candidates(M, N, K, taps, geglu, generic)
    if auto not in cands:
        cands.append(auto)
    for tile, ks in cands:
        us, err = time_cfg(op, tile, ks)
        if us is not None:
            res[f"{tile}:{ks}"] = us
    best = min(res, key=res.get)
    a_us = res.get(f"{auto[0]}:{auto[1]}")
    tot_auto += count * (a_us or 0)
    tot_best += count * res[best]
    rows.append(dict(M=M, N=N, K=K, taps=taps, geglu=geglu, ln=ln, two_source=two, stride=stride, up=up, count=count,
                     name=name, flops=flops, auto=f"{auto[0]}:{auto[1]}", auto_us=a_us, best=best, best_us=res[best],
                     all=res))
    tf = flops / res[best] / 1e6
    print(f"{M:7d} {N:5d} {K:6d} t{taps} g{geglu} l{ln} s{two} x{count:3d}  auto {auto[0]:2d}:{auto[1]:<2d} "
          f"{(a_us or 0):7.1f} us | best {best:>6s} {res[best]:7.1f} us {tf:6.1f} TF/s | "
          + " ".join(f"{k}={v:.1f}" for k, v in sorted(res.items(), key=lambda kv: kv[1])[:6]), flush=True)
print(f"B={B}: conv_gemm per forward with the current rule {tot_auto / 1e3:.3f} ms, with per-shape best {tot_best / 1e3:.3f} ms")
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open(f"gpurun_out/tile_sweep_{'dit_' if DIT else ''}B{B}{f'_cus{SWEEP_CUS}' if SWEEP_CUS else ''}"
                     f"{'_x6' if SWEEP_ARITH == 'bf16x6' else ''}.json", "w"), indent=1)

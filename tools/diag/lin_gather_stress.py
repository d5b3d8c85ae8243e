"""Diagnostic 8 (round 6): the latency-regime GEMM (csrc/lin_gemm.hip) under CU co-residency, outside any engine.

One AED_OP_CONV_GEMM record on a 128-CU masked stream is launched R times into R separate output buffers while a stressor runs
on an unmasked stream (its workgroups land on the same CUs); every output is compared with the solo launch.  Census over tile
codes 10..19, gather / uniform loader, shapes of the U-Net's latency regime; stressors with and without MFMA / LDS.

    python tools/diag/lin_gather_stress.py [cases=head|one|census] [stress=x6,f32,copy,none] [R=60] [lib=path/to/libaed_variant.so] [dump=1|2]

`lib=`: a variant of the library, e.g. the round-5 form of the gather loader, to see that the harness still bites (about half of all
launches of tiles 11 / 15 perturbed under the x6 stressor; the shipped kernel: none):

    cd audioeditingcode_amd/csrc && bash build.sh && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DLIN_GATHER_R5_FORM -c lin_gemm.hip \
        -o ../../scratch/lin_r5.o && hipcc --offload-arch=gfx950 -shared -fPIC $(ls obj/*.o | grep -v lin_gemm) ../../scratch/lin_r5.o \
        -o ../../scratch/libaed_r5form.so
    python tools/diag/lin_gather_stress.py cases=one stress=x6 R=200 lib=scratch/libaed_r5form.so

(`dump=1|2` belong to instrumented builds of round 6 -- per-thread partial tiles / per-lane loader checksums -- whose kernel hooks were
removed again; profiles/r06_lin_gather_hazard.md says what they showed.)
"""
import os
import sys
import traceback

import torch

sys.path.insert(0, ".")
from audioeditingcode_amd import _lib as L                                  # noqa: E402

o = dict(cases="head", stress="x6,copy,none", R="60", lib="", dump="0")
for a in sys.argv[1:]:
    k, _, v = a.partition("=")
    o[k] = v
if o["lib"]:
    L.LIB_PATH = os.path.abspath(o["lib"])

from audioeditingcode_amd import tape as tape_mod                           # noqa: E402
from audioeditingcode_amd.streams import PartitionStream                   # noqa: E402
from audioeditingcode_amd.tape import Tape                                 # noqa: E402

DEV = torch.device("cuda:0")


def make_case(name, B, IH, IW, Cin, N, k, stride, tile, R, gen, in_act=0):
    """R records of one convolution (k x k, stride, 'same'-style padding k // 2), each with its own output buffer."""
    pad = k // 2
    OH, OW = (IH + 2 * pad - k) // stride + 1, (IW + 2 * pad - k) // stride + 1
    x = torch.randn(B, IH, IW, Cin, generator=gen, device=DEV)
    w = torch.randn(N, k * k * Cin, generator=gen, device=DEV) / (k * k * Cin) ** 0.5
    b = torch.randn(N, generator=gen, device=DEV)
    outs = torch.zeros(R + 1, B, OH, OW, N, device=DEV)
    tp = Tape(DEV)
    with tape_mod.arith_mode("f32"):
        for r in range(R + 1):
            tp.conv(x, w, b, outs[r], B=B, IH=IH, IW=IW, Cin=Cin, OH=OH, OW=OW, N=N, KH=k, KW=k, stride=stride, pad_h=pad,
                    pad_w=pad, tile=tile, in_act=in_act, in_slope=0.1, name=name)
    dbg = None
    if o["dump"] == "2":                    # LIN_DIAG=60 library: per lane and loader row j: sum of raw loads, of masks, of staged values
        NW = {10: 4, 11: 8, 12: 16, 13: 4, 14: 8, 15: 4, 16: 4, 17: 8, 18: 10, 19: 12}[tile]
        bm, bn = Tape.LIN_TILES[tile]
        nwg = -(-B * OH * OW // bm) * -(-N // bn)
        dbg = torch.zeros(R + 1, nwg * NW * 64 * (bm // 8) * 3, device=DEV)
        for r in range(R + 1):
            tp.ops[r].flags |= 1
            tp.ops[r].p[7] = dbg[r].data_ptr()
        tp._arr = None
    if o["dump"] == "1":                    # LIN_DIAG=11 library: every partial tile value as the finishing thread read it
        NW = {10: 4, 11: 8, 12: 16, 13: 4, 14: 8, 15: 4, 16: 4, 17: 8, 18: 10, 19: 12}[tile]
        dbg = torch.zeros(R + 1, B * OH * OW * N * NW, device=DEV)
        for r in range(R + 1):
            tp.ops[r].flags |= 1
            tp.ops[r].p[7] = dbg[r].data_ptr()
        tp._arr = None
    tp.finalize()
    return dict(name=name, tape=tp, outs=outs, x=x, w=w, M=B * OH * OW, N=N, K=k * k * Cin, tile=tile, R=R, dbg=dbg)


def make_victim(kind, R, gen):
    """R + 1 records of another kernel family, each with its own output: do THEY depend on co-resident workgroups?  (Round 6 found the
    defect in lin_gemm's gather loader only; this is the census over everything else an edit lane runs.)"""
    tp = Tape(DEV)
    fam, *arg = kind
    if fam == "conv":                       # LDS-staged GEMM: (arith, tile code, k)
        arith, tile, k = arg
        B, IH, IW, Cin, N = 2, 128, 8, 256, 256
        if tile in (8, 9):                  # the 512-thread split-bf16 tiles are the launcher's own pick at throughput shapes
            B, N, tile = 32, (256 if tile == 8 else 768), 0
        x = torch.randn(B, IH, IW, Cin, generator=gen, device=DEV)
        w = torch.randn(N, k * k * Cin, generator=gen, device=DEV) / (k * k * Cin) ** 0.5
        b = torch.randn(N, generator=gen, device=DEV)
        outs = torch.zeros(R + 1, B, IH, IW, N, device=DEV)
        with tape_mod.arith_mode(arith):
            for r in range(R + 1):
                tp.conv(x, w, b, outs[r], B=B, IH=IH, IW=IW, Cin=Cin, OH=IH, OW=IW, N=N, KH=k, KW=k, pad_h=k // 2, pad_w=k // 2,
                        tile=tile, in_act=1 if k == 3 else 0, name="victim")
    elif fam == "geglu":                    # FF1 + LayerNorm fold + GEGLU on the lin tiles that have it
        (tile,) = arg
        M, C = 2048, 256
        x = torch.randn(M, C, generator=gen, device=DEV)
        w = torch.randn(8 * C, C, generator=gen, device=DEV) / C ** 0.5
        b = torch.randn(8 * C, generator=gen, device=DEV)
        rs = w.sum(1).contiguous()
        outs = torch.zeros(R + 1, M, 4 * C, device=DEV)
        with tape_mod.arith_mode("f32"):
            for r in range(R + 1):
                tp.linear(x, w, b, outs[r], M=M, K=C, N=8 * C, ln_rowsum=rs, geglu=1, tile=tile, name="victim")
    elif fam == "attn":                     # (variant, Nk, D)
        variant, Nk, D = arg
        B, H, Nq = 2, 8, 1024
        C = H * D
        q = torch.randn(B, Nq, C, generator=gen, device=DEV)
        kk = torch.randn(B, Nk, C, generator=gen, device=DEV)
        v = torch.randn(B, Nk, C, generator=gen, device=DEV)
        outs = torch.zeros(R + 1, B, Nq, C, device=DEV)
        with tape_mod.arith_mode("bf16x6" if variant == 3 else "f32"):
            for r in range(R + 1):
                tp.attention(q, kk, v, outs[r], B=B, H=H, Nq=Nq, Nk=Nk, D=D, ldq=C, ldk=C, ldv=C, ldo=C, bsq=Nq * C, bsk=Nk * C,
                             bsv=Nk * C, bso=Nq * C, scale=D ** -0.5, variant=variant)
    elif fam == "gn":                       # (HW, C): the single-launch kernel for small maps, stats + apply for large ones
        HW, C = arg
        x = torch.randn(2, HW, 1, C, generator=gen, device=DEV)
        ga, be = torch.randn(C, generator=gen, device=DEV), torch.randn(C, generator=gen, device=DEV)
        outs = torch.zeros(R + 1, 2, HW, 1, C, device=DEV)
        for r in range(R + 1):
            tp.groupnorm(x, ga, be, outs[r], B=2, HW=HW, C=C, G=32, eps=1e-5, act=1)
    else:
        raise ValueError(kind)
    tp.finalize()
    per = len(tp.ops) // (R + 1)
    return dict(name=str(kind), tape=tp, outs=outs, per=per)


VICTIMS = ([("conv", a, t, k) for a in ("bf16x6", "f32") for t in (1, 2, 3, 4) for k in (3, 1)] +
           [("conv", "bf16x6", 8, 3), ("conv", "bf16x6", 9, 1)] +
           [("geglu", t) for t in (13, 14, 15, 17)] +
           [("attn", 0, 1024, 32), ("attn", 1, 1024, 32), ("attn", 3, 1024, 32), ("attn", 3, 1024, 64), ("attn", 0, 16, 32)] +
           [("gn", 1024, 256), ("gn", 4096, 128), ("gn", 64, 1280)])


def run_victims(R, lane, side, gen):
    stress = make_stressor("x6", gen)
    clean = 0
    for kind in VICTIMS:
        try:
            c = make_victim(kind, R, gen)
        except Exception as e:                                          # noqa: BLE001
            print(f"victim {kind}: not built ({e!r})", flush=True)
            continue
        tp, outs, per = c["tape"], c["outs"], c["per"]
        try:
            with torch.cuda.stream(lane.stream):
                tp.run(0, per)
            torch.cuda.synchronize()
            with torch.cuda.stream(side.stream):
                for _ in range(max(1, R // 8)):
                    stress.run()
            with torch.cuda.stream(lane.stream):
                tp.run(per, per * (R + 1))
            torch.cuda.synchronize()
        except Exception as e:                                          # noqa: BLE001
            print(f"victim {c['name']}: not launched ({e!r})", flush=True)
            continue
        bad = [r for r in range(1, R + 1) if not torch.equal(outs[r], outs[0])]
        clean += not bad
        kernels = sorted({int(op.i[29]) for op in tp.ops if op.code == 1})
        print(f"victim {c['name']}: {per} launch(es) per record, conv tiles {kernels}; perturbed records {len(bad)} of {R}"
              + (f"; first: max |d| {float((outs[bad[0]] - outs[0]).abs().max()):.3g}" if bad else ""), flush=True)
        del c, tp, outs
        torch.cuda.empty_cache()
    print(f"victims clean: {clean} of {len(VICTIMS)}", flush=True)


def make_stressor(kind, gen):
    tp = Tape(DEV)
    if kind == "none":
        return None
    if kind == "copy":
        src = torch.randn(64, 1 << 20, generator=gen, device=DEV)
        dst = torch.empty_like(src)
        for _ in range(4):
            tp.copy2d(src, dst, rows=64, cols=1 << 20, ld_src=1 << 20, ld_dst=1 << 20)
        tp.finalize()
        return tp
    # a VAE-encoder-like 3x3 convolution: 128 -> 128 channels on a 1024 x 64 map (MFMA + LDS staged tiles)
    B, H, W, C = 1, 1024, 64, 128
    x = torch.randn(B, H, W, C, generator=gen, device=DEV)
    w = torch.randn(C, 9 * C, generator=gen, device=DEV) / (9 * C) ** 0.5
    out = torch.empty(B, H, W, C, device=DEV)
    with tape_mod.arith_mode("bf16x6" if kind == "x6" else "f32"):
        for _ in range(4):
            tp.conv(x, w, None, out, B=B, IH=H, IW=W, Cin=C, OH=H, OW=W, N=C, KH=3, KW=3, pad_h=1, pad_w=1, name="stress")
    tp.finalize()
    return tp


def rows_mod32(out, ref):
    d = (out != ref).reshape(-1, out.shape[-1]).any(1).nonzero().reshape(-1)
    return sorted(set((d % 32).tolist()))


def main():
    R = int(o["R"])
    gen = torch.Generator(device=DEV)
    gen.manual_seed(0)
    lane = PartitionStream.acquire(DEV, cus=range(0, 128), total=256, index=0)
    side = PartitionStream.acquire(DEV, index=17)
    if o["cases"] == "victims":
        print(f"lib={L.LIB_PATH}", flush=True)
        return run_victims(R, lane, side, gen)
    if o["cases"] == "head":
        specs = [("downsampler 3x3 s2 (M=1024,K=1152)", 1, 256, 16, 128, 128, 3, 2, t) for t in (11, 17, 10, 12)]
        specs += [("resnet conv1 3x3 (M=1024,K=1152->256)", 1, 128, 8, 128, 256, 3, 1, t) for t in (11, 17)]
    elif o["cases"] == "one":
        specs = [("3x3 s1 M=1024 K=2304 N=256", 1, 128, 8, 256, 256, 3, 1, t) for t in (11, 15)]
    else:
        specs = []
        for t in (10, 11, 12, 13, 14, 15, 16, 17, 18, 19):
            specs.append((f"3x3 s1 M=1024 K=2304 N=256", 1, 128, 8, 256, 256, 3, 1, t))
            specs.append((f"1x1 M=1024 K=256 N=256", 1, 128, 8, 256, 256, 1, 1, t))
            specs.append((f"3x3 s1 M=512 K=3456 N=384", 2, 64, 4, 384, 384, 3, 1, t))
            specs.append((f"1x1 M=2048 K=256 N=256", 2, 128, 8, 256, 256, 1, 1, t))
            specs.append((f"1x1 M=128 K=640 N=640", 2, 32, 2, 640, 640, 1, 1, t))
        for t in (11, 15, 10, 17):             # loader activations (SiLU / LeakyReLU) on the gather path
            specs.append((f"3x3 s1 M=1024 K=2304 N=256 SiLU loader", 1, 128, 8, 256, 256, 3, 1, t, 1))
            specs.append((f"3x3 s1 M=1024 K=2304 N=256 LeakyReLU loader", 1, 128, 8, 256, 256, 3, 1, t, 2))
    print(f"lib={L.LIB_PATH}", flush=True)
    for kind in o["stress"].split(","):
        stress = make_stressor(kind, gen)
        for spec in specs:
            name, tile = spec[0], spec[8]
            try:
                c = make_case(name, *spec[1:9], R, gen, *spec[9:])
            except Exception as e:                                      # noqa: BLE001
                print(f"stress={kind:5s} tile={tile} {name}: not built ({e})", flush=True)
                continue
            tp, outs = c["tape"], c["outs"]
            with torch.cuda.stream(lane.stream):
                tp.run(0, 1)                                            # solo reference (record 0)
            torch.cuda.synchronize()
            with torch.cuda.stream(lane.stream):
                tp.run(1, 3)
            torch.cuda.synchronize()
            solo_ok = torch.equal(outs[1], outs[0]) and torch.equal(outs[2], outs[0])
            outs[1:].zero_()
            torch.cuda.synchronize()
            if stress is not None:
                with torch.cuda.stream(side.stream):
                    for _ in range(max(1, R // 8)):
                        stress.run()
            with torch.cuda.stream(lane.stream):
                tp.run(1, R + 1)
            torch.cuda.synchronize()
            bad = [r for r in range(1, R + 1) if not torch.equal(outs[r], outs[0])]
            msg = f"stress={kind:5s} tile={tile} {name}: solo repeat ok={solo_ok}; perturbed launches {len(bad)} of {R}"
            if bad:
                r = bad[0]
                ne = int((outs[r] != outs[0]).sum())
                msg += (f"; first: {ne} of {outs[0].numel()} elements, max |d| {float((outs[r] - outs[0]).abs().max()):.3g}, "
                        f"rows mod 32 {rows_mod32(outs[r], outs[0])}")
            print(msg, flush=True)
            if bad and c["dbg"] is not None and o["dump"] == "2":
                NW = {10: 4, 11: 8, 12: 16, 13: 4, 14: 8, 15: 4, 16: 4, 17: 8, 18: 10, 19: 12}[tile]
                PA = Tape.LIN_TILES[tile][0] // 8
                d0 = c["dbg"][0].view(-1, NW, 64, PA, 3)
                for r in bad[:4]:
                    d = c["dbg"][r].view(-1, NW, 64, PA, 3)
                    diff = (d != d0)
                    print(f"  launch {r}: checksums that differ: raw loads {int(diff[..., 0].sum())}, masks {int(diff[..., 1].sum())}, "
                          f"staged (LDS read-back) {int(diff[..., 2].sum())}; lanes {sorted(set(diff.any(-1).nonzero()[:, 2].tolist()))}; "
                          f"loader rows j {sorted(set(diff.any(-1).nonzero()[:, 3].tolist()))}; waves {sorted(set(diff.any(-1).nonzero()[:, 1].tolist()))}",
                          flush=True)
                    for wg, w, ln, j in diff.any(-1).nonzero()[:6].tolist():
                        print(f"    wg {wg} wave {w} lane {ln} j {j}: solo raw/mask/staged {[round(float(v), 5) for v in d0[wg, w, ln, j]]} "
                              f"now {[round(float(v), 5) for v in d[wg, w, ln, j]]}", flush=True)
            elif bad and c["dbg"] is not None:
                NW = c["dbg"].shape[1] // outs[0].numel()
                d0 = c["dbg"][0].view(-1, c["N"], NW)
                for r in bad[:3]:
                    d = c["dbg"][r].view(-1, c["N"], NW)
                    o2, o0 = outs[r].reshape(-1, c["N"]), outs[0].reshape(-1, c["N"])
                    idx = (o2 != o0).nonzero()
                    pd = (d != d0)
                    print(f"  launch {r}: {idx.shape[0]} outputs differ; partials that differ per wave slab w: "
                          f"{[int(pd[..., w].sum()) for w in range(NW)]}; outputs that differ although every partial read equal: "
                          f"{int(((o2 != o0) & ~pd.any(-1)).sum())}", flush=True)
                    for m_, n_ in idx[:4].tolist():
                        print(f"    out[{m_},{n_}] solo {float(o0[m_, n_]):.6f} now {float(o2[m_, n_]):.6f}; partials solo "
                              f"{[round(float(v), 5) for v in d0[m_, n_]]} now {[round(float(v), 5) for v in d[m_, n_]]}", flush=True)
                        # is the wrong partial some OTHER row's partial of the same slab?
                        for w in range(NW):
                            if d[m_, n_, w] != d0[m_, n_, w]:
                                hit = (d0[:, :, w] == d[m_, n_, w]).nonzero()[:3].tolist()
                                print(f"      slab {w}: the value read equals the solo partial of outputs {hit}", flush=True)
            del c, tp, outs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    try:
        with torch.inference_mode():
            main()
    except BaseException:                       # noqa: BLE001
        traceback.print_exc()

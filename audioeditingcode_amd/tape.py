"""Op-tape builder/executor: the host side of include/aed.h's `aed_op` records.

A model forward (U-Net, VAE, vocoder, STFT) is compiled ONCE into a flat array of op records that
reference pre-allocated device buffers; `Tape.run()` hands the array to the native executor
(aed_tape_run), which issues every kernel from C++ on the caller's HIP stream -- no Python per
kernel, and the whole tape can be captured in a hipGraph (`Tape.capture()` / `Tape.replay()`).

Layout convention: activations are channels-last fp32 -- a [B,H,W,C] feature map IS the
[B*H*W, C] token matrix; convolution weights are [Cout, KH, KW, Cin].
"""
import contextlib
import ctypes
import math
import threading

import torch

from . import _lib as L

CU_COUNT = 256          # MI355X; tiling heuristics target this, overridable for tests


# Build-variant constants.  These were environment switches while the round-2 kernels were being A/B'd; the product has one
# configuration now, and the profiling tools (tools/unet_profile.py) set the module attributes directly for A/B runs.
FORCE_BK = 0            # K-chunk width of conv_gemm: 0 auto, 1 force 32, 2 allow 64 on the 128x128 tile too
ATTN_VARIANT = 0        # 0 auto (transposed-score kernel when Nk > 64), 1 forces the single-pass kernel
ATTN_X6 = 1             # 0: attention records stay on the fp32 kernels under arith_mode("bf16x6") as well (A/B)
GN_VARIANT = 0          # 1: never use the register-resident single-launch GroupNorm
GN_FORCE_SMALL = 0      # 1: always take the single-launch GroupNorm when it fits
GN_APPLY_BLOCKS_PER_CU = 8   # gn_apply grid: blocks per CU over the whole batch.  2 until round 6: at the inversion's batch (200 items x 2
#                              blocks) the kernel had ~1.5 blocks per CU in flight and ran at 3.5 of 8 TB/s; 8 -> 4.8-5.7 TB/s, bit-identical
#                              (tools/gn_bench.py, profiles/r06_gn_grid.jsonl)
LIN_MODE = 1            # 1: small contractions go to the latency-regime kernels of lin_gemm.hip
LATE_EPILOGUE = 0       # 1: lin_gemm fetches the residual after its reduction (measured slower)
WIDE_CHUNKS = 1         # 1: split-bf16 3x3 convolutions on the 512-thread tiles take 32-wide K chunks (flag bit 8); 0 = A/B
# measured (tile, ksplit) per (M, N, K, geglu), filled from tools/tile_sweep.py runs (see tile_table.py)
try:
    from .tile_table import TILE_TABLE
except ImportError:                                              # pragma: no cover
    TILE_TABLE = {}
TILE_TABLE = dict(TILE_TABLE)
try:                                   # the Stable Audio DiT's shapes, swept separately (tools/tile_sweep.py ... dit)
    from .tile_table_dit import TILE_TABLE as _DIT_TABLE
    TILE_TABLE.update(_DIT_TABLE)
except ImportError:
    pass

# Tile tables of other REGIMES than "one chain alone on the whole chip".  The regime is a property of the thread that builds an
# engine (a pipeline worker builds its lane's engines on its own thread): `with tile_regime("cus128"): ...`.
REGIME_TABLES = {}
try:
    from .tile_table_cus128 import TILE_TABLE as _CUS128
    REGIME_TABLES["cus128"] = dict(_CUS128)
except ImportError:                                              # pragma: no cover
    pass
try:                                   # lanes of 64 CUs (several edit loops side by side); not swept yet
    from .tile_table_cus64 import TILE_TABLE as _CUS64
    REGIME_TABLES["cus64"] = dict(_CUS64)
except ImportError:
    pass
# Tables for engines built under arith_mode("bf16x6"), per regime (None = whole chip): (M, N, K, geglu) -> (tile, ksplit) with
# tile >= 100 meaning "split-bf16 kernel, tile code - 100" and tile < 100 "stay on the fp32 kernel with this tile" (the sweep
# of tools/tile_sweep.py with AED_SWEEP_ARITH=bf16x6 decides per shape).  Shapes without an entry follow the rule in
# Tape.conv: fp32 choice, flagged + x6_tile() when that choice is an LDS-staged tile.  No table has been swept yet.
X6_TABLES = {}
for _reg, _mod in ((None, "tile_table_x6"), ("cus128", "tile_table_x6_cus128"), ("cus64", "tile_table_x6_cus64")):
    try:
        X6_TABLES[_reg] = dict(__import__(f"{__package__}.{_mod}", fromlist=["TILE_TABLE"]).TILE_TABLE)
    except ImportError:
        pass
# Round 6: the BATCH-1 shapes of the edit lanes -- the CFG-shared head of the batch-2 edit engine (unet.UNetEngine share=2) runs at
# batch 1 (M = 4096 / 1024 rows); without entries those shapes took the rule's fp32 lin tiles at twice the time of the swept
# split-bf16 tiles (profiles/r06_sweeps/).  A shape that also occurs at batch 2 keeps the entry of the table it was first swept for.
for _reg, _mod in (("cus128", "tile_table_x6_cus128_b1"), ("cus64", "tile_table_x6_cus64_b1")):
    try:
        for _k, _v in __import__(f"{__package__}.{_mod}", fromlist=["TILE_TABLE"]).TILE_TABLE.items():
            X6_TABLES.setdefault(_reg, {}).setdefault(_k, _v)
    except ImportError:
        pass
REGIME_CUS = {"cus128": 128, "cus64": 64}       # CUs of the stream a regime's engines run on
_regime = threading.local()


@contextlib.contextmanager
def tile_regime(name):
    """Engines built inside this context (on this thread) take their tile choices from REGIME_TABLES[name] first."""
    if name is not None and name not in REGIME_TABLES:
        raise KeyError(f"unknown tile regime {name!r} (have {sorted(REGIME_TABLES)})")
    prev = getattr(_regime, "name", None)
    _regime.name = name
    try:
        yield
    finally:
        _regime.name = prev


# ---- arithmetic of the LDS-staged GEMMs (EXPERIMENTAL, round 3) ------------------------------------------------------------
# "f32": v_mfma_f32_32x32x2_f32 on fp32 operands (the product's arithmetic, conv_gemm.hip).
# "bf16x6": every fp32 operand is cut exactly into three bf16 pieces in the kernel's loader and the six piece products of
#   relative size >= 2^-16 are accumulated in fp32 (conv_gemm_x6.hip): as close to fp64 as the fp32 chain
#   (tools/bf16_split_study.py, profiles/r03_x6_gemm.md), 1.5-1.8x faster at the inversion's batch-200 shapes.  Ops the
#   split kernel does not take (latency-regime tiles >= 10, skinny tiles 5 / 6, scalar-gather shapes) stay fp32.
# "fp8" (EXPERIMENT, BASELINE config 5's "fp8 MFMA path"; never a parity path): the flagged GEMMs quantise both operands to OCP
#   MX-FP8 (e4m3 elements, one e8m0 scale per 32 k of a row) in the loader and run v_mfma_scale_f32_32x32x64_f8f6f4
#   (conv_gemm_f8.hip); ops that kernel does not take (channel counts not a multiple of 64, ...) and attention run split-bf16.
ARITH_FLAGS = {"f32": 0, "bf16x6": 4 | 8, "fp8": 64 | 4 | 8}
FP8_ONLY = None         # fp8 EXPERIMENT: callable(op name) -> bool restricting arith_mode("fp8") to some records, the rest stay
#                         split-bf16 (tools/fp8_layer_budget.py: which GEMM family carries how much of the deviation)
DEFAULT_ARITH = "f32"   # arithmetic of engines built OUTSIDE any arith_mode context (tests/conftest.py --codec-arith sets it)


@contextlib.contextmanager
def arith_mode(name):
    """Engines built inside this context (on this thread) mark their eligible AED_OP_CONV_GEMM records with ARITH_FLAGS[name]."""
    if name not in ARITH_FLAGS:
        raise KeyError(f"unknown arithmetic {name!r} (have {sorted(ARITH_FLAGS)})")
    prev = getattr(_regime, "arith", None)
    _regime.arith = name
    try:
        yield
    finally:
        _regime.arith = prev


def x6_tile(M, N, tile, ksplit, cus=None):
    """Tile code for the split-bf16 kernel given the fp32 choice (profiles/r03_x6_gemm.md, batch-200 shapes): the 512-thread
    tiles where two of them per CU still fill the chip -- 128x256 when N is a whole number of 256-wide tiles (wide Linears:
    qkv, FF1; +3-6 % over 256x128), else 256x128 (+2-8 % over 128x128) -- otherwise the fp32 table's tile."""
    cus = cus or CU_COUNT
    if tile == 1 and N >= 128:
        ks = max(ksplit, 1)
        if N >= 512 and N % 256 == 0 and math.ceil(M / 128) * (N // 256) * ks >= 2 * cus:
            return 9
        if math.ceil(M / 256) * math.ceil(N / 128) * ks >= 2 * cus:
            return 8
    return tile


FP8_PREQUANT = 1        # fp8 experiment: 1 = weights quantised once at engine build (aed_mx_quantize_rows), 0 = in the loader (A/B)


def mx_quantized_weights(w, N, K):
    """(e4m3 bytes [N, K], e8m0 scale bytes [N, K/32]) of a weight matrix on the device, computed once per weight TENSOR OBJECT
    (cached as an attribute of it: engines of several batch shapes share the packed weights; a freed-and-reused address
    cannot alias because the cache dies with the tensor)."""
    hit = getattr(w, "_aed_mxq", None)
    if hit is not None and hit[2] == (w.data_ptr(), N, K):
        return hit[0], hit[1]
    q = torch.empty((N, K), dtype=torch.uint8, device=w.device)
    sc = torch.empty((N, K // 32), dtype=torch.uint8, device=w.device)
    with torch.cuda.device(w.device):
        L.check(L.lib().aed_mx_quantize_rows(w.data_ptr(), q.data_ptr(), sc.data_ptr(), N, K, L.current_stream_ptr()),
                "aed_mx_quantize_rows")
    try:
        w._aed_mxq = (q, sc, (w.data_ptr(), N, K))
    except AttributeError:                       # pragma: no cover
        pass
    return q, sc


class Tape:
    def __init__(self, device):
        self.device = torch.device(device)
        self.ops = []
        self.meta = []          # per-op dict(name, flops, bytes)
        self.keep = []          # tensors referenced by raw pointer
        self._arr = None
        self._ws_need = 0
        self._ws_ops = []
        self.ws = None
        self._graph = None

    # ------------------------------------------------------------------ buffers
    def alloc(self, *shape, zero=False):
        t = (torch.zeros if zero else torch.empty)(shape, device=self.device, dtype=torch.float32)
        self.keep.append(t)
        return t

    def hold(self, t):
        self.keep.append(t)
        return t

    @staticmethod
    def _ptr(t):
        if t is None:
            return None
        if isinstance(t, int):
            return t
        return t.data_ptr()

    def _add(self, code, i=(), f=(), p=(), name="", flops=0, nbytes=0, flags=0, exec_flops=None):
        op = L.aed_op()
        op.code = code
        op.flags = flags
        for k, v in enumerate(i):
            op.i[k] = int(v)
        for k, v in enumerate(f):
            op.f[k] = float(v)
        for k, v in enumerate(p):
            op.p[k] = self._ptr(v)
            if torch.is_tensor(v):
                self.keep.append(v)         # the record holds a raw pointer: keep the storage alive
        self.ops.append(op)
        # `flops` is ALGORITHMIC work (what the reference's formulation of the op costs: SURVEY 8d); ops that execute
        # fewer (the folded cross-attention) pass it explicitly; `exec_flops` is what the kernel executes (MFMA
        # utilisation is priced against THAT, path-level throughput against the algorithmic count)
        self.meta.append(dict(name=name or L.OP_NAMES[code], code=code, flops=flops, bytes=nbytes,
                              exec_flops=flops if exec_flops is None else exec_flops))
        self._arr = None
        return len(self.ops) - 1

    # ------------------------------------------------------------------ conv / linear
    # tile codes of AED_OP_CONV_GEMM (slot i29): 1/2/3/4/5/6 = LDS-staged block tiles 128x128 / 128x64 / 64x128 / 64x64 /
    # 128x32 / 32x128 (conv_gemm.hip), 10..19 = latency-regime kernels of lin_gemm.hip:
    # (waves, tile) 10 = (4, 32x32), 11 = (8, 32x32), 12 = (16, 32x32), 13 = (4, 32x64), 14 = (8, 32x64),
    # 15 = (4, 64x64), 16 = (4, 64x32), 17 = (8, 64x64), 18 = (10, 32x32), 19 = (12, 32x32)
    LIN_TILES = {10: (32, 32), 11: (32, 32), 12: (32, 32), 13: (32, 64), 14: (32, 64), 15: (64, 64), 16: (64, 32),
                 17: (64, 64), 18: (32, 32), 19: (32, 32)}

    @staticmethod
    def pick_tile(M, N, K, cus=None, vector_ok=True, geglu=False, lin_ok=True):
        """Tile / split-K choice.  Measured table first (tools/tile_sweep.py on the MI355X, shapes of the benchmark's
        U-Net at batch 2 and 2G), then the rule the table was distilled into:
        few output tiles -> lin_gemm (K split across the wavefronts of a workgroup, no reduce launch), more waves per
        tile the fewer tiles there are and the longer K is; otherwise the LDS-staged kernel with the largest tile that
        still fills the chip, split-K (deterministic slab reduce) when the grid would be < 1 block per CU."""
        cus = cus or CU_COUNT
        key = (M, N, K, int(bool(geglu)))
        reg = getattr(_regime, "name", None)
        for table in ((REGIME_TABLES[reg],) if reg else ()) + (TILE_TABLE,):
            hit = table.get(key)
            if hit is not None and (lin_ok or hit[0] < 10):
                return hit

        def blocks(bm, bn):
            return math.ceil(M / bm) * math.ceil(N / bn)
        if LIN_MODE and lin_ok and vector_ok and K % 32 == 0:
            if geglu:
                t = blocks(32, 64)
                if t <= 2 * cus:
                    return (14 if (t <= cus // 2 and K >= 512) else 13), 1
                if blocks(64, 64) <= 4 * cus:
                    return 15, 1
            else:
                t = blocks(32, 32)
                if t <= 4 * cus:
                    if t <= cus // 2 and K >= 2048:
                        return 12, 1
                    if t <= cus and K >= 512:
                        return 11, 1
                    return 10, 1
                # (beyond ~1000 32x32 tiles the LDS-staged block tiles win every measured shape: the 64x64 lin tile was
                # 10-20 % behind them at M = 5120 and is only kept for the GEGLU shapes above)
        if geglu:
            return 1, 1
        if N <= 32:
            cfg, bm, bn = 5, 128, 32
        elif M <= 32:
            cfg, bm, bn = 6, 32, 128
        elif N >= 128 and blocks(128, 128) >= cus:
            cfg, bm, bn = 1, 128, 128
        elif N >= 64 and blocks(128, 64) >= 2 * cus:
            cfg, bm, bn = 2, 128, 64
        elif N >= 128 and blocks(64, 128) >= cus:
            cfg, bm, bn = 3, 64, 128
        elif N < 64:
            cfg, bm, bn = 5, 128, 32
        else:
            cfg, bm, bn = 4, 64, 64
        nblk = blocks(bm, bn)
        nchunks = math.ceil(K / 32)
        ksplit = 1
        if nblk < cus and nchunks >= 8:
            ksplit = max(1, min(math.ceil(2 * cus / nblk), nchunks // 4, 32))
        return cfg, ksplit

    def conv(self, x, w, bias, out, *, B, IH, IW, Cin, OH, OW, N, KH=1, KW=1, stride=1, pad_h=0, pad_w=0,
             dil_h=1, dil_w=1, up=0, lda=None, a_bs=None, res=None, rowvec=None, ld_rv=0, in_act=0, in_slope=0.0,
             out_act=0, out_p=0.0, accumulate=0, out_div=1.0, o_mul=1, o_add=0, o_len=None, out_bs=None,
             ldc=None, ldr=None, ksplit=0, tile=0, ln_rowsum=None, ln_eps=1e-5, x2=None, C1=0, lda2=None, a_bs2=None,
             geglu=0, w_bs=0, vec_ld=1, vec_bs=0, sm_group=0, sm_scale=1.0, kbias=None, alg_flops=None, name="conv"):
        """Implicit-GEMM conv (AED_OP_CONV_GEMM).  x: [B,IH,IW,>=Cin] view, w: [N, KH*KW*Cin].
        x2/C1: two-source A -- channels [0,C1) of every tap come from x, [C1,Cin) from x2 (a concat that is never
        materialised).  geglu: w rows are packed [32 value | 32 gate] per 32 features and out[:, f] =
        value_f * gelu(gate_f) has N/2 columns (geglu=2: value_f * silu(gate_f), SwiGLU)."""
        M = B * OH * OW
        K = KH * KW * Cin
        lda = x.stride(-2) if lda is None else lda
        a_bs = IH * IW * lda if a_bs is None else a_bs
        ldc = out.stride(-2) if ldc is None else ldc
        if res is not None and ldr is None:
            ldr = res.stride(-2)
        o_len = OH * OW if o_len is None else o_len
        out_bs = OH * OW if out_bs is None else out_bs
        if x2 is not None:
            assert 0 < C1 < Cin and C1 % 64 == 0
            lda2 = x2.stride(-2) if lda2 is None else lda2
            a_bs2 = IH * IW * lda2 if a_bs2 is None else a_bs2
        vec_ok = (Cin % 32 == 0) and (lda % 4 == 0) and (x2 is None or lda2 % 4 == 0)
        # lin_gemm needs an unsplit epilogue without the accumulate modes (vocoder MRF) and 32-wide channel chunks
        lin_ok = vec_ok and accumulate == 0 and (x2 is None or C1 % 32 == 0)
        auto_tile, auto_split = self.pick_tile(M, N, K, vector_ok=vec_ok, geglu=bool(geglu), lin_ok=lin_ok)
        tile_forced = bool(tile)
        tile = tile or auto_tile
        ksplit = ksplit or auto_split
        if w_bs or vec_bs or sm_group or vec_ld != 1:
            # per-batch weights / grouped softmax (cross-attention as two skinny GEMMs): lin_gemm kernels only
            assert lin_ok and KH * KW == 1 and (OH * OW) % 64 == 0, "per-batch operands need the lin_gemm path"
            if tile < 10:
                tile = 10
        if tile >= 10:
            ksplit = 1
        ln_mode = 0
        if ln_rowsum is not None:
            # fused LayerNorm: `w` carries gamma, `bias` = W.beta (+bias), `ln_rowsum`[n] = sum_k w[n,k]
            assert KH * KW == 1 and rowvec is None and bias is not None and vec_ok
            ln_mode, rowvec, ksplit = 1, ln_rowsum, 1
        i = [M, N, K, lda, ldc, ldr or 0, ld_rv, IH, IW, OH, OW, Cin, KH, KW, stride, pad_h, pad_w, dil_h, dil_w, up,
             a_bs, o_mul, o_add, o_len, out_bs, in_act, out_act, accumulate, ksplit, tile, FORCE_BK, ln_mode,
             C1 if x2 is not None else 0, lda2 or 0, a_bs2 or 0, int(geglu), sm_group, w_bs, vec_ld, vec_bs]
        n_out = N // 2 if geglu else N
        flags = 2 if LATE_EPILOGUE else 0
        arith = ARITH_FLAGS[getattr(_regime, "arith", None) or DEFAULT_ARITH]
        if arith & 64 and FP8_ONLY is not None and not FP8_ONLY(name):
            arith = ARITH_FLAGS["bf16x6"]
        x6_ok = arith and vec_ok and B * a_bs + IH * IW * lda < (1 << 29) and N * K < (1 << 29) and \
            (x2 is None or B * a_bs2 + IH * IW * lda2 < (1 << 29)) and not (w_bs or vec_bs or sm_group or vec_ld != 1) and \
            self._ptr(x) % 16 == 0 and self._ptr(w) % 16 == 0      # buffer_load_dwordx4 (the launcher's `fits`)
        swept = X6_TABLES.get(getattr(_regime, "name", None), {}).get((M, N, K, int(bool(geglu)))) if x6_ok else None
        if swept is not None and not tile_forced and (swept[0] >= 100 or lin_ok):
            # measured per shape: tile >= 100 = split-bf16 kernel with tile - 100, else the fp32 kernel with that tile
            t, ks = swept
            if t >= 100:
                flags |= arith
                t -= 100
            ksplit = 1 if (t >= 10 or ln_mode) else max(ks, 1)
            i[28], i[29] = ksplit, t
        elif x6_ok and tile in (1, 2, 3, 4):
            flags |= arith
            i[29] = x6_tile(M, N, tile, ksplit)
        if flags & 4 and i[29] in (8, 9) and KH * KW > 1 and Cin % 32 == 0 and (x2 is None or C1 % 32 == 0) and WIDE_CHUNKS:
            # wide chunks (two bf16 k-blocks per LDS stage and barrier; csrc/conv_gemm_x6.hip, flag bit 8): +3-4 % on the long-K
            # 3x3 convolutions at the inversion's batch, nothing on the short-K Linears (profiles/r05_small_m.md section 4)
            flags |= 256
        if flags & 4:
            # the CU budget of the stream this engine runs on (a pipeline partition): sizes conv_gemm_x6's tile groups
            lane_cus = REGIME_CUS.get(getattr(_regime, "name", None), CU_COUNT)
            flags |= {1: 0, 2: 1, 4: 2, 8: 3}.get(max(1, CU_COUNT // max(lane_cus, 1)), 0) << 16
        idx = self._add(L.OP_CONV_GEMM, i, [in_slope, out_p, out_div, ln_eps, sm_scale],
                        [x, w, bias, out, res, rowvec, None, None, x2, kbias], name=name,
                        flops=2 * M * N * K if alg_flops is None else alg_flops, exec_flops=2 * M * N * K,
                        nbytes=4 * (B * IH * IW * Cin + N * K + M * n_out), flags=flags)
        f8_fits = x6_ok and Cin % 64 == 0 and lda % 4 == 0 and (x2 is None or C1 % 64 == 0) and i[29] in (0, 1, 2, 3, 4, 8, 9) \
            and not (i[36] or i[37] or i[39]) and i[38] <= 1      # launch_conv_gemm_f8's `fits`: anything else falls back to
        #                                                           the x6 / fp32 kernels, which read p[9] as the key bias
        if flags & 64 and FP8_PREQUANT and f8_fits and kbias is None and torch.is_tensor(w) and w.is_cuda:
            # fp8 experiment: the (frozen) weights are quantised to MX-FP8 ONCE, here, and the record points at the bytes
            q, sc = mx_quantized_weights(w, N, K)
            self.keep += [q, sc]
            self.ops[idx].p[7], self.ops[idx].p[9] = q.data_ptr(), sc.data_ptr()
            self.ops[idx].flags |= 128
        if ksplit > 1:
            self._ws_need = max(self._ws_need, ksplit * M * N)
            self._ws_ops.append(idx)
        return out

    def linear(self, x, w, bias, out, *, M, K, N, lda=None, **kw):
        """out[M,N] = x[M,K] @ w[N,K]^T (+bias ...).  x/out may be column views of wider buffers."""
        lda = x.stride(-2) if lda is None else lda
        return self.conv(x, w, bias, out, B=1, IH=M, IW=1, Cin=K, OH=M, OW=1, N=N, lda=lda, a_bs=0, **kw)

    TILE_BM = {1: 128, 2: 128, 3: 64, 4: 64, 5: 128, 6: 32}

    # ------------------------------------------------------------------ norms
    def groupnorm(self, x, gamma, beta, out, *, B, HW, C, G=32, eps=1e-5, act=0, variant=None, x2=None, C1=0,
                  name="gn"):
        """GroupNorm(+act) over [B, HW, C].  x2/C1: two-source rows -- channels [0,C1) from x, [C1,C) from x2."""
        ldx = x.stride(-2)
        ldy = out.stride(-2)
        ldx2 = x2.stride(-2) if x2 is not None else 0
        C1 = C1 if x2 is not None else 0
        variant = GN_VARIANT if variant is None else variant
        # single launch, one block per (group, batch item) -- unless a group's channels are a sliver of each row
        # (< 32 B contiguous at >= 2048 rows: every 128-byte line is fetched for 16 useful bytes and the 64 blocks crawl;
        # measured 25-48 us at the U-Net's level 0 against ~2 x 5 us for the coalesced stats + apply pair)
        sliver = (C // G) < 8 and HW >= 2048 and not GN_FORCE_SMALL
        # channels per group that are not a float4 granule (TANGO at full size: 10 and 30): the single-launch op's scalar kernel
        odd = (C // G) % 4 != 0 or ldx % 4 != 0 or ldy % 4 != 0 or (x2 is not None and (C1 % 4 != 0 or ldx2 % 4 != 0))
        if odd or (HW * (C // G) <= 65536 and B * G >= 32 and B * HW * C <= (1 << 22) and not sliver):
            # small map: one launch, one block per (group, batch item) -- latency, not bandwidth, is the cost
            # (variant 1 = second-generation kernel with batched loads, see norm.hip)
            self._add(L.OP_GN_SMALL, [B, HW, C, G, ldx, ldy, act, variant, C1, ldx2], [eps], [x, gamma, beta, out, x2],
                      name=name + ".gn1", nbytes=12 * B * HW * C)
            return out
        # stats: <=32 coarse slabs per batch item (few partials to re-reduce);
        # apply: fine slabs for parallelism (~2 blocks per CU)
        s_rpc = max(4, math.ceil(HW / 32))
        s_chunks = math.ceil(HW / s_rpc)
        a_target = max(1, (GN_APPLY_BLOCKS_PER_CU * CU_COUNT) // max(1, B))
        a_rpc = max(16, math.ceil(HW / a_target))       # (>= 16 rows per block: every block first re-reduces the batch item's partials)
        a_chunks = math.ceil(HW / a_rpc)
        part = self.alloc(B, s_chunks, G, 2)
        nb = 4 * B * HW * C
        self._add(L.OP_GN_STATS, [B, HW, C, G, ldx, s_rpc, s_chunks, C1, ldx2], [], [x, part, x2],
                  name=name + ".stats", nbytes=nb)
        self._add(L.OP_GN_APPLY, [B, HW, C, G, ldx, a_rpc, s_chunks, act, ldy, a_chunks, C1, ldx2], [eps],
                  [x, part, gamma, beta, out, x2], name=name + ".apply", nbytes=2 * nb)
        return out

    # ------------------------------------------------------------------ attention & friends
    def attention(self, q, k, v, out, *, B, H, Nq, Nk, D, ldq, ldk, ldv, ldo, bsq, bsk, bsv, bso, scale,
                  bias=None, ld_bias=0, variant=None, name="attn"):
        """variant: 0 auto (transposed-score kernel when Nk > 64), 1 single-pass kernel, 3 split-bf16 kernel (tests)."""
        variant = ATTN_VARIANT if variant is None else variant
        # under arith_mode("bf16x6") the record carries flag bit 2: the launcher then takes the split-bf16 kernel
        # (attention_x6.hip) in the throughput regime for the head dims it has (32 / 48 / 64), the fp32 kernels otherwise
        arith = ARITH_FLAGS[getattr(_regime, "arith", None) or DEFAULT_ARITH] if ATTN_X6 else 0
        # The launcher's own rule (every wave its own query tile only when that gives >= 2 workgroups per CU of the WHOLE chip)
        # keeps the edit loop's batch-2 calls on the key-split fp32 kernel.  On a CU-masked lane the split-bf16 kernel wins as
        # soon as there is a workgroup per CU of the LANE (measured, profiles/r04_attn_x6_ab.jsonl: 1024 tokens at batch 2 on a
        # 64-CU stream 86 -> 51 us, on 128 CUs 48 -> 34 us, on the whole chip 30 -> 35 us; 256 tokens: fp32 wins everywhere)
        if arith & 4 and variant == 0 and D in (32, 48, 64) and Nk > 64 and \
                self._ptr(q) % 16 == 0 and self._ptr(k) % 16 == 0 and \
                math.ceil(Nq / 128) * H * B >= REGIME_CUS.get(getattr(_regime, "name", None), CU_COUNT):
            variant = 3         # (the kernel's float4 fragment loads need 16-byte aligned q / k; a forced variant does not fall back)
        self._add(L.OP_ATTENTION, [B, H, Nq, Nk, D, ldq, ldk, ldv, ldo, ld_bias, bsq, bsk, bsv, bso, variant], [scale],
                  [q, k, v, bias, out], name=name, flops=4 * B * H * Nq * Nk * D,
                  nbytes=4 * B * H * D * (2 * Nq + 2 * Nk), flags=arith & 4)
        return out

    def xattn_fold(self, kv, xq, xs, xo, G, gs, VOt, *, B, Lk, H, C, D, name="xattn_fold"):
        """Per-prompt operands of the folded cross-attention (AED_OP_XATTN_FOLD; see elementwise.hip)."""
        self._add(L.OP_XATTN_FOLD, [B, Lk, H, C, D, kv.stride(-2)], [], [kv, xq, xs, xo, G, gs, VOt], name=name,
                  flops=0, nbytes=4 * (2 * B * H * Lk * C))

    def copy2d(self, src, dst, *, rows, cols, ld_src=None, ld_dst=None, state=None, idx_off=0, idx_mul=0,
               idx_stride=0, coef=None, c_mul=0, c_off=0, c_stride=0, c_col=0, name="copy"):
        """dst[r, :cols] = src[r, :cols]; with `state`, src is first advanced by
        (idx_off + idx_mul*state[0]) * idx_stride elements on the device (trajectory walk); with `coef`, every element
        is multiplied by coef[(state[0]*c_mul + c_off)*c_stride + c_col] (a per-step scalar of a device table)."""
        ld_src = src.stride(-2) if ld_src is None else ld_src
        ld_dst = dst.stride(-2) if ld_dst is None else ld_dst
        self._add(L.OP_COPY2D, [rows, cols, ld_src, ld_dst, idx_off, idx_mul, idx_stride, c_mul, c_off, c_stride, c_col],
                  [], [src, dst, state, coef], name=name, nbytes=8 * rows * cols)
        return dst

    @staticmethod
    def graph_capture(fn):
        """Capture everything `fn()` launches on the current stream (may run several tapes)."""
        lib = L.lib()
        sp = L.current_stream_ptr()
        L.check(lib.aed_graph_begin(sp), "aed_graph_begin")
        try:
            fn()
        finally:
            g = ctypes.c_void_p()
            rc = lib.aed_graph_end(sp, ctypes.byref(g))
        L.check(rc, "aed_graph_end")
        return g

    @staticmethod
    def graph_replay(g):
        L.check(L.lib().aed_graph_launch(g, L.current_stream_ptr()), "aed_graph_launch")

    def time_embed(self, out, *, B, dim, flip=True, shift=0.0, timesteps=None, state=None, t_imm=0, freqs=None,
                   float_table=False, name="time_embed"):
        """Sinusoidal features of a timestep read from a device table.  Default: diffusers' Timesteps (int64 table,
        log-spaced frequencies).  freqs + float_table: learned Fourier features of a continuous timestep (Stable Audio's
        time_proj: the fp32 table holds 2*pi*t, `freqs` the learned weights)."""
        half = dim // 2
        if freqs is None:
            exponent = -math.log(10000) * torch.arange(half, dtype=torch.float32) / (half - shift)
            freqs = torch.exp(exponent).to(self.device)
        self._add(L.OP_TIME_EMBED, [B, dim, int(flip), out.stride(-2), t_imm, 1, int(float_table)], [shift, 10000.0],
                  [out, timesteps, state, freqs, None], name=name)
        return out

    def softmax_rows(self, x, out, *, rows, cols, scale=1.0, name="softmax"):
        self._add(L.OP_SOFTMAX_ROWS, [rows, cols, x.stride(-2), out.stride(-2)], [scale], [x, out], name=name,
                  nbytes=8 * rows * cols)
        return out

    def transpose(self, src, dst, *, Bt, R, C, ld_src=None, ld_dst=None, bs_src=None, bs_dst=None, name="transpose"):
        ld_src = C if ld_src is None else ld_src
        ld_dst = R if ld_dst is None else ld_dst
        bs_src = R * ld_src if bs_src is None else bs_src
        bs_dst = C * ld_dst if bs_dst is None else bs_dst
        self._add(L.OP_TRANSPOSE, [Bt, R, C, ld_src, ld_dst, bs_src, bs_dst], [], [src, dst], name=name,
                  nbytes=8 * Bt * R * C)
        return dst

    def axpby(self, x, y, *, numel, a=1.0, b=0.0, name="axpby"):
        self._add(L.OP_AXPBY, [numel & 0xFFFFFFFF, numel >> 32], [a, b], [x, y], name=name, nbytes=12 * numel)
        return y

    def advance(self, state, by=1):
        self._add(L.OP_ADVANCE, [by], [], [state], name="advance")

    def step(self, code, *, xts, zs, eps_u, eps_c, cfg, coef, state, out, numel, P, T, s_imm=0, v_pred=0, flag=1,
             cfg_scalar=1.0, c_imm=(0, 0, 0, 0, 0), s_mul=1, s_off=0, name="step"):
        self._add(code, [numel & 0xFFFFFFFF, numel >> 32, P, T, s_imm, v_pred, flag, s_mul, s_off],
                  [cfg_scalar, *c_imm],
                  [xts, zs, eps_u, eps_c, cfg, coef, state, out], name=name, nbytes=4 * numel * 6)

    # ------------------------------------------------------------------ Stable Audio Open ops (csrc/stable_audio.hip)
    def rotary(self, x, cos, sin, *, M, N, H, D, R, ld=None, nsec=2, sec_stride=None, name="rotary"):
        """Rotate the first R features of every head of q (and k) inside a fused projection buffer x[M, ld], in place."""
        ld = x.stride(-2) if ld is None else ld
        sec_stride = H * D if sec_stride is None else sec_stride
        self._add(L.OP_ROTARY, [M, N, H, D, R, ld, nsec, sec_stride], [], [x, cos, sin], name=name,
                  nbytes=8 * M * nsec * H * R)
        return x

    def snake(self, x, out, a, inv_b, *, rows, C, name="snake"):
        self._add(L.OP_SNAKE, [rows & 0xFFFFFFFF, rows >> 32, C, x.stride(-2), out.stride(-2)], [], [x, out, a, inv_b],
                  name=name, nbytes=8 * rows * C)
        return out

    def sa_step(self, mode, *, xts, zs, v_u, v_c, coef, state, hist, numel, T, out=None, extra=None, fix=1, cfg=1.0,
                s_imm=0, s_mul=1, s_off=0, name="sa_step"):
        """mode 0: get_zs_from_xts, mode 1: reverse_step_with_custom_noise of the Stable Audio wrapper (see aed.h)."""
        self._add(L.OP_SA_STEP, [numel & 0xFFFFFFFF, numel >> 32, mode, T, s_imm, int(fix), s_mul, s_off], [cfg],
                  [xts, zs, v_u, v_c, coef, state, hist, out, extra], name=name, nbytes=4 * numel * 7)

    def gauss_sample(self, moments, noise, out, *, rows, C, name="gauss_sample"):
        self._add(L.OP_GAUSS_SAMPLE, [rows & 0xFFFFFFFF, rows >> 32, C, moments.stride(-2)], [], [moments, noise, out],
                  name=name, nbytes=16 * rows * C)
        return out

    def reflect_pad(self, src, dst, *, B, N, pad, ldd, name="reflect_pad"):
        self._add(L.OP_REFLECT_PAD, [B, N, pad, ldd], [], [src, dst], name=name, nbytes=8 * B * N)

    def magnitude(self, ft, mag, *, F, cut, ld_ft, ld_mag, name="magnitude"):
        self._add(L.OP_MAGNITUDE, [F, cut, ld_ft, ld_mag], [], [ft, mag], name=name, nbytes=4 * F * (2 * cut + ld_mag))

    # ------------------------------------------------------------------ execution
    def finalize(self):
        if self._arr is None:
            if self._ws_need and (self.ws is None or self.ws.numel() < self._ws_need):
                self.ws = torch.empty(self._ws_need, device=self.device, dtype=torch.float32)
            for idx in self._ws_ops:
                self.ops[idx].p[6] = self.ws.data_ptr()
            self._arr = (L.aed_op * len(self.ops))(*self.ops)
        return self._arr

    @property
    def flops(self):
        return sum(m["flops"] for m in self.meta)

    @property
    def exec_flops(self):
        return sum(m["exec_flops"] for m in self.meta)

    def run(self, start=0, end=None):
        arr = self.finalize()
        end = len(self.ops) if end is None else end
        if end <= start:
            return
        sub = ctypes.cast(ctypes.byref(arr, start * ctypes.sizeof(L.aed_op)), ctypes.POINTER(L.aed_op))
        L.check(L.lib().aed_tape_run(sub, end - start, L.current_stream_ptr()), "aed_tape_run")

    def profile(self):
        """Per-op milliseconds (HIP events on the current stream)."""
        arr = self.finalize()
        ms = (ctypes.c_float * len(self.ops))()
        L.check(L.lib().aed_tape_profile(arr, len(self.ops), L.current_stream_ptr(), ms), "aed_tape_profile")
        return list(ms)

    def capture(self):
        """Capture the whole tape into a hipGraph on the current (non-default) stream."""
        lib = L.lib()
        sp = L.current_stream_ptr()
        L.check(lib.aed_graph_begin(sp), "aed_graph_begin")
        try:
            self.run()
        finally:
            g = ctypes.c_void_p()
            rc = lib.aed_graph_end(sp, ctypes.byref(g))
        L.check(rc, "aed_graph_end")
        self._graph = g
        return g

    def replay(self):
        L.check(L.lib().aed_graph_launch(self._graph, L.current_stream_ptr()), "aed_graph_launch")

    def __del__(self):
        try:
            if self._graph is not None:
                L.lib().aed_graph_destroy(self._graph)
        except Exception:
            pass

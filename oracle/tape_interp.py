"""CPU interpreter for `aed_op` tapes.

TEST INFRASTRUCTURE (see oracle/__init__.py): an independent, plain-torch statement of what every opcode of
include/aed.h means, executed over tapes that the product's builders (unet.UNetEngine, codec.*, editing.EditEngine)
lay out on device="cpu".  It lets the `-m "not gpu"` suite check the product's HOST logic -- graph wiring, weight
packing, LayerNorm folding, skip/concat strides, device-indexed loops, timestep batching -- against the oracle and
the reference fixtures without a GPU.  It is never imported by the product; kernels are still only proven by the
`-m gpu` tests.  Slot meanings follow the "// slots:" comments next to each launcher in audioeditingcode_amd/csrc.
"""
import ctypes
import math

import numpy as np
import torch

ACT_NONE, ACT_SILU, ACT_LEAKY, ACT_TANH, ACT_LOGCLAMP = range(5)


def _view(ptr, n, ctype=ctypes.c_float, dtype=np.float32):
    """Writable torch view of n elements of raw CPU memory at `ptr`."""
    if not ptr or n <= 0:
        return None
    buf = (ctype * int(n)).from_address(int(ptr))
    return torch.from_numpy(np.frombuffer(buf, dtype=dtype))


def _f32(ptr, n):
    return _view(ptr, n)


def _act(v, code, p):
    if code == ACT_SILU:
        return v / (1.0 + torch.exp(-v))
    if code == ACT_LEAKY:
        return torch.where(v > 0, v, v * p)
    if code == ACT_TANH:
        return torch.tanh(v)
    if code == ACT_LOGCLAMP:
        return torch.log(torch.clamp(v, min=p))
    return v


def _state0(op_ptr):
    s = _view(op_ptr, 1, ctypes.c_int32, np.int32)
    return 0 if s is None else int(s[0])


# ------------------------------------------------------------------------------------------------- conv / linear
def conv_gemm(op):
    i, f, p = op.i, op.f, op.p
    (M, N, K, lda, ldc, ldr, ld_rv, IH, IW, OH, OW, Cin, KH, KW, stride, pad_h, pad_w, dil_h, dil_w, up, a_bs, o_mul,
     o_add, o_len, out_bs, in_act, out_act, accumulate) = [int(i[k]) for k in range(28)]
    ln_mode = int(i[31])
    in_slope, out_p, out_div, ln_eps = float(f[0]), float(f[1]), float(f[2]), float(f[3])
    rpb = OH * OW
    nb = M // rpb
    vIH, vIW = IH << up, IW << up
    if up:
        vIH = min(vIH, (OH - 1) * stride - 2 * pad_h + dil_h * (KH - 1) + 1)
        vIW = min(vIW, (OW - 1) * stride - 2 * pad_w + dil_w * (KW - 1) + 1)
    m = torch.arange(M)
    b, r = m // rpb, m % rpb
    oy, ox = r // OW, r % OW
    taps = torch.arange(KH * KW)
    ty, tx = taps // KW, taps % KW
    iy = (oy * stride - pad_h)[:, None] + (ty * dil_h)[None, :]                # [M, taps]
    ix = (ox * stride - pad_w)[:, None] + (tx * dil_w)[None, :]
    ok = (iy >= 0) & (iy < vIH) & (ix >= 0) & (ix < vIW)
    cy, cx = torch.where(ok, iy >> up, 0), torch.where(ok, ix >> up, 0)
    pix = b[:, None] * a_bs + (cy * IW + cx) * lda                              # element offset of channel 0
    C1, lda2, a_bs2, geglu_on = int(i[32]), int(i[33]), int(i[34]), int(i[35])

    def gather(ptr, pix_, c_lo, c_hi):
        flat = _f32(ptr, int(pix_.max()) + c_hi)
        off = pix_[:, :, None] + torch.arange(c_lo, c_hi)[None, None, :]        # [M, taps, channels]
        return torch.where(ok[:, :, None], flat[off.reshape(-1)].reshape(off.shape), torch.zeros(()))
    if C1 > 0:          # two-source A: channels [0,C1) from p0, [C1,Cin) from p8 (a concat that is never materialised)
        pix2 = b[:, None] * a_bs2 + (cy * IW + cx) * lda2
        A = torch.cat([gather(p[0], pix, 0, C1), gather(p[8], pix2, 0, Cin - C1)], 2).reshape(M, K)
    else:
        A = gather(p[0], pix, 0, Cin).reshape(M, K)
    if ln_mode:
        mean = A.sum(1) / K
        var = torch.clamp((A * A).sum(1) / K - mean * mean, min=0.0)
        rstd = 1.0 / torch.sqrt(var + ln_eps)
    if in_act:
        A = _act(A, in_act, in_slope)
    sm_group, w_bs, vec_ld, vec_bs = int(i[36]), int(i[37]), max(1, int(i[38])), int(i[39])
    batched = bool(w_bs or vec_bs or vec_ld != 1 or sm_group)
    if batched:
        # per-batch-item weights / vectors and the grouped softmax epilogue (cross-attention as two skinny GEMMs)
        Wb = _f32(p[1], (nb - 1) * w_bs + N * K).as_strided((nb, N, K), (w_bs, K, 1)).double()
        acc = torch.einsum("mk,mnk->mn", A.double(), Wb[b])
        vec = lambda ptr: None if not ptr else _f32(ptr, (nb - 1) * vec_bs + (N - 1) * vec_ld + 1).as_strided(  # noqa: E731
            (nb, N), (vec_bs, vec_ld)).double()[b]
        bias_m, sn_m = vec(p[2]), vec(p[5])
        if ln_mode:
            val = rstd.double()[:, None] * (acc - mean.double()[:, None] * sn_m) + bias_m
        else:
            val = acc if bias_m is None else acc + bias_m
        if sm_group:
            val = val * float(f[4])
            if p[9]:
                kb = _f32(p[9], nb * sm_group).reshape(nb, sm_group).double()
                val = val + kb[b].repeat(1, N // sm_group)
            val = torch.softmax(val.reshape(M, N // sm_group, sm_group), -1).reshape(M, N)
    else:
        W = _f32(p[1], N * K).reshape(N, K)
        tile = int(i[29])
        if (op.flags & 64) and Cin % 64 == 0 and lda % 4 == 0 and (tile < 5 or tile in (8, 9)) and (C1 == 0 or C1 % 64 == 0):
            # fp8 EXPERIMENT (csrc/conv_gemm_f8.hip; the launcher's `fits`): both operands as OCP MX-FP8 -- e4m3 elements, one
            # power-of-two scale per 32 k of a row, K in (tap, channel) order so a block never straddles a tap -- products and
            # scales exact, accumulation in fp64 here (fp32 with an inexact 64-term inner sum on the device: profiles/r04_f8_probes.md)
            from . import mxfp8
            acc = mxfp8.mx_dequantized(A).double() @ mxfp8.mx_dequantized(W).double().T
        else:
            acc = A.double() @ W.double().T                                     # independent of the MFMA k-order
        bias = _f32(p[2], N)
    if batched:
        pass
    elif ln_mode:
        sn = _f32(p[5], N)
        val = rstd.double()[:, None] * (acc - mean.double()[:, None] * sn.double()[None, :]) + bias.double()[None, :]
    else:
        val = acc if bias is None else acc + bias.double()[None, :]
        if p[5]:
            rv = _f32(p[5], (nb - 1) * ld_rv + N)
            val = val + rv[(b[:, None] * ld_rv + torch.arange(N)[None, :])].double()
    o = r * o_mul + o_add
    okr = (o >= 0) & (o < o_len)
    row = b * out_bs + torch.clamp(o, 0, o_len - 1)
    if geglu_on:        # packed columns: per 64 columns [32 value | 32 gate] -> 32 output features value * gelu(gate)
        v3 = val.float().reshape(M, N // 64, 2, 32)
        g = v3[:, :, 1]
        gate = g / (1.0 + torch.exp(-g)) if geglu_on == 2 else 0.5 * g * (1.0 + torch.erf(g * 0.70710678118654752440))
        val = (v3[:, :, 0] * gate).reshape(M, N // 2).double()
        N = N // 2
    cols = torch.arange(N)[None, :]
    if p[4]:
        res = _f32(p[4], int(row.max()) * ldr + N)
        val = val + res[(row[:, None] * ldr + cols)].double()
    val = _act(val.float(), out_act, out_p)
    n_c = int(row.max()) * ldc + N
    C = _f32(p[3], n_c)
    idx = (row[:, None] * ldc + cols)
    if accumulate == 1:
        val = val + C[idx]
    elif accumulate == 2:
        val = (C[idx] + val) / out_div
    sel = okr.nonzero().reshape(-1)
    C[idx[sel].reshape(-1)] = val[sel].reshape(-1)


# ------------------------------------------------------------------------------------------------- norms
def _gn_input(op, px, px2, B, HW, C, ldx, C1, ldx2):
    """[B, HW, C] input of a GroupNorm op; two-source rows (channels [0,C1) from px, the rest from px2) concatenated."""
    if not op.p[px2]:
        return _f32(op.p[px], (B * HW - 1) * ldx + C).as_strided((B, HW, C), (HW * ldx, ldx, 1))
    a = _f32(op.p[px], (B * HW - 1) * ldx + C1).as_strided((B, HW, C1), (HW * ldx, ldx, 1))
    c = _f32(op.p[px2], (B * HW - 1) * ldx2 + C - C1).as_strided((B, HW, C - C1), (HW * ldx2, ldx2, 1))
    return torch.cat([a, c], 2)


def _group_norm(x, gamma, beta, B, HW, C, G, ldx, eps):
    xs = x.double() if x.dim() == 3 else \
        x[: (B * HW - 1) * ldx + C].as_strided((B, HW, C), (HW * ldx, ldx, 1)).double()
    g = xs.reshape(B, HW, G, C // G)
    mean = g.mean(dim=(1, 3), keepdim=True)
    var = g.var(dim=(1, 3), unbiased=False, keepdim=True)
    y = ((g - mean) / torch.sqrt(var + eps)).reshape(B, HW, C)
    return (y * gamma.double() + beta.double()).float()


def _store_rows(y_flat, vals, ld):
    B, HW, C = vals.shape
    y_flat[: (B * HW - 1) * ld + C].as_strided((B, HW, C), (HW * ld, ld, 1)).copy_(vals)


def gn_stats(op):          # the partial sums are consumed only by gn_apply, which recomputes them here
    return


def gn_apply(op):
    i, p = op.i, op.p
    B, HW, C, G, ldx, act, ldy = int(i[0]), int(i[1]), int(i[2]), int(i[3]), int(i[4]), int(i[7]), int(i[8])
    x = _gn_input(op, 0, 5, B, HW, C, ldx, int(i[10]), int(i[11]))
    y = _group_norm(x, _f32(p[2], C), _f32(p[3], C), B, HW, C, G, ldx, float(op.f[0]))
    _store_rows(_f32(p[4], (B * HW - 1) * ldy + C), _act(y, act, 0.0), ldy)


def gn_small(op):
    i, p = op.i, op.p
    B, HW, C, G, ldx, ldy, act = [int(i[k]) for k in range(7)]
    x = _gn_input(op, 0, 4, B, HW, C, ldx, int(i[8]), int(i[9]))
    y = _group_norm(x, _f32(p[1], C), _f32(p[2], C), B, HW, C, G, ldx, float(op.f[0]))
    _store_rows(_f32(p[3], (B * HW - 1) * ldy + C), _act(y, act, 0.0), ldy)


def gn_scale_shift(op):
    i, p = op.i, op.p
    B, HW, C, G, ldx = [int(i[k]) for k in range(5)]
    x = _f32(p[0], (B * HW - 1) * ldx + C)[: (B * HW - 1) * ldx + C].as_strided((B, HW, C), (HW * ldx, ldx, 1)).double()
    g = x.reshape(B, HW, G, C // G)
    mean = g.mean(dim=(1, 3))
    rstd = 1.0 / torch.sqrt(g.var(dim=(1, 3), unbiased=False) + float(op.f[0]))          # [B, G]
    a = rstd.repeat_interleave(C // G, 1) * _f32(p[1], C).double()
    d = _f32(p[2], C).double() - mean.repeat_interleave(C // G, 1) * a
    _f32(p[3], B * 2 * C).reshape(B, 2, C).copy_(torch.stack([a, d], 1).float())


def attention(op):
    i, p = op.i, op.p
    B, H, Nq, Nk, D, ldq, ldk, ldv, ldo, ld_bias, bsq, bsk, bsv, bso = [int(i[k]) for k in range(14)]
    C = H * D

    def mat(ptr, n_rows, ld, bs):
        flat = _f32(ptr, (B - 1) * bs + (n_rows - 1) * ld + C)
        return flat.as_strided((B, n_rows, H, D), (bs, ld, D, 1))
    q, k, v = mat(p[0], Nq, ldq, bsq).double(), mat(p[1], Nk, ldk, bsk).double(), mat(p[2], Nk, ldv, bsv).double()
    s = torch.einsum("bqhd,bkhd->bhqk", q, k) * float(op.f[0])
    if p[3]:
        bias = _f32(p[3], (B - 1) * ld_bias + Nk).as_strided((B, Nk), (ld_bias, 1)).double()
        s = s + bias[:, None, None, :]
    o = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s, -1), v)
    mat(p[4], Nq, ldo, bso).copy_(o.float())


def copy2d(op):
    i, p = op.i, op.p
    rows, cols, lds, ldd, idx_off, idx_mul, idx_stride = [int(i[k]) for k in range(7)]
    st = _state0(p[2]) if p[2] else 0
    shift = (idx_off + idx_mul * st) * idx_stride if p[2] else 0
    src = _f32(int(p[0]) + 4 * shift, (rows - 1) * lds + cols).as_strided((rows, cols), (lds, 1)).clone()
    if p[3]:
        c_mul, c_off, c_stride, c_col = [int(i[k]) for k in range(7, 11)]
        src = src * _f32(int(p[3]) + 4 * ((st * c_mul + c_off) * c_stride + c_col), 1)[0]
    _f32(p[1], (rows - 1) * ldd + cols).as_strided((rows, cols), (ldd, 1)).copy_(src)


def time_embed(op):
    i, p = op.i, op.p
    B, dim, flip, ld, t_imm, tgroup = [int(i[k]) for k in range(6)]
    tgroup = max(1, tgroup)
    half = dim // 2
    s = _state0(p[2]) if p[2] else 0
    if p[1]:
        tidx = _view(p[4], B, ctypes.c_int32, np.int32) if p[4] else torch.zeros(B, dtype=torch.int32)
        if int(i[6]):
            table = _view(p[1], s * tgroup + int(tidx.max()) + 1)
        else:
            table = _view(p[1], s * tgroup + int(tidx.max()) + 1, ctypes.c_int64, np.int64)
        t = table[s * tgroup + tidx.long()].float()
    else:
        t = torch.full((B,), float(t_imm))
    if p[3]:
        fr = _f32(p[3], half)
    else:
        fr = torch.exp(-math.log(float(op.f[1])) * torch.arange(half, dtype=torch.float32) / (half - float(op.f[0])))
    arg = t[:, None] * fr[None, :]
    sv, cv = torch.sin(arg), torch.cos(arg)
    out = _f32(p[0], (B - 1) * ld + dim).as_strided((B, dim), (ld, 1))
    out.copy_(torch.cat([cv, sv], 1) if flip else torch.cat([sv, cv], 1))


def softmax_rows(op):
    i, p = op.i, op.p
    rows, cols, ldx, ldy = [int(i[k]) for k in range(4)]
    x = _f32(p[0], (rows - 1) * ldx + cols).as_strided((rows, cols), (ldx, 1)).double() * float(op.f[0])
    _f32(p[1], (rows - 1) * ldy + cols).as_strided((rows, cols), (ldy, 1)).copy_(torch.softmax(x, -1).float())


def transpose(op):
    i, p = op.i, op.p
    Bt, R, C, lds, ldd, bss, bsd = [int(i[k]) for k in range(7)]
    src = _f32(p[0], (Bt - 1) * bss + (R - 1) * lds + C).as_strided((Bt, R, C), (bss, lds, 1))
    _f32(p[1], (Bt - 1) * bsd + (C - 1) * ldd + R).as_strided((Bt, C, R), (bsd, ldd, 1)).copy_(src.transpose(1, 2))


def axpby(op):
    n = (int(op.i[0]) & 0xFFFFFFFF) | ((int(op.i[1]) & 0xFFFFFFFF) << 32)
    a, b = float(op.f[0]), float(op.f[1])
    x, y = _f32(op.p[0], n), _f32(op.p[1], n)
    y.copy_(a * x if b == 0.0 else a * x + b * y)


# ------------------------------------------------------------------------------------------------- step math (K1)
def _step_common(op):
    i, f, p = op.i, op.f, op.p
    numel = (int(i[0]) & 0xFFFFFFFF) | ((int(i[1]) & 0xFFFFFFFF) << 32)
    P, T, s_imm, v_pred, flag = int(i[2]), int(i[3]), int(i[4]), int(i[5]), int(i[6])
    s_mul, s_off = (int(i[7]) if int(i[7]) > 0 else 1), int(i[8])
    s = _state0(p[6]) * s_mul + s_off if p[6] else s_imm
    c = _f32(int(p[5]) + 4 * 8 * s, 8) if p[5] else torch.tensor([float(f[1 + k]) for k in range(5)])
    eps_u = _f32(p[2], numel)
    eps = eps_u
    if p[3]:
        eps_c = _f32(p[3], P * numel).reshape(P, numel)
        acc = None
        for j in range(P):
            g = _f32(p[4], P * numel).reshape(P, numel)[j] if p[4] else float(f[0])
            term = g * (eps_c[j] - eps_u)
            acc = term if acc is None else acc + term
        eps = eps_u + acc
    return numel, T, s, v_pred, flag, c, eps


def _x0_dir(x, eps, c, v_pred):
    if not v_pred:
        return (x - c[0] * eps) / c[1], eps
    return c[1] * x - c[0] * eps, c[1] * eps + c[0] * x


def invert_step(op):
    numel, T, s, v_pred, fix, c, eps = _step_common(op)
    idx = T - s - 1
    base = int(op.p[0])
    xt = _f32(base + 4 * (idx + 1) * numel, numel)
    xtm1 = _f32(base + 4 * idx * numel, numel)
    z = _f32(int(op.p[1]) + 4 * idx * numel, numel)
    x0, d = _x0_dir(xt, eps, c, v_pred)
    mu = c[2] * x0 + c[3] * d
    zz = (xtm1 - mu) / c[4]
    z.copy_(zz)
    if fix:
        xtm1.copy_(mu + c[4] * zz)
    if op.p[7]:
        _f32(op.p[7], numel).copy_(eps)


def reverse_step(op):
    numel, T, s, v_pred, has_noise, c, eps = _step_common(op)
    xt = _f32(op.p[0], numel)
    x0, d = _x0_dir(xt, eps, c, v_pred)
    prev = c[2] * x0 + c[3] * d
    if has_noise:
        z = _f32(int(op.p[1]) + 4 * (T - s - 1) * numel, numel) if T > 0 else _f32(op.p[1], numel)
        prev = prev + c[4] * z
    _f32(op.p[7], numel).copy_(prev)


def advance(op):
    st = _view(op.p[0], 1, ctypes.c_int32, np.int32)
    st[0] += int(op.i[0]) if int(op.i[0]) else 1


# ------------------------------------------------------------------------------------------------- STFT helpers
def reflect_pad(op):
    B, N, pad, ldd = [int(op.i[k]) for k in range(4)]
    src = _f32(op.p[0], B * N).reshape(B, N)
    j = torch.arange(N + 2 * pad) - pad
    j = torch.where(j < 0, -j, j)
    j = torch.where(j >= N, 2 * (N - 1) - j, j)
    _f32(op.p[1], (B - 1) * ldd + N + 2 * pad).as_strided((B, N + 2 * pad), (ldd, 1)).copy_(src[:, j])


def magnitude(op):
    F, cut, ldf, ldm = [int(op.i[k]) for k in range(4)]
    ft = _f32(op.p[0], (F - 1) * ldf + 2 * cut).as_strided((F, 2 * cut), (ldf, 1))
    mag = _f32(op.p[1], F * ldm).reshape(F, ldm)
    mag.zero_()
    mag[:, :cut] = torch.sqrt(ft[:, :cut] ** 2 + ft[:, cut:] ** 2)


def xattn_fold(op):
    B, Lk, H, C, D, ldkv = [int(op.i[k]) for k in range(6)]
    HL = H * Lk
    kv = _f32(op.p[0], (B * Lk - 1) * ldkv + 2 * C).as_strided((B, Lk, 2 * C), (Lk * ldkv, ldkv, 1)).double()
    k, v = kv[..., :C].reshape(B, Lk, H, D), kv[..., C:].reshape(B, Lk, H, D)
    xq = _f32(op.p[1], H * C * D).reshape(H, C, D).double()
    xs = _f32(op.p[2], H * 2 * D).reshape(H, 2, D).double()
    xo = _f32(op.p[3], H * C * D).reshape(H, C, D).double()
    _f32(op.p[4], B * HL * C).reshape(B, H, Lk, C).copy_(torch.einsum("bjhd,hcd->bhjc", k, xq).float())
    _f32(op.p[5], B * HL * 2).reshape(B, H, Lk, 2).copy_(torch.einsum("bjhd,hsd->bhjs", k, xs).float())
    _f32(op.p[6], B * C * HL).reshape(B, C, H, Lk).copy_(torch.einsum("bjhd,hcd->bchj", v, xo).float())


def nop(op):
    return


# ------------------------------------------------------------------------------------------------- Stable Audio Open ops
def rotary(op):
    i, p = op.i, op.p
    M, N, H, D, R, ld, nsec, sec_stride = [int(i[k]) for k in range(8)]
    hr = R // 2
    x = _f32(p[0], (M - 1) * ld + (nsec - 1) * sec_stride + H * D)
    ct, st = _f32(p[1], N * hr).reshape(N, hr), _f32(p[2], N * hr).reshape(N, hr)
    pos = torch.arange(M) % N
    for sec in range(nsec):
        v = x.as_strided((M, H, D), (ld, D, 1), sec * sec_stride)
        re, im = v[:, :, :hr].clone(), v[:, :, hr:R].clone()
        cs, sn = ct[pos][:, None, :], st[pos][:, None, :]
        v[:, :, :hr] = re * cs + (-im) * sn
        v[:, :, hr:R] = im * cs + re * sn


def snake(op):
    i, p = op.i, op.p
    rows = int(i[0]) & 0xFFFFFFFF | (int(i[1]) << 32)
    C, ldx, ldy = int(i[2]), int(i[3]), int(i[4])
    x = _f32(p[0], (rows - 1) * ldx + C).as_strided((rows, C), (ldx, 1))
    a, ib = _f32(p[2], C), _f32(p[3], C)
    out = x + ib[None] * torch.sin(a[None] * x).pow(2)
    _f32(p[1], (rows - 1) * ldy + C).as_strided((rows, C), (ldy, 1)).copy_(out)


def _sa_update(x, d, m1, c, zz):
    k1, k2, k3, inv_r0 = c[3], c[4], c[5], c[6]
    if c[7] > 1.5:
        D1 = inv_r0 * (d - m1)
        return ((k1 * x + k2 * d) + (0.5 * k2) * D1) + k3 * zz
    return (k1 * x + k2 * d) + k3 * zz


def _sa_step_math(mode, x, xm1, u, vc, cfg, c, hist, z_in, fix):
    """Shared by the tape op and the FakeLib entry points: returns (z, new x_{t-1} or None, m1, d)."""
    v = u + cfg * (vc - u) if vc is not None else u
    d = c[1] * x + c[2] * v
    m1 = hist.clone()
    k1, k2, k3, inv_r0 = c[3], c[4], c[5], c[6]
    if mode == 0:
        if c[8] > 0.5:
            zz = torch.zeros_like(x)
        elif c[7] > 1.5:
            zz = (((xm1 - k1 * x) - k2 * d) - (0.5 * k2) * (inv_r0 * (d - m1))) / k3
        else:
            zz = ((xm1 - k1 * x) - k2 * d) / k3
        return zz, (_sa_update(x, d, m1, c, zz) if fix else None), m1, d
    zz = z_in if z_in is not None else torch.zeros_like(x)
    return zz, _sa_update(x, d, m1, c, zz), m1, d


def sa_step(op):
    i, p = op.i, op.p
    numel = int(i[0]) & 0xFFFFFFFF | (int(i[1]) << 32)
    mode, T, s_imm, fix, s_mul, s_off = [int(i[k]) for k in range(2, 8)]
    s = _state0(p[5]) * max(1, s_mul) + s_off if p[5] else s_imm
    c = _f32(int(p[4]) + 4 * 12 * s, 12)
    u = _f32(p[2], numel)
    vc = _f32(p[3], numel) if p[3] else None
    hist = _f32(p[6], numel)
    cfg = torch.tensor(float(op.f[0]))
    if mode == 0:
        idx = T - s - 1
        x = _f32(int(p[0]) + 4 * (idx + 1) * numel, numel)
        xm1 = _f32(int(p[0]) + 4 * idx * numel, numel)
        zz, nx, m1, d = _sa_step_math(0, x, xm1, u, vc, cfg, c, hist, None, fix)
        _f32(int(p[1]) + 4 * idx * numel, numel).copy_(zz)
        if nx is not None:
            xm1.copy_(nx)
        if p[8]:
            _f32(int(p[8]) + 4 * idx * numel, numel).copy_(m1)
    else:
        x = _f32(p[0], numel)
        z_in = None
        if p[1]:
            z_in = _f32(int(p[1]) + 4 * (T - s - 1) * numel, numel) if T > 0 else _f32(p[1], numel)
        _, nx, m1, d = _sa_step_math(1, x.clone(), None, u, vc, cfg, c, hist, z_in, 1)
        _f32(p[7], numel).copy_(nx)
    hist.copy_(d)


def gauss_sample(op):
    i, p = op.i, op.p
    rows = int(i[0]) & 0xFFFFFFFF | (int(i[1]) << 32)
    C, ldm = int(i[2]), int(i[3])
    mom = _f32(p[0], (rows - 1) * ldm + 2 * C).as_strided((rows, 2 * C), (ldm, 1))
    noise = _f32(p[1], rows * C).reshape(rows, C)
    out = mom[:, :C] + (torch.nn.functional.softplus(mom[:, C:]) + 1e-4) * noise
    _f32(p[2], rows * C).reshape(rows, C).copy_(out)


DISPATCH = {0: nop, 1: conv_gemm, 2: gn_stats, 3: gn_apply, 5: attention, 7: copy2d,
            8: time_embed, 9: softmax_rows, 10: transpose, 11: axpby, 12: invert_step, 13: reverse_step,
            14: reverse_step, 15: advance, 16: reflect_pad, 17: magnitude, 18: transpose, 19: transpose, 20: nop,
            21: gn_scale_shift, 22: gn_small, 23: xattn_fold, 24: rotary, 25: snake, 26: sa_step, 27: gauss_sample}


def run_tape(tape, start=0, end=None):
    """Drop-in for Tape.run on a tape built with device="cpu"."""
    tape.finalize()
    end = len(tape.ops) if end is None else end
    with torch.no_grad():
        for op in tape.ops[start:end]:
            DISPATCH[int(op.code)](op)


# ------------------------------------------------------------------------------------------------- C-ABI stand-ins
class FakeLib:
    """Stand-in for the ctypes handle of libaed.so in CPU tests: the path-level entry points of include/aed.h that the
    wrapper calls directly (not through a tape), stated in plain torch over raw CPU pointers."""

    @staticmethod
    def _ptr(v):
        return v.value if isinstance(v, ctypes.c_void_p) else v

    def aed_get_zs_from_xts(self, xt, xtm1, eps_u, eps_c, cfg, cfg_scalar, n_prompts, coef_host, v_pred, fix, z,
                            noise_pred_out, numel, stream):
        c = torch.tensor([float(coef_host[k]) for k in range(5)])
        x, xm1, eps = _f32(xt, numel), _f32(xtm1, numel), _f32(eps_u, numel)
        assert not eps_c, "the wrapper passes the combined prediction"
        x0, d = _x0_dir(x, eps, c, v_pred)
        mu = c[2] * x0 + c[3] * d
        zz = (xm1 - mu) / c[4]
        _f32(z, numel).copy_(zz)
        if fix:
            xm1.copy_(mu + c[4] * zz)
        return 0

    def aed_reverse_step_with_custom_noise(self, xt, eps_u, eps_c, cfg, cfg_scalar, n_prompts, coef_host, v_pred, z,
                                           prev_out, numel, stream):
        c = torch.tensor([float(coef_host[k]) for k in range(5)])
        x0, d = _x0_dir(_f32(xt, numel), _f32(eps_u, numel), c, v_pred)
        prev = c[2] * x0 + c[3] * d
        if z:
            prev = prev + c[4] * _f32(z, numel)
        _f32(prev_out, numel).copy_(prev)
        return 0

    def aed_sa_get_zs_from_xts(self, xt, xtm1, v_u, v_c, cfg_scalar, coef_host, hist, fix, z, extra_out, numel, stream):
        c = torch.tensor([float(coef_host[k]) for k in range(12)])
        p = self._ptr
        xm1, h = _f32(p(xtm1), numel), _f32(p(hist), numel)
        zz, nx, m1, d = _sa_step_math(0, _f32(p(xt), numel), xm1, _f32(p(v_u), numel),
                                      _f32(p(v_c), numel) if p(v_c) else None, torch.tensor(float(cfg_scalar)), c, h,
                                      None, fix)
        _f32(p(z), numel).copy_(zz)
        if nx is not None:
            xm1.copy_(nx)
        if p(extra_out):
            _f32(p(extra_out), numel).copy_(m1)
        h.copy_(d)
        return 0

    def aed_sa_reverse_step_with_custom_noise(self, xt, v_u, v_c, cfg_scalar, coef_host, hist, z, prev_out, numel, stream):
        c = torch.tensor([float(coef_host[k]) for k in range(12)])
        p = self._ptr
        h = _f32(p(hist), numel)
        _, nx, m1, d = _sa_step_math(1, _f32(p(xt), numel).clone(), None, _f32(p(v_u), numel),
                                     _f32(p(v_c), numel) if p(v_c) else None, torch.tensor(float(cfg_scalar)), c, h,
                                     _f32(p(z), numel) if p(z) else None, 1)
        _f32(p(prev_out), numel).copy_(nx)
        h.copy_(d)
        return 0

    def aed_last_error(self):
        return b""

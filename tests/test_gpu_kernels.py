"""GPU parity of the individual HIP kernels against plain torch fp32 (CPU) references."""
import math

import numpy as np
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from audioeditingcode_amd import _lib as L          # noqa: E402
from audioeditingcode_amd.tape import Tape          # noqa: E402

DEV = "cuda:0"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def run(tp):
    tp.run()
    torch.cuda.synchronize()


@pytest.mark.parametrize("B,C,H,W,N,k,stride,pad,up", [
    (2, 32, 16, 8, 64, 3, 1, 1, 0),      # plain 3x3
    (1, 64, 12, 6, 96, 3, 2, 1, 0),      # stride-2 downsample
    (2, 32, 8, 4, 32, 3, 1, 1, 1),       # nearest-2x upsample fused
    (1, 128, 64, 16, 128, 3, 1, 1, 0),   # level-0-like
    (2, 8, 16, 8, 32, 3, 1, 1, 0),       # Cin=8 generic path (conv_in)
    (1, 1, 20, 12, 32, 3, 1, 1, 0),      # Cin=1 generic path (VAE conv_in)
    (2, 64, 8, 4, 8, 3, 1, 1, 0),        # N=8 (conv_out)
    (3, 96, 5, 3, 160, 1, 1, 0, 0),      # 1x1, ragged M
    (2, 640, 4, 2, 640, 3, 1, 1, 0),     # deep K, tiny M -> split-K
])
def test_conv_gemm_vs_torch(B, C, H, W, N, k, stride, pad, up):
    x = rnd(B, C, H, W, seed=1)
    w = rnd(N, C, k, k, seed=2, scale=1 / math.sqrt(C * k * k))
    b = rnd(N, seed=3, scale=0.1)
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if up else x
    ref = F.conv2d(xin, w, b, stride=stride, padding=pad)
    OH, OW = ref.shape[2], ref.shape[3]
    tp = Tape(DEV)
    xd = nhwc(x).to(DEV)
    wd = w.permute(0, 2, 3, 1).reshape(N, -1).contiguous().to(DEV)
    out = tp.alloc(B, OH, OW, N)
    tp.conv(xd, wd, b.to(DEV), out, B=B, IH=H, IW=W, Cin=C, OH=OH, OW=OW, N=N, KH=k, KW=k, stride=stride, pad_h=pad,
            pad_w=pad, up=up)
    run(tp)
    got = nchw(out.cpu())
    err = (got - ref).abs().max().item()
    assert err < 2e-5 * max(1.0, ref.abs().max().item()), err


def test_conv_asymmetric_pad_stride2():
    """VAE Downsample: F.pad (0,1,0,1) then 3x3 stride 2 pad 0 (variational_autoencoder/modules.py:85-100)."""
    x, w, b = rnd(1, 32, 17, 10, seed=4), rnd(32, 32, 3, 3, seed=5, scale=0.06), rnd(32, seed=6)
    ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, b, stride=2)
    OH, OW = ref.shape[2:]
    tp = Tape(DEV)
    out = tp.alloc(1, OH, OW, 32)
    tp.conv(nhwc(x).to(DEV), w.permute(0, 2, 3, 1).reshape(32, -1).contiguous().to(DEV), b.to(DEV), out, B=1, IH=17,
            IW=10, Cin=32, OH=OH, OW=OW, N=32, KH=3, KW=3, stride=2, pad_h=0, pad_w=0)
    run(tp)
    assert (nchw(out.cpu()) - ref).abs().max() < 2e-5


def test_linear_epilogues():
    M, K, N = 200, 96, 72
    x, w, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.1), rnd(N, seed=3), rnd(M, N, seed=4)
    rv = rnd(4, N, seed=5)
    tp = Tape(DEV)
    out = tp.alloc(M, N)
    tp.conv(x.to(DEV), w.to(DEV), b.to(DEV), out, B=4, IH=50, IW=1, Cin=K, OH=50, OW=1, N=N, res=r.to(DEV),
            rowvec=rv.to(DEV), ld_rv=N, in_act=L.ACT_SILU, out_act=L.ACT_SILU)
    run(tp)
    ref = F.silu(F.linear(F.silu(x), w, b) + r + rv.repeat_interleave(50, 0))
    assert (out.cpu() - ref).abs().max() < 3e-5


@pytest.mark.parametrize("C,G,HW,B,act", [(128, 32, 4096, 2, 1), (384, 32, 256, 2, 0), (640, 32, 64, 3, 1),
                                          (1280, 32, 64, 1, 1), (32, 8, 100, 2, 1), (512, 32, 4096, 1, 1)])
def test_groupnorm(C, G, HW, B, act):
    _groupnorm_case(C, G, HW, B, act, variant=0)


@pytest.mark.parametrize("C,G,HW,B,act", [(320, 32, 4096, 2, 1), (320, 32, 64, 3, 0), (96, 16, 300, 2, 1)])
def test_groupnorm_channel_groups_that_are_not_float4_granules(C, G, HW, B, act):
    """TANGO at full size: 320 channels in 32 groups = 10 per group (round 5: found by the full-size TANGO test)."""
    _groupnorm_case(C, G, HW, B, act, variant=0)


@pytest.mark.parametrize("C,G,HW,B,act", [(256, 32, 1000, 2, 0), (256, 32, 1021, 2, 1)])
def test_groupnorm_ragged_rows(C, G, HW, B, act):
    """row counts that are not a multiple of the batched-load width of the single-launch kernel"""
    _groupnorm_case(C, G, HW, B, act, variant=0)


def _groupnorm_case(C, G, HW, B, act, variant):
    x = rnd(B, C, HW, 1, seed=1) * 2 + 0.5
    ga, be = 1 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3)
    ref = F.group_norm(x, G, ga, be, 1e-5)
    ref = F.silu(ref) if act else ref
    tp = Tape(DEV)
    out = tp.alloc(B, HW, 1, C)
    tp.groupnorm(nhwc(x).to(DEV), ga.to(DEV), be.to(DEV), out, B=B, HW=HW, C=C, G=G, eps=1e-5, act=act, variant=variant)
    run(tp)
    assert (nchw(out.cpu()) - ref).abs().max() < 2e-5


def test_retired_opcodes_fail_loudly():
    """ABI v4: the standalone LayerNorm / GEGLU kernels are gone (fused into conv_gemm); their opcodes return an error
    instead of silently doing nothing."""
    from audioeditingcode_amd import _lib as L
    for code in (L.OP_LAYERNORM, L.OP_GEGLU):
        op = L.aed_op()
        op.code = code
        assert L.lib().aed_launch(op, None) != 0
        assert b"retired" in L.lib().aed_last_error()


@pytest.mark.parametrize("B,H,Nq,Nk,D,masked", [(2, 8, 1024, 1024, 32, False), (2, 8, 256, 256, 48, False),
                                                 (2, 8, 64, 64, 80, False), (2, 8, 256, 8, 48, False),
                                                 (2, 4, 100, 19, 32, True), (1, 2, 70, 130, 64, True),
                                                 (1, 2, 33, 5, 16, False)])
def test_attention(B, H, Nq, Nk, D, masked):
    _attention_case(B, H, Nq, Nk, D, masked, variant=0)


@pytest.mark.parametrize("B,H,Nq,Nk,D,masked", [(1, 2, 40, 200, 16, False), (1, 1, 33, 97, 80, True)])
def test_attention_ragged_long_keys(B, H, Nq, Nk, D, masked):
    """long-key kernels with key / query counts that are not multiples of their tiles"""
    _attention_case(B, H, Nq, Nk, D, masked, variant=0)


@pytest.mark.parametrize("B,H,Nq,Nk,D,masked", [(16, 8, 512, 512, 32, False), (32, 8, 256, 256, 48, True),
                                                 (40, 8, 128, 100, 64, True)])
def test_attention_transposed_kernel_throughput_mode(B, H, Nq, Nk, D, masked):
    """enough workgroups that every wave takes its own query tile (KSPLIT = 1; U-Net batch 2G of the inversion)"""
    _attention_case(B, H, Nq, Nk, D, masked, variant=0)




def _attention_case(B, H, Nq, Nk, D, masked, variant):
    C = H * D
    q, k, v = rnd(B, Nq, C, seed=1), rnd(B, Nk, C, seed=2), rnd(B, Nk, C, seed=3)
    bias = None
    if masked:
        m = (rnd(B, Nk, seed=4) > -0.3).float()
        m[:, 0] = 1
        bias = (1 - m) * -10000.0
    qh, kh, vh = (t.view(B, -1, H, D).transpose(1, 2) for t in (q, k, v))
    s = qh @ kh.transpose(-1, -2) * D ** -0.5
    if bias is not None:
        s = s + bias[:, None, None, :]
    ref = (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, Nq, C)
    tp = Tape(DEV)
    out = tp.alloc(B, Nq, C)
    tp.attention(q.to(DEV), k.to(DEV), v.to(DEV), out, B=B, H=H, Nq=Nq, Nk=Nk, D=D, ldq=C, ldk=C, ldv=C, ldo=C,
                 bsq=Nq * C, bsk=Nk * C, bsv=Nk * C, bso=Nq * C, scale=D ** -0.5,
                 bias=None if bias is None else bias.to(DEV), ld_bias=Nk, variant=variant)
    run(tp)
    assert (out.cpu() - ref).abs().max() < 2e-5


def test_misc_elementwise():
    tp = Tape(DEV)
    src = rnd(3, 40, 24, seed=1)
    dst = tp.alloc(3, 24, 40)
    tp.transpose(src.to(DEV), dst, Bt=3, R=40, C=24)
    x = rnd(50, 300, seed=2)
    sm = tp.alloc(50, 300)
    tp.softmax_rows(x.to(DEV), sm, rows=50, cols=300, scale=0.5)
    wav = rnd(2, 1000, seed=3)
    pad = tp.alloc(2, 1000 + 64)
    tp.reflect_pad(wav.to(DEV), pad, B=2, N=1000, pad=32, ldd=1064)
    te = tp.alloc(2, 128)
    tp.time_embed(te, B=2, dim=128, t_imm=501)
    run(tp)
    assert torch.equal(dst.cpu(), src.transpose(1, 2))
    assert (sm.cpu() - torch.softmax(x * 0.5, -1)).abs().max() < 1e-6
    assert torch.equal(pad.cpu(), F.pad(wav[:, None], (32, 32), mode="reflect")[:, 0])
    from oracle.unet import timestep_embedding
    assert (te.cpu() - timestep_embedding(torch.tensor([501, 501]), 128)).abs().max() < 2e-5


def test_step_math_bit_exact_vs_reference_vectors(golden_dir):
    """K1 against vectors produced by the reference's own get_zs_from_xts / reverse_step (bit-exact)."""
    import os, ctypes
    g = np.load(os.path.join(golden_dir, "step_math_T200.npz"))
    lib = L.lib()
    for i in range(int(g["n"])):
        # the scheduler table is host arithmetic and differs in the last bit between CPUs (torch
        # vectorisation); the fixture carries the coefficients of the machine that ran the reference
        cf = (ctypes.c_float * 8)(*g[f"coef{i}"].tolist())
        xt, xtm1, eps = (torch.from_numpy(g[f"{n}{i}"]).to(DEV) for n in ("xt", "xtm1", "eps"))
        z = torch.empty_like(xt)
        L.check(lib.aed_get_zs_from_xts(xt.data_ptr(), xtm1.data_ptr(), eps.data_ptr(), None, None, 0.0, 0, cf, 0, 1,
                                        z.data_ptr(), None, xt.numel(), L.current_stream_ptr()))
        prev = torch.empty_like(xt)
        zin = torch.from_numpy(g[f"z_in{i}"]).to(DEV)
        L.check(lib.aed_reverse_step_with_custom_noise(xt.data_ptr(), eps.data_ptr(), None, None, 0.0, 0, cf, 0,
                                                       zin.data_ptr(), prev.data_ptr(), xt.numel(),
                                                       L.current_stream_ptr()))
        torch.cuda.synchronize()
        np.testing.assert_array_equal(z.cpu().numpy(), g[f"z{i}"])
        np.testing.assert_array_equal(xtm1.cpu().numpy(), g[f"xfix{i}"])
        np.testing.assert_array_equal(prev.cpu().numpy(), g[f"prev{i}"])


@pytest.mark.parametrize("H,W,th,tw", [(5, 2, 9, 4), (9, 4, 17, 7), (4, 3, 8, 6)])
def test_conv_nearest_resize_to_explicit_size(H, W, th, tw):
    """Upsample2D with output_size (forward_upsample_size path): nearest resize to 2x or 2x-1, then 3x3 conv."""
    B, C, N = 2, 32, 64
    x, w, b = rnd(B, C, H, W, seed=1), rnd(N, C, 3, 3, seed=2, scale=(C * 9) ** -0.5), rnd(N, seed=3)
    ref = F.conv2d(F.interpolate(x, size=(th, tw), mode="nearest"), w, b, padding=1)
    tp = Tape(DEV)
    out = tp.alloc(B, th, tw, N)
    tp.conv(nhwc(x).to(DEV), w.permute(0, 2, 3, 1).reshape(N, -1).contiguous().to(DEV), b.to(DEV), out, B=B, IH=H,
            IW=W, Cin=C, OH=th, OW=tw, N=N, KH=3, KW=3, pad_h=1, pad_w=1, up=1)
    run(tp)
    assert (nchw(out.cpu()) - ref).abs().max() < 3e-5


# --------------------------------------------------------------------------------- lin_gemm (latency-regime kernels)
LIN_TILES = [10, 11, 12, 13, 14, 15, 16, 17, 18, 19]


@pytest.mark.parametrize("tile", LIN_TILES)
@pytest.mark.parametrize("B,C,H,W,N,k,stride,pad,up", [
    (2, 64, 16, 8, 96, 1, 1, 0, 0),      # linear-like 1x1, N not a tile multiple
    (2, 32, 16, 8, 64, 3, 1, 1, 0),      # 3x3 with zero padding taps
    (1, 64, 12, 6, 96, 3, 2, 1, 0),      # stride-2 downsample
    (2, 32, 8, 4, 32, 3, 1, 1, 1),       # nearest-2x upsample fused
    (3, 96, 5, 3, 160, 1, 1, 0, 0),      # ragged M (45 rows), K = 96 = 3 chunks over up to 16 waves
    (2, 640, 4, 2, 640, 3, 1, 1, 0),     # deep K (5760), tiny M: the shape that used to need split-K + reduce
])
def test_lin_gemm_vs_torch(tile, B, C, H, W, N, k, stride, pad, up):
    x = rnd(B, C, H, W, seed=1)
    w = rnd(N, C, k, k, seed=2, scale=1 / math.sqrt(C * k * k))
    b = rnd(N, seed=3, scale=0.1)
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if up else x
    ref = F.conv2d(xin, w, b, stride=stride, padding=pad)
    OH, OW = ref.shape[2], ref.shape[3]
    tp = Tape(DEV)
    out = tp.alloc(B, OH, OW, N)
    tp.conv(nhwc(x).to(DEV), w.permute(0, 2, 3, 1).reshape(N, -1).contiguous().to(DEV), b.to(DEV), out, B=B, IH=H, IW=W,
            Cin=C, OH=OH, OW=OW, N=N, KH=k, KW=k, stride=stride, pad_h=pad, pad_w=pad, up=up, tile=tile)
    assert tp.ops[-1].i[29] == tile
    run(tp)
    err = (nchw(out.cpu()) - ref).abs().max().item()
    assert err < 2e-5 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("tile", LIN_TILES)
def test_lin_gemm_epilogues(tile):
    """bias + per-batch row vector + residual + loader/epilogue activations, row-strided operands."""
    M, K, N = 200, 96, 72
    x, w, b, r = rnd(M, K + 32, seed=1), rnd(N, K, seed=2, scale=0.1), rnd(N, seed=3), rnd(M, N + 8, seed=4)
    rv = rnd(4, N, seed=5)
    tp = Tape(DEV)
    outw = tp.alloc(M, N + 16, zero=True)
    xd, rd = x.to(DEV), r.to(DEV)
    tp.conv(xd[:, :K], w.to(DEV), b.to(DEV), outw[:, :N], B=4, IH=50, IW=1, Cin=K, OH=50, OW=1, N=N, res=rd[:, :N],
            rowvec=rv.to(DEV), ld_rv=N, in_act=L.ACT_SILU, out_act=L.ACT_SILU, tile=tile)
    run(tp)
    ref = F.silu(F.linear(F.silu(x[:, :K]), w, b) + r[:, :N] + rv.repeat_interleave(50, 0))
    assert (outw.cpu()[:, :N] - ref).abs().max() < 3e-5
    assert outw.cpu()[:, N:].abs().max() == 0          # nothing written past the N columns


def _ln_fold(w, gamma, beta, bias):
    wf = (w.double() * gamma.double()[None, :]).float()
    t = w.double() @ beta.double() + (bias.double() if bias is not None else 0)
    return wf, wf.double().sum(1).float(), t.float()


@pytest.mark.parametrize("tile", LIN_TILES)
def test_lin_gemm_fused_layernorm(tile):
    M, C, N = 300, 384, 160
    x = rnd(M, C, seed=1) * 2 + 0.3
    ga, be = 1 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3)
    w, b = rnd(N, C, seed=4, scale=0.05), rnd(N, seed=5, scale=0.1)
    wf, rs, t = _ln_fold(w, ga, be, b)
    tp = Tape(DEV)
    out = tp.alloc(M, N)
    tp.linear(x.to(DEV), wf.to(DEV), t.to(DEV), out, M=M, K=C, N=N, ln_rowsum=rs.to(DEV), tile=tile)
    run(tp)
    ref = F.linear(F.layer_norm(x, (C,), ga, be), w, b)
    assert (out.cpu() - ref).abs().max() < 5e-5


@pytest.mark.parametrize("tile", [13, 14, 15, 17, 1, 3])
@pytest.mark.parametrize("ln", [0, 1])
def test_fused_geglu(tile, ln):
    """FF1 with the GEGLU gate in the epilogue (packed value/gate rows) vs Linear -> chunk -> x * gelu(gate)."""
    from audioeditingcode_amd.unet import geglu_pack_index
    M, C = (300, 128) if tile not in (1, 3) else (1100, 128)
    dff = 4 * C
    x = rnd(M, C, seed=1) * 1.5 + 0.2
    w, b = rnd(2 * dff, C, seed=2, scale=0.08), rnd(2 * dff, seed=3, scale=0.1)
    ga, be = 1 + 0.1 * rnd(C, seed=4), 0.1 * rnd(C, seed=5)
    perm = geglu_pack_index(dff)
    tp = Tape(DEV)
    out = tp.alloc(M, dff)
    if ln:
        wf, rs, t = _ln_fold(w[perm], ga, be, b[perm])
        tp.linear(x.to(DEV), wf.to(DEV), t.to(DEV), out, M=M, K=C, N=2 * dff, ln_rowsum=rs.to(DEV), geglu=1, tile=tile)
        h = F.linear(F.layer_norm(x, (C,), ga, be), w, b)
    else:
        tp.linear(x.to(DEV), w[perm].contiguous().to(DEV), b[perm].contiguous().to(DEV), out, M=M, K=C, N=2 * dff,
                  geglu=1, tile=tile)
        h = F.linear(x, w, b)
    run(tp)
    a, gate = h.chunk(2, dim=-1)
    assert (out.cpu() - a * F.gelu(gate)).abs().max() < 5e-5


@pytest.mark.parametrize("tile", [10, 11, 15, 4, 1])
def test_two_source_conv(tile):
    """1x1 conv over a channel concat (h | skip) that is never materialised (up-block shortcut conv)."""
    B, H, W, C1, C2, N = 2, 24, 8, 128, 64, 160
    if tile == 1:
        B, H = 4, 64
    h, sk = rnd(B, H, W, C1 + 32, seed=1), rnd(B, H, W, C2, seed=2)
    w, b = rnd(N, C1 + C2, seed=3, scale=0.07), rnd(N, seed=4)
    tp = Tape(DEV)
    out = tp.alloc(B, H, W, N)
    hd = h.to(DEV)
    tp.conv(hd[..., :C1], w.to(DEV), b.to(DEV), out, B=B, IH=H, IW=W, Cin=C1 + C2, OH=H, OW=W, N=N, x2=sk.to(DEV), C1=C1,
            tile=tile)
    run(tp)
    ref = F.linear(torch.cat([h[..., :C1], sk], -1), w, b)
    assert (out.cpu() - ref).abs().max() < 3e-5


@pytest.mark.parametrize("C1,C2,HW,B,variant", [(384, 256, 256, 2, 0), (640, 640, 64, 2, 0), (128, 128, 4096, 2, 0),
                                                (384, 256, 256, 2, 1), (256, 128, 4096, 6, 0),
                                                (640, 320, 1024, 2, 0)])         # TANGO up block: 960 / 32 = 30 per group
def test_groupnorm_two_source(C1, C2, HW, B, variant):
    """GroupNorm over (h | skip) read in place; 640 channels in 32 groups of 20 straddle the 384|256 boundary."""
    C, G = C1 + C2, 32
    h, sk = rnd(B, HW, C1 + 64, seed=1) * 2 + 0.5, rnd(B, HW, C2, seed=2) - 0.3
    ga, be = 1 + 0.1 * rnd(C, seed=3), 0.1 * rnd(C, seed=4)
    x = torch.cat([h[..., :C1], sk], -1)
    ref = F.silu(F.group_norm(x.permute(0, 2, 1), G, ga, be, 1e-5)).permute(0, 2, 1)
    tp = Tape(DEV)
    out = tp.alloc(B, HW, C)
    hd = h.to(DEV)
    tp.groupnorm(hd[..., :C1], ga.to(DEV), be.to(DEV), out, B=B, HW=HW, C=C, G=G, eps=1e-5, act=1, x2=sk.to(DEV), C1=C1,
                 variant=variant)
    run(tp)
    assert (out.cpu() - ref).abs().max() < 2e-5


@pytest.mark.parametrize("tile", [10, 11, 13, 15])
@pytest.mark.parametrize("Lk", [8, 16])
def test_lin_gemm_per_batch_weights_and_grouped_softmax(tile, Lk):
    """The two skinny GEMMs of the folded cross-attention: (1) LayerNorm-folded scores with per-batch-item weights and
    interleaved (rowsum, bias) vectors + per-head softmax over Lk columns with a key mask; (2) P . VO^T + bias +
    residual with per-batch-item weights."""
    B, N, C, H = 3, 128, 96, 4
    HL, M = H * Lk, B * N
    x = rnd(M, C, seed=1) * 1.5 + 0.2
    ga, be = 1 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3)
    G = rnd(B, HL, C, seed=4, scale=0.2)                       # per batch item: k_h . Wq_h
    kb = torch.zeros(B, Lk)
    kb[1, Lk // 2:] = -10000.0
    kb[2, -1] = -1.0e30
    Gp = G * ga                                                # gamma folded
    gs = torch.stack([Gp.double().sum(-1).float(), (G.double() @ be.double()).float()], -1).contiguous()   # [B, HL, 2]
    tp = Tape(DEV)
    P = tp.alloc(M, HL)
    gsd = gs.to(DEV).view(-1)
    tp.conv(x.to(DEV), Gp.contiguous().to(DEV), gsd[1:], P, B=B, IH=N, IW=1, Cin=C, OH=N, OW=1, N=HL, ln_rowsum=gsd,
            w_bs=HL * C, vec_ld=2, vec_bs=HL * 2, sm_group=Lk, sm_scale=0.25, kbias=kb.to(DEV), tile=tile)
    VOt = rnd(B, C, HL, seed=5, scale=0.3)
    bo, res = rnd(C, seed=6), rnd(M, C, seed=7)
    out = tp.alloc(M, C)
    tp.conv(P, VOt.to(DEV), bo.to(DEV), out, B=B, IH=N, IW=1, Cin=HL, OH=N, OW=1, N=C, res=res.to(DEV), w_bs=C * HL,
            tile=tile)
    run(tp)
    xn = F.layer_norm(x, (C,), ga, be).view(B, N, C)
    sc = torch.einsum("bnc,bkc->bnk", xn, G) * 0.25 + kb.repeat(1, H)[:, None, :]
    Pr = torch.softmax(sc.view(B, N, H, Lk), -1).view(B, N, HL)
    assert (P.cpu().view(B, N, HL) - Pr).abs().max() < 2e-5
    ref = torch.einsum("bnk,bck->bnc", Pr, VOt) + bo + res.view(B, N, C)
    assert (out.cpu().view(B, N, C) - ref).abs().max() < 5e-5


@pytest.mark.parametrize("B,Lk,H,D", [(2, 8, 8, 80), (3, 16, 4, 48), (2, 32, 2, 16)])
def test_xattn_fold_operands(B, Lk, H, D):
    """per-prompt operands of the folded cross-attention vs einsum"""
    C, HL = H * D, H * Lk
    kv = rnd(B * Lk, 2 * C + 8, seed=1)
    xq, xs, xo = rnd(H, C, D, seed=2, scale=0.2), rnd(H, 2, D, seed=3, scale=0.2), rnd(H, C, D, seed=4, scale=0.2)
    tp = Tape(DEV)
    G, gs, VOt = tp.alloc(B, HL, C), tp.alloc(B, HL, 2), tp.alloc(B, C, HL)
    kvd = kv.to(DEV)
    tp.xattn_fold(kvd[:, :2 * C], xq.to(DEV), xs.to(DEV), xo.to(DEV), G, gs, VOt, B=B, Lk=Lk, H=H, C=C, D=D)
    run(tp)
    k = kv[:, :C].reshape(B, Lk, H, D)
    v = kv[:, C:2 * C].reshape(B, Lk, H, D)
    assert (G.cpu().view(B, H, Lk, C) - torch.einsum("bjhd,hcd->bhjc", k, xq)).abs().max() < 2e-5
    assert (gs.cpu().view(B, H, Lk, 2) - torch.einsum("bjhd,hsd->bhjs", k, xs)).abs().max() < 2e-5
    assert (VOt.cpu().view(B, C, H, Lk) - torch.einsum("bjhd,hcd->bchj", v, xo)).abs().max() < 2e-5

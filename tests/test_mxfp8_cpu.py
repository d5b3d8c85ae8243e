"""The CPU emulation of the MX-FP8 experiment's quantisation (oracle/mxfp8.py): format properties only -- there is no reference
code for this arithmetic (the reference computes in fp32), so nothing here is a parity claim."""
import torch

from oracle import mxfp8


def test_mx_dequantised_values_are_e4m3_times_a_power_of_two_and_close():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(7, 256, generator=g) * torch.exp(2 * torch.randn(7, 8, generator=g)).repeat_interleave(32, 1)
    d = mxfp8.mx_dequantized(x)
    assert d.shape == x.shape and torch.isfinite(d).all()
    assert torch.equal(mxfp8.mx_dequantized(d), d)                       # idempotent: already representable
    xb, db = x.reshape(7, 8, 32), d.reshape(7, 8, 32)
    amax = xb.abs().amax(-1, keepdim=True)
    # a block's maximum lands in [256, 512) of the element range: rounding costs at most half an ulp of the top binade (2^-4 of
    # amax), and the spec's saturation of (448, 512) to 448 at most 2^-3 of amax
    assert ((db - xb).abs() <= amax * 2.0 ** -3 + 1e-30).all()
    scale = 2.0 ** (torch.floor(torch.log2(amax)) - 8)
    assert (db.abs() <= 448 * scale).all()


def test_mx_zero_and_tiny_blocks_and_k_check():
    x = torch.zeros(2, 64)
    x[1, 40] = 1e-41                                                     # denormal: the smallest scale, flushed or kept, finite
    d = mxfp8.mx_dequantized(x)
    assert torch.isfinite(d).all() and torch.equal(d[0], x[0])
    try:
        mxfp8.mx_dequantized(torch.zeros(3, 48))
    except ValueError:
        pass
    else:
        raise AssertionError("K = 48 must be rejected")


def test_mx_linear_deviation_is_percent_level_not_parity():
    g = torch.Generator().manual_seed(1)
    a, w = torch.randn(64, 512, generator=g), torch.randn(96, 512, generator=g) / 512 ** 0.5
    ref = a.double() @ w.double().T
    got = mxfp8.mx_linear(a, w)
    rel = float((got - ref).norm() / ref.norm())
    assert 5e-3 < rel < 8e-2, rel                                        # a few percent: this arithmetic is an experiment


def test_tape_interpreter_states_the_fp8_flag_like_the_kernel_launcher():
    """oracle/tape_interp.py on AED_OP_CONV_GEMM records with flag bit 6: shapes the device kernel takes (Cin % 64 == 0, LDS-staged
    tile) contract MX-FP8 operands -- LayerNorm statistics from the RAW rows, activation before the quantisation, K in (tap, channel)
    order --, shapes it does not take stay fp32.  (CPU semantics of the experiment; the kernel itself is tested on the GPU.)"""
    import torch.nn.functional as F

    from audioeditingcode_amd import tape as tape_mod
    from audioeditingcode_amd.tape import Tape
    from oracle import tape_interp
    g = torch.Generator().manual_seed(0)
    M, K, N = 96, 128, 64
    x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g)
    tp = Tape("cpu")
    out, out32, conv_out = tp.alloc(M, N), tp.alloc(M, N), tp.alloc(2, 6, 5, 64)
    xi = torch.randn(2, 6, 5, 64, generator=g)
    wc = torch.randn(64, 3, 3, 64, generator=g) / 24.0
    with tape_mod.arith_mode("fp8"):
        tp.linear(tp.hold(x), tp.hold(w), tp.hold(b), out, M=M, K=K, N=N, in_act=1, tile=1)          # SiLU on A, then quantise
        tp.linear(tp.hold(x[:, :32].contiguous()), tp.hold(w[:, :32].contiguous()), None, out32, M=M, K=32, N=N, tile=4)
        tp.conv(tp.hold(xi), tp.hold(wc.reshape(64, -1).contiguous()), None, conv_out, B=2, IH=6, IW=5, Cin=64, OH=6, OW=5, N=64,
                KH=3, KW=3, pad_h=1, pad_w=1, tile=4)
    assert all(op.flags & 64 for op in tp.ops)
    tape_interp.run_tape(tp)
    assert torch.allclose(out.double(), mxfp8.mx_linear(F.silu(x), w) + b.double(), rtol=1e-6, atol=1e-6)
    assert torch.allclose(out32.double(), x[:, :32].double() @ w[:, :32].double().T, rtol=1e-6, atol=1e-6)      # Cin = 32: fp32
    assert torch.allclose(conv_out.double(), mxfp8.mx_conv2d_nhwc(xi, wc), rtol=1e-6, atol=1e-6)
    exact = F.silu(x).double() @ w.double().T + b.double()
    dev = float((out.double() - exact).norm() / exact.norm())
    assert 5e-3 < dev < 8e-2, dev

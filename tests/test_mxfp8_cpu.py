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

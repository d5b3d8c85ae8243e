"""The arithmetic claim behind csrc/conv_gemm_x6.hip, checked on the CPU (tools/bf16_split_study.py's emulation): an fp32
number is EXACTLY three bf16 pieces; the six piece products of relative size >= 2^-16, accumulated in fp32, are as close to
the fp64 product as a plain fp32 GEMM; three terms are not."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import bf16_split_study as study                   # noqa: E402


@pytest.mark.parametrize("mode", ["rne", "trunc"])
def test_three_bf16_pieces_are_the_fp32_number(mode):
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(1 << 16) * np.exp(rng.standard_normal(1 << 16) * 6)).astype(np.float32)      # 1e-8 .. 1e8
    h, m, lo, resid = study.split3(x, mode)
    assert resid == 0.0
    assert np.array_equal((h.astype(np.float64) + m.astype(np.float64) + lo.astype(np.float64)).astype(np.float32), x)
    for piece in (h, m, lo):                        # every piece is a bf16 value: low 16 bits of its fp32 encoding are zero
        assert not np.any(piece.view(np.uint32) & np.uint32(0xFFFF))


@pytest.mark.parametrize("K", [256, 2304])
def test_six_terms_match_the_fp32_chain_three_do_not(K):
    rng = np.random.default_rng(K)
    A = (rng.standard_normal((256, K)) * np.exp(rng.standard_normal((1, K)) * 1.5)).astype(np.float32)
    W = (rng.standard_normal((128, K)) / np.sqrt(K)).astype(np.float32)
    ref = A.astype(np.float64) @ W.astype(np.float64).T
    err = lambda c: float(np.linalg.norm(c.astype(np.float64) - ref) / np.linalg.norm(ref))         # noqa: E731
    e32 = err(A @ W.T)
    e6 = err(study.emulate(A, W, "x6", "rne")[0])
    e9 = err(study.emulate(A, W, "x9", "rne")[0])
    e3 = err(study.emulate(A, W, "x3", "rne")[0])
    assert e6 < 1.5 * e32 and e6 < 1e-6 and abs(e6 - e9) < 0.2 * e9      # the three dropped terms are invisible
    assert 1e-6 < e3 < 2e-5 and e3 > 5 * e6                               # three terms: a 16-bit significand, not fp32

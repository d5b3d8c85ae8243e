"""Deterministic synthetic stand-ins shared by make_golden.py and the tests.

TEST INFRASTRUCTURE.  `synthetic_unet` is a tiny, smooth, seed-free eps-model
used to drive BOTH the reference loops (when generating golden vectors) and the
oracle / product loops (when checking them), so loop parity is independent of
any real network.
"""
import math

import torch
import torch.nn.functional as F


def _kernel(c):
    # fixed, asymmetric 3x3 mixing kernel (no RNG)
    k = torch.zeros(c, c, 3, 3)
    for o in range(c):
        for i in range(c):
            for y in range(3):
                for x in range(3):
                    k[o, i, y, x] = math.sin(1.0 + 0.7 * o + 1.3 * i + 2.1 * y + 0.37 * x) / (3.0 * c)
    return k


_KCACHE = {}


def synthetic_unet(x, t, cond_vec):
    """x[B,C,H,W], t scalar (int tensor), cond_vec[B,D] -> eps[B,C,H,W] (fp32)."""
    c = x.shape[1]
    if c not in _KCACHE:
        _KCACHE[c] = _kernel(c)
    k = _KCACHE[c].to(x.dtype)
    tt = float(int(t)) / 1000.0
    h = F.conv2d(x, k, padding=1)
    shift = cond_vec.to(x.dtype).mean(dim=1).view(-1, 1, 1, 1)
    return torch.tanh(h * (0.5 + tt) + 0.25 * shift) + 0.1 * x * (1.0 - tt)


def prompt_vec(prompt, dim=4):
    """Deterministic 'text embedding' for a prompt string."""
    v = torch.zeros(dim)
    for i, ch in enumerate(prompt):
        v[i % dim] += ((ord(ch) % 17) - 8) / 8.0
    return v


def chirp_waveform(n=163840, sr=16000, seed=1234):
    """SURVEY 8(d) synthetic clip: 100->4000 Hz chirp + 0.05 randn (CPU generator)."""
    g = torch.Generator().manual_seed(seed)
    tt = torch.arange(n, dtype=torch.float64) / sr
    dur = n / sr
    f0, f1 = 100.0, 4000.0
    phase = 2 * math.pi * (f0 * tt + 0.5 * (f1 - f0) / dur * tt * tt)
    x = 0.5 * torch.sin(phase).float() + 0.05 * torch.randn(n, generator=g)
    return x

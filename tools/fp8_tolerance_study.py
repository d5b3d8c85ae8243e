"""What an fp8 (e4m3) GEMM path would cost the Stable Audio edit numerically -- a CPU study, no kernel involved.

BASELINE config 5 names an "fp8 MFMA path"; the reference computes in fp32 and so does this build.  Before writing such a
kernel the tolerance has to be known, and it can be measured without one: an fp8 MFMA with fp32 accumulation is emulated
exactly by rounding both GEMM operands to e4m3 (per-row dynamic scale for activations, per-output-channel scale for
weights) and multiplying in fp32 -- products of two e4m3 numbers are exact in fp32.  The oracle DiT (oracle/stable_audio.py)
is run with every Linear of the transformer blocks routed through that emulation (attention products, LayerNorm, rotary,
the solver stay fp32) and compared with the fp32 run:

  1. error of one forward (full width, a few layers) at high / mid / low sigma;
  2. how the solver turns a model-output error into a noise-map error: |dz/dv| = |k2 * c_out / k3| per step of the T=200 table;
  3. inversion + edit on the tiny DiT, fp8-emulated model in BOTH passes vs fp32 in both passes: deviation of the edited latent,
     and the reconstruction property (same prompt both ways) under fp8.

    python tools/fp8_tolerance_study.py [--layers 4] [--T 50]"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioeditingcode_amd import configs, weights                                    # noqa: E402
from audioeditingcode_amd.scheduler import CosineDPMSolverMultistepScheduler, sa_coefficient_table   # noqa: E402
from oracle import stable_audio as osa                                               # noqa: E402

E4M3_MAX = 448.0


def q_rows(x):
    """Round to e4m3 with one scale per row (last dim = the contraction dim)."""
    s = x.abs().amax(dim=-1, keepdim=True).clamp_min(1e-12) / E4M3_MAX
    return (x / s).to(torch.float8_e4m3fn).float() * s


_WQ = {}
_LINEAR = F.linear                                # the real one (fp8_blocks patches the module attribute)


def fp8_linear(x, w, b=None):
    key = (w.data_ptr(), tuple(w.shape))
    if key not in _WQ:
        _WQ[key] = q_rows(w)                      # per-output-channel weight scale
    return _LINEAR(q_rows(x), _WQ[key], b)


class fp8_blocks:
    """Context manager: F.linear inside the transformer blocks is replaced by the fp8 emulation (the time / global / context
    MLPs and the in / out projections are tiny and stay fp32, as a real path would keep them)."""

    def __enter__(self):
        self.orig = F.linear

        def patched(x, w, b=None):
            big = w.shape[0] >= 64 and w.shape[1] >= 64 and x.dim() == 3 and x.shape[1] > 2
            return fp8_linear(x, w, b) if big else self.orig(x, w, b)
        F.linear = patched
        osa.F.linear = patched
        return self

    def __exit__(self, *a):
        F.linear = self.orig
        osa.F.linear = self.orig


def rel(a, b):
    return float((a - b).norm() / b.norm())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--T", type=int, default=50)
    args = ap.parse_args()
    torch.manual_seed(0)

    # ---- 1. one forward, full width
    cfg = dict(configs.FAMILIES["stable_audio"]["dit"], num_layers=args.layers)
    sd = weights.random_state_dict(weights.dit_param_shapes(cfg), seed=0)
    g = torch.Generator().manual_seed(1)
    ctx = torch.randn(1, 130, cfg["cross_attention_input_dim"], generator=g)
    glob = torch.randn(1, 1, cfg["global_states_input_dim"], generator=g)
    rot = osa.rotary_table(cfg["attention_head_dim"] // 2, cfg["sample_size"] + 1)
    s = CosineDPMSolverMultistepScheduler()
    s.set_timesteps(200)
    x0 = torch.randn(1, cfg["in_channels"], cfg["sample_size"], generator=g)
    print(f"1. one DiT forward, width {cfg['num_attention_heads'] * cfg['attention_head_dim']}, {args.layers} layers, e4m3 GEMM "
          f"operands (fp32 accumulate) vs fp32:")
    for i in (10, 100, 190):
        sig = s.sigmas[i]
        x = (x0 + sig * torch.randn(x0.shape, generator=g)) / (sig ** 2 + 1) ** 0.5
        with torch.no_grad():
            ref = osa.dit_forward(sd, cfg, x, s.timesteps[i].reshape(1), ctx, glob, rot)
            with fp8_blocks():
                got = osa.dit_forward(sd, cfg, x, s.timesteps[i].reshape(1), ctx, glob, rot)
        print(f"   step {i:3d} sigma {float(sig):8.3f}: rel L2 error of the model output {rel(got, ref):.3e}")

    # ---- 2. error transfer of the solver
    tab = sa_coefficient_table(s, 0, 200, 0, invert=True)
    gain = (tab[:, 4] * tab[:, 2].abs() / tab[:, 5].clamp_min(1e-30))[:199]
    print(f"2. |dz/dv| = k2*|c_out|/k3 over the T=200 schedule: min {float(gain.min()):.3f}  median "
          f"{float(gain.median()):.3f}  max {float(gain.max()):.3f} (at step {int(gain.argmax())})")

    # ---- 3. loops on the tiny DiT
    fam = configs.get_family("tiny/stable-audio-open-1.0")
    tcfg = fam["dit"]
    tsd = weights.random_state_dict(weights.dit_param_shapes(tcfg), seed=4)
    T, tstart = args.T, args.T // 2
    g = torch.Generator().manual_seed(11)
    z0 = torch.randn(1, tcfg["in_channels"], tcfg["sample_size"], generator=g)
    c_src, c_tgt = (torch.randn(1, 6, tcfg["cross_attention_input_dim"], generator=g) for _ in range(2))
    c_unc = torch.zeros_like(c_src)
    tglob = torch.randn(1, 1, tcfg["global_states_input_dim"], generator=g)
    trot = osa.rotary_table(tcfg["attention_head_dim"] // 2, tcfg["sample_size"] + 1)
    noise_seed = 5

    def run(fp8, tgt):
        osched = osa.OracleCosineDPMSolverScheduler()
        osched.set_timesteps(T)

        def dit(x_inp, t, c):
            return osa.dit_forward(tsd, tcfg, x_inp, t.reshape(1), c, tglob, trot)
        ow = osa.OracleStableAudio(osched, dit, in_channels=tcfg["in_channels"], sample_size=tcfg["sample_size"])
        cm = fp8_blocks() if fp8 else _null()
        with torch.no_grad(), cm:
            _, zs, xts, extra = osa.invert(ow, z0, c_src, c_unc, 1.0, T, generator=torch.Generator().manual_seed(noise_seed))
            out = osa.edit(ow, xts, tstart, tgt, c_unc, 1.0 if tgt is c_src else 6.0, zs[:tstart], extra_info=extra)
        return out, xts

    e32, _ = run(False, c_tgt)
    e8, _ = run(True, c_tgt)
    r8, xts8 = run(True, c_src)
    print(f"3. tiny DiT, T={T}, tstart={tstart}: edited latent, fp8-emulated model in both passes vs fp32 in both: rel L2 "
          f"{rel(e8, e32):.3e}")
    print(f"   reconstruction with the fp8-emulated model (same prompt and guidance both ways): rel L2 "
          f"{rel(r8, xts8[0:1]):.3e} (the inversion absorbs the model's error, whatever the model is)")


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        pass


if __name__ == "__main__":
    main()

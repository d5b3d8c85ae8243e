"""TEST INFRASTRUCTURE: one run of the CPU oracle's DDPM inversion + edit at the BENCHED size and length (AudioLDM2 U-Net
346.9 M, latent 8x256x16, T=200, tstart=100, cfg 3 / 12, eta 1) on seeded inputs, stored as a fixture so that the GPU suite can
compare the HIP loops with the oracle at full length without paying ~6 minutes of CPU per run
(tests/test_gpu_zzz_fullsize_oracle_fixture.py).

Everything the test needs besides the outputs is regenerated from seeds (weights: weights.random_state_dict seed 0, the
seeded-random U-Net of every full-size test; conditioning and x0: generator seed 11; the x_t draws: oracle
sample_xts_from_x0 with generator seed 1), so the file holds oracle OUTPUTS only: the edited latent, x_T, four noise maps and
the per-step norms of all of them.  This is the oracle's statement of inversion_utils.py:52-144 / :199-321 (oracle/loops.py),
not a reference fixture: it pins the HIP path to the oracle at the benched length, the oracle itself stays pinned to the
reference by tests/golden/loop_*.npz.

    PYTHONPATH=. python oracle/make_fullsize_loop_golden.py        # ~6 min on 8 cores -> tests/golden/fullsize_loop_T200.npz
    PYTHONPATH=. python oracle/make_fullsize_loop_golden.py eight  # 8-clips-per-engine shape, 3 clips -> fullsize_eight_clips_T4.npz"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audioeditingcode_amd import configs, weights          # noqa: E402
from oracle import loops as oloops, unet as ounet          # noqa: E402
from oracle.scheduler import OracleDDIMScheduler           # noqa: E402

T, TSTART = 200, 100


def inputs():
    """The seeded inputs, in the order the GPU test regenerates them."""
    g = torch.Generator().manual_seed(11)
    mk = lambda L1: dict(encoder_hidden_states=torch.randn(1, 8, 768, generator=g),          # noqa: E731
                         encoder_hidden_states_1=torch.randn(1, L1, 1024, generator=g),
                         encoder_attention_mask_1=torch.ones(1, L1))
    src, tgt, unc = mk(7), mk(9), mk(1)
    x0 = torch.randn(1, 8, 256, 16, generator=g) * 0.8
    return src, tgt, unc, x0


def main():
    cfg = configs.FAMILIES["audioldm2"]["unet"]
    sd = weights.random_state_dict(weights.unet_param_shapes(cfg), seed=0)
    src, tgt, unc, x0 = inputs()
    osched = OracleDDIMScheduler()
    osched.set_timesteps(T)
    ow = oloops.OracleWrapper(osched, lambda x, t, c: ounet.unet_forward(
        cfg, sd, x, t, **{k: v.expand(x.shape[0], *v.shape[1:]) for k, v in c.items()})[0])
    xts0 = ow.sample_xts_from_x0(x0, T, generator=torch.Generator().manual_seed(1))
    t0 = time.time()
    with torch.inference_mode():
        _, zs, xts = oloops.invert(ow, x0, src, unc, [3.0], T, eta=1.0, xts=xts0.clone())
        print(f"inversion {time.time() - t0:.0f} s", flush=True)
        w = oloops.edit(ow, xts, torch.tensor([TSTART]), tgt, unc, [12.0], zs[:TSTART], eta=1.0)
    print(f"inversion + edit {time.time() - t0:.0f} s", flush=True)
    keep = [1, 50, 100, 199]
    out = os.path.join(ROOT, "tests", "golden", "fullsize_loop_T200.npz")
    np.savez_compressed(out, w_edit=w.numpy(), xT=xts[-1].numpy() if xts.shape[0] == T + 1 else xts[0].numpy(),
                        zs_keep=zs[keep].numpy(), keep=np.array(keep), zs_norms=zs.flatten(1).norm(dim=1).numpy(),
                        xts_norms=xts.flatten(1).norm(dim=1).numpy(), T=np.array(T), tstart=np.array(TSTART),
                        zs_shape=np.array(zs.shape), xts_shape=np.array(xts.shape))
    print("wrote", out, os.path.getsize(out), "bytes; zs", tuple(zs.shape), "xts", tuple(xts.shape))


# ---- BASELINE config 3's per-rank shape: 8 clips per engine at full size, short schedule ---------------------------------
T8, TSTART8, N8, CLIPS8 = 4, 2, 8, (0, 5, 7)


def inputs_eight():
    """Seeded inputs of tests/test_gpu_loops.py::test_eight_clips_per_engine_full_size_audioldm2, same draw order."""
    g = torch.Generator().manual_seed(11)
    mk = lambda L1: dict(encoder_hidden_states=torch.randn(1, 8, 768, generator=g),          # noqa: E731
                         encoder_hidden_states_1=torch.randn(1, L1, 1024, generator=g),
                         encoder_attention_mask_1=torch.ones(1, L1))
    src, tgt, unc = mk(7), mk(9), mk(1)
    x0s = torch.randn(N8, 8, 256, 16, generator=g) * 0.8
    noise = torch.randn(T8, N8, 8, 256, 16, generator=g)
    return src, tgt, unc, x0s, noise


def main_eight():
    """The oracle's edit of clips 0, 5 and 7 of that batch, one at a time, from the SAME per-clip noise maps."""
    cfg = configs.FAMILIES["audioldm2"]["unet"]
    sd = weights.random_state_dict(weights.unet_param_shapes(cfg), seed=0)
    src, tgt, unc, x0s, noise = inputs_eight()
    osched = OracleDDIMScheduler()
    osched.set_timesteps(T8)
    ow = oloops.OracleWrapper(osched, lambda x, t, c: ounet.unet_forward(
        cfg, sd, x, t, **{k: v.expand(x.shape[0], *v.shape[1:]) for k, v in c.items()})[0])
    ts, abar = osched.timesteps, osched.alphas_cumprod
    outs = []
    with torch.inference_mode():
        for i in CLIPS8:
            x0 = x0s[i:i + 1]
            xts0 = torch.zeros(T8 + 1, 8, 256, 16)
            xts0[0] = x0[0]
            for r in range(T8):                  # draw r belongs to the r-th smallest timestep (models.py:76-81 order)
                t = int(ts[T8 - 1 - r])
                xts0[r + 1] = x0[0] * abar[t] ** 0.5 + noise[r, i] * (1 - abar[t]) ** 0.5
            _, zs, xts = oloops.invert(ow, x0, src, unc, [3.0], T8, eta=1.0, xts=xts0)
            outs.append(oloops.edit(ow, xts, torch.tensor([TSTART8]), tgt, unc, [12.0], zs[:TSTART8], eta=1.0))
    out = os.path.join(ROOT, "tests", "golden", "fullsize_eight_clips_T4.npz")
    np.savez_compressed(out, w_edit=torch.cat(outs).numpy(), clips=np.array(CLIPS8), T=np.array(T8), tstart=np.array(TSTART8))
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "eight":
        main_eight()
    else:
        main()

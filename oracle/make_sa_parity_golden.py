"""TEST INFRASTRUCTURE: the CPU oracle's Stable Audio Open inversion + edit at FULL size and length -- DiT with all 24 layers
(1.06 B seeded-random parameters), latent 64 x 1024, T=200, tstart=100, cfg 1 / 7, the prompts / duration of
tools/bench_stable_audio.py, reference step order -- from a seeded latent, stored as a fixture
(tests/golden/sa_parity_T200.npz: the latent, the edited latent, x_T, three noise maps).  tools/bench_stable_audio.py runs the
HIP loops on the same latent with the same seed and REPORTS the rel L2 (`parity_T200`); nothing asserts on it: the solver
history makes this loop more sensitive than the DDIM-table one (the tiny-model test allows 5e-3 against the oracle).

The conditioning is the wrapper's synthetic stand-in, computed by the wrapper class itself on the CPU (a subclass that skips
the HIP-device check: test instrumentation, as in tests/test_stable_audio_cpu.py); the oracle DiT / solver are
oracle/stable_audio.py (restated from the published diffusers definitions: PARITY UNPINNED).

    PYTHONPATH=. python oracle/make_sa_parity_golden.py        # ~45 min on 8 cores"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audioeditingcode_amd import models                    # noqa: E402
from oracle import stable_audio as osa                     # noqa: E402

T, TSTART, SEED = 200, 100, 77
SRC, TGT, NEG = "a recording of a piano melody", "a recording of an electric guitar melody", ""
CFG_SRC, CFG_TAR = 1.0, 7.0


class _CpuSA(models.StableAudWrapper):
    def _require_device(self):          # instrumentation: the product class refuses a CPU device
        pass


def seeded_latent(cfg):
    return torch.randn(1, cfg["in_channels"], cfg["sample_size"], generator=torch.Generator().manual_seed(123)) * 0.8


def main():
    t0 = time.time()
    m = _CpuSA(model_id="stabilityai/stable-audio-open-1.0", device="cpu", allow_synthetic=True)
    m.load_scheduler()
    m.model.scheduler.set_timesteps(T, device=None)
    print(f"wrapper on the CPU after {time.time() - t0:.0f} s", flush=True)
    cfg, sd = m.family["dit"], m.state_dicts["transformer"]
    duration = cfg["sample_size"] * m.model.vae.hop_length / m.get_sr()
    w0 = seeded_latent(cfg)
    m.setup_extra_inputs(w0, init_timestep=m.model.scheduler.timesteps[0], audio_end_in_s=duration)
    glob = m.audio_duration_embeds.cpu()
    ctxs = [m.assemble_context(*[m.encode_text([p], negative=neg)[k] for k in (0, 2)]).cpu()
            for p, neg in ((SRC, False), (TGT, False), (NEG, True))]
    osched = osa.OracleCosineDPMSolverScheduler()
    osched.set_timesteps(T)
    rot = osa.rotary_table(cfg["attention_head_dim"] // 2, cfg["sample_size"] + 1)
    ow = osa.OracleStableAudio(osched, lambda x, t, c: osa.dit_forward(sd, cfg, x, t.reshape(1), c, glob, rot),
                               in_channels=cfg["in_channels"], sample_size=cfg["sample_size"])
    torch.manual_seed(SEED)
    with torch.inference_mode():
        xts0 = ow.sample_xts_from_x0(w0, T)
        _, zs, xts, extra = osa.invert(ow, w0, ctxs[0], ctxs[2], CFG_SRC, T, xts=xts0)
        print(f"inversion done at {time.time() - t0:.0f} s", flush=True)
        w_o = osa.edit(ow, xts, TSTART, ctxs[1], ctxs[2], CFG_TAR, zs[:TSTART], extra_info=extra)
    print(f"edit done at {time.time() - t0:.0f} s", flush=True)
    keep = [1, 100, 199]
    out = os.path.join(ROOT, "tests", "golden", "sa_parity_T200.npz")
    np.savez_compressed(out, w0=w0.numpy(), w_edit=w_o.numpy(), xT=xts[-1].numpy(), zs_keep=zs[keep].numpy(),
                        keep=np.array(keep), zs_norms=zs.flatten(1).norm(dim=1).numpy(), T=np.array(T),
                        tstart=np.array(TSTART), seed=np.array(SEED), duration=np.array(duration),
                        cfg=np.array([CFG_SRC, CFG_TAR]), prompts=np.array([SRC, TGT, NEG]))
    print("wrote", out, os.path.getsize(out), "bytes; zs", tuple(zs.shape), "xts", tuple(xts.shape), "w", tuple(w_o.shape))


if __name__ == "__main__":
    main()

"""TEST INFRASTRUCTURE: one forward of the CPU oracle's Stable Audio DiT at FULL depth (Stable Audio Open 1.0 configuration:
24 layers, width 1536, 24 / 12 heads, 1025-token sequence, 130-token context; 1.06 B seeded-random parameters) stored as a
fixture, so that the GPU suite can compare the compiled DiT tape with the oracle at full depth without ~1 minute of CPU and
4 GB of oracle weights per run (tests/test_gpu_zzz_fullsize_oracle_fixture.py).  Inputs and weights are regenerated from
seeds by the test (weights: weights.random_state_dict(dit_param_shapes, seed=3); inputs: generator seed 5, the same case as
tests/test_gpu_stable_audio.py::_dit_case); the file holds the oracle's OUTPUT and the timestep used.

This is the oracle's statement of the published diffusers StableAudioDiTModel (oracle/stable_audio.py, PARITY UNPINNED by
absence of diffusers -- DESIGN.md section 7): the fixture pins engine == oracle at full depth, nothing more.

    PYTHONPATH=. python oracle/make_fullsize_dit_golden.py       -> tests/golden/dit_full_depth.npz"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audioeditingcode_amd import configs, weights          # noqa: E402
from oracle import stable_audio as osa                     # noqa: E402

S, STEP = 130, 90


def case(cfg, seed=5):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(2, cfg["in_channels"], cfg["sample_size"], generator=g)
    x[1] = x[0]
    ctx = torch.randn(2, S, cfg["cross_attention_input_dim"], generator=g)
    ctx[0] = 0
    glob = torch.randn(1, cfg["global_states_input_dim"], generator=g).expand(2, -1).contiguous()
    return x, ctx, glob


def main():
    cfg = dict(configs.FAMILIES["stable_audio"]["dit"])
    t0 = time.time()
    sd = weights.random_state_dict(weights.dit_param_shapes(cfg), seed=3)
    print(f"weights {time.time() - t0:.0f} s ({sum(v.numel() for v in sd.values()) / 1e9:.2f} B parameters)", flush=True)
    x, ctx, glob = case(cfg)
    s = osa.OracleCosineDPMSolverScheduler()
    s.set_timesteps(200)
    t = s.timesteps[STEP]
    with torch.inference_mode():
        v = osa.dit_forward(sd, cfg, x, t.reshape(1), ctx, glob[:, None, :],
                            osa.rotary_table(cfg["attention_head_dim"] // 2, cfg["sample_size"] + 1))
    print(f"forward done at {time.time() - t0:.0f} s; |v| = {float(v.norm()):.4f}", flush=True)
    out = os.path.join(ROOT, "tests", "golden", "dit_full_depth.npz")
    np.savez_compressed(out, v=v.numpy(), t=np.array(float(t)), num_layers=np.array(cfg["num_layers"]))
    print("wrote", out, os.path.getsize(out), "bytes", tuple(v.shape))


if __name__ == "__main__":
    main()

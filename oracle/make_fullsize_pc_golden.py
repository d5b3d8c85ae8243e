"""TEST INFRASTRUCTURE: BASELINE config 4's inner loop at FULL SIZE through the CPU oracle (oracle/pc.py, pinned to
/root/reference/code/pc_drift.py:96-198 by tests/golden/pc_drift.npz): AudioLDM2 U-Net (346.9 M), latent 8x256x16, T=200, one
drift timestep, n_evs=4 directions, ITERS power iterations from CPU-drawn start vectors, then apply_drift along PCs 1 and 2.
Stored as oracle OUTPUTS only (eigenvalues, eigenvectors, the guided step, the drifted sample); every input is regenerated
from seeds by the consumer (tests/test_gpu_pc.py::test_full_size_power_iteration_vs_the_oracle_fixture).

    PYTHONPATH=. python oracle/make_fullsize_pc_golden.py        # ~1-2 min of CPU -> tests/golden/fullsize_pc.npz"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audioeditingcode_amd import configs, weights          # noqa: E402
from oracle import loops as oloops, pc as opc, unet as ounet   # noqa: E402
from oracle.scheduler import OracleDDIMScheduler           # noqa: E402

# CONST: the finite-difference step.  The reference's default is 1e-3 (main_pc_extract_inv.py -c); at full size with x0_hat of norm
# ~670 a probe of norm c moves x0_hat by ~2.5 c, so the ~4e-6 relative deviation between two correct fp32 U-Net forwards (HIP vs
# CPU, or two GPUs) is amplified to 670 * 4e-6 / (2.5 c) = 1e-3 / c PER ITERATION of the un-contracting power iteration (seeded-
# random weights: flat spectrum).  c = 0.3 keeps the comparison well-conditioned (3e-3 per iteration); the computation is the same.
T, STEP, N_EV, ITERS, CONST, CFG, AMOUNT = 200, 120, 4, 5, 0.3, 3.0, 1.5


def inputs():
    """Seeded inputs, in the order the GPU test regenerates them: conditioning (uncond, text), x_t, the step's noise map, the
    N_EV start vectors (the reference draws them with randn_like(xt), pc_drift.py:130; both sides take these CPU draws)."""
    g = torch.Generator().manual_seed(21)
    mk = lambda L1: dict(encoder_hidden_states=torch.randn(1, 8, 768, generator=g),          # noqa: E731
                         encoder_hidden_states_1=torch.randn(1, L1, 1024, generator=g),
                         encoder_attention_mask_1=torch.ones(1, L1))
    unc, txt = mk(1), mk(9)
    xt = torch.randn(1, 8, 256, 16, generator=g) * 0.9
    latent = torch.randn(1, 8, 256, 16, generator=g)
    init = torch.randn(N_EV, 8, 256, 16, generator=g)
    return unc, txt, xt, latent, init


def main():
    cfg = configs.FAMILIES["audioldm2"]["unet"]
    sd = weights.random_state_dict(weights.unet_param_shapes(cfg), seed=0)
    unc, txt, xt, latent, init = inputs()
    osched = OracleDDIMScheduler()
    osched.set_timesteps(T)
    t = osched.timesteps[STEP]
    ow = oloops.OracleWrapper(osched, lambda x, tt, c: ounet.unet_forward(
        cfg, sd, x, tt, **{k: v.expand(x.shape[0], *v.shape[1:]) for k, v in c.items()})[0])
    mask = torch.ones_like(xt)
    t0 = time.time()
    with torch.inference_mode():
        xtm1, x0p = opc.forward_directional(ow, xt, t, latent, unc, txt, CFG, eta=1.0)
        ev1, val1, _, _ = opc.get_eigenvectors(ow, xt, txt, unc, latent, mask, t, x0p * mask, init, const=CONST, cfg_tar=CFG,
                                               iters=1, eta=1.0, n_ev=N_EV)
        ev, val, in_corr, in_norm = opc.get_eigenvectors(ow, xt, txt, unc, latent, mask, t, x0p * mask, init, const=CONST,
                                                         cfg_tar=CFG, iters=ITERS, eta=1.0, n_ev=N_EV)
        val, val1 = torch.as_tensor(val).reshape(-1), torch.as_tensor(val1).reshape(-1)
        drift = opc.apply_drift(ow, xtm1, x0p, t, ev, val, latent, amount=AMOUNT, eta=1.0, ev_nums=(1, 2))
    print(f"oracle: guided step + {ITERS} power iterations at full size in {time.time() - t0:.0f} s; eigenvalues {val.tolist()}")
    out = os.path.join(ROOT, "tests", "golden", "fullsize_pc.npz")
    np.savez_compressed(out, t=np.array(int(t)), step=np.array(STEP), T=np.array(T), n_ev=np.array(N_EV), iters=np.array(ITERS),
                        const=np.array(CONST), eigvec_iter1=ev1.numpy(), eigval_iter1=val1.numpy(), xtm1=xtm1.numpy(), x0_pred=x0p.numpy(), eigvec=ev.numpy(), eigval=val.numpy(),
                        in_corr=torch.stack([c.reshape(-1) for c in in_corr]).numpy(),
                        in_norm=torch.stack([torch.as_tensor(n).reshape(-1) for n in in_norm]).numpy(), drift=drift.numpy())
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()

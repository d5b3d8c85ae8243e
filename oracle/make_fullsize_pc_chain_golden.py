"""TEST INFRASTRUCTURE: BASELINE config 4's OUTER loops at full size through the CPU oracle -- three CONSECUTIVE drift timesteps of the
stated window (the last three of drift 120 -> 80 at T = 200: `--drift_start 83 --drift_end 80`, iterations 117, 118, 119 of the
trajectory, t = 411 / 406 / 401) with the chained state between them:

  * /root/reference/code/main_pc_extract_inv.py:199-256: the guided trajectory from x_T (117 lead-in steps, then the window), per
    window step `get_eigenvectors` (n_evs = 4, ITERS power iterations) and the sign-continuity rule against the previous step's PCs
    (`corr_to_swap`);
  * /root/reference/code/main_pc_apply_drift.py:141-191: the same trajectory again with `apply_drift` along PCs 1 + 2 at every
    window step, the drifted x_{t-1} feeding the next step.

AudioLDM2 U-Net (346.9 M seeded-random parameters), latent 8x256x16.  oracle/pc.py is pinned to the reference's pc_drift.py by
tests/golden/pc_drift.npz and the two CLI loops by tests/golden/pc_cli.npz (the reference's own scripts run on a stand-in model);
this fixture is the full-size, multi-timestep oracle OUTPUT for tests/test_gpu_pc.py::test_config4_three_consecutive_drift_
timesteps_at_full_size.  Every input is regenerated from seeds by the consumer.

The start vectors of window steps 2 and 3 are MINUS the previous step's PCs (the reference draws randn_like there, pc_drift.py:130
-- both sides take THESE tensors) and corr_to_swap is 0.3.  Random weights give a degenerate spectrum (four eigenvalues within
3e-4), so the per-iteration sort by eigenvalue estimate permutes the directions freely and the per-index correlations between
consecutive timesteps come out ~0 or ~+1: the sign rule is exercised by the CPU fixtures of the reference's own scripts
(tests/golden/pc_cli.npz), this fixture records what it did here (`corrs`, `flips`) for the cases where the ordering agrees.

Why the END of the window: the finite-difference power iteration amplifies the ~4e-6 deviation between two correct fp32 forwards by
a factor that grows with the noise level (x0_hat's norm over the Jacobian's gain); at the window's start (t = 596) five iterations
leave HIP and CPU 5e-2 apart per direction on random weights (principal cosine 0.9973, measured in round 6), at its end (t ~ 400,
where oracle/make_fullsize_pc_golden.py also sits) 1e-2.  The computation is the same at every timestep.

    PYTHONPATH=. python oracle/make_fullsize_pc_chain_golden.py        # ~6 min of CPU -> tests/golden/fullsize_pc_chain.npz"""
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

# CONST as in oracle/make_fullsize_pc_golden.py (a finite difference of step c amplifies the deviation between two correct fp32
# forwards by ~1e-3 / c per un-contracting iteration at this size)
T, DRIFT_START, DRIFT_END, N_EV, ITERS, CONST, CFG, AMOUNT, CORR_TO_SWAP = 200, 83, 80, 4, 5, 0.3, 3.0, 1.5, 0.3
EVS = (1, 2)
IT0, IT1 = T - DRIFT_START, T - DRIFT_END          # window iterations [117, 120)


def inputs():
    """Seeded inputs in the order the GPU test regenerates them: conditioning (uncond, text), the trajectory's latents in the
    reference's layout (latents[0] = x_T, latents[it + 1] = noise map of step it; main_pc_extract_inv.py:178-180), the start
    vectors of the first window step."""
    g = torch.Generator().manual_seed(22)
    mk = lambda L1: dict(encoder_hidden_states=torch.randn(1, 8, 768, generator=g),          # noqa: E731
                         encoder_hidden_states_1=torch.randn(1, L1, 1024, generator=g),
                         encoder_attention_mask_1=torch.ones(1, L1))
    unc, txt = mk(1), mk(9)
    latents = [torch.randn(1, 8, 256, 16, generator=g) for _ in range(T + 1)]
    init0 = torch.randn(N_EV, 8, 256, 16, generator=g)
    return unc, txt, latents, init0


def main():
    cfg = configs.FAMILIES["audioldm2"]["unet"]
    sd = weights.random_state_dict(weights.unet_param_shapes(cfg), seed=0)
    unc, txt, latents, init0 = inputs()
    osched = OracleDDIMScheduler()
    osched.set_timesteps(T)
    ow = oloops.OracleWrapper(osched, lambda x, tt, c: ounet.unet_forward(
        cfg, sd, x, tt, **{k: v.expand(x.shape[0], *v.shape[1:]) for k, v in c.items()})[0])
    mask = torch.ones_like(latents[0])
    t0 = time.time()
    with torch.inference_mode():
        # ---- extraction trajectory (main_pc_extract_inv.py:199-256)
        xt = latents[0]
        xts, evs, vals, corrs, flips = {}, [], [], [], []
        prev_pc = None
        for it in range(IT1):
            t = osched.timesteps[it]
            xtm1, x0p = opc.forward_directional(ow, xt, t, latents[it + 1], unc, txt, CFG, eta=1.0)
            if it >= IT0:
                xts[it] = xt.clone()
                init = init0 if prev_pc is None else -prev_pc
                ev, val, _, _ = opc.get_eigenvectors(ow, xt, txt, unc, latents[it + 1], mask, t, x0p * mask, init, const=CONST,
                                                     cfg_tar=CFG, iters=ITERS, eta=1.0, n_ev=N_EV)
                ev, val = ev.clone(), torch.as_tensor(val).reshape(-1).clone()
                if it > IT0:
                    corr = (prev_pc.reshape(N_EV, -1) @ ev.reshape(N_EV, -1).T).diag().clone()
                    flip = corr <= -CORR_TO_SWAP
                    ev[flip] *= -1
                    corr[flip] *= -1
                    corrs.append(corr)
                    flips.append(flip)
                prev_pc = ev
                evs.append(ev)
                vals.append(val)
                print(f"it {it} (t = {int(t)}): eigenvalues {val.tolist()}"
                      + (f", corr {corrs[-1].tolist()}, flipped {flips[-1].tolist()}" if it > IT0 else "")
                      + f"   [{time.time() - t0:.0f} s]", flush=True)
            elif it % 20 == 0:
                print(f"lead-in step {it}   [{time.time() - t0:.0f} s]", flush=True)
            xt = xtm1
        xts[IT1] = xt.clone()
        # ---- drifted trajectory (main_pc_apply_drift.py:141-191), the window only: before it the two trajectories coincide
        xd = xts[IT0]
        drifted = []
        for j, it in enumerate(range(IT0, IT1)):
            t = osched.timesteps[it]
            xtm1, x0p = opc.forward_directional(ow, xd, t, latents[it + 1], unc, txt, CFG, eta=1.0)
            xd = opc.apply_drift(ow, xtm1, x0p, t, evs[j], vals[j], latents[it + 1], amount=AMOUNT, eta=1.0, ev_nums=EVS)
            drifted.append(xd.clone())
    print(f"oracle: {IT0} lead-in steps + {IT1 - IT0} window steps x {ITERS} iterations + the drifted window in {time.time() - t0:.0f} s")
    out = os.path.join(ROOT, "tests", "golden", "fullsize_pc_chain.npz")
    np.savez_compressed(
        out, T=np.array(T), drift=np.array([DRIFT_START, DRIFT_END]), n_ev=np.array(N_EV), iters=np.array(ITERS),
        const=np.array(CONST), corr_to_swap=np.array(CORR_TO_SWAP), amount=np.array(AMOUNT), evs=np.array(EVS),
        timesteps=np.array([int(osched.timesteps[it]) for it in range(IT0, IT1)]),
        xts=torch.cat([xts[it] for it in range(IT0, IT1 + 1)]).numpy(),
        eigvec=torch.stack(evs).numpy(), eigval=torch.stack(vals).numpy(),
        corrs=torch.stack(corrs).numpy(), flips=torch.stack(flips).numpy(), drifted=torch.cat(drifted).numpy())
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()

"""BASELINE config 4 at FULL size on one MI355X: unsupervised PC extraction + application on AudioLDM2 (346.9 M U-Net,
seeded-random weights), 10 s clip, T=200, n_evs=4, 50 power iterations per timestep, drift window 120 -> 80
(main_pc_extract_inv.py:95-262, main_pc_apply_drift.py:71-199): the 200-step inversion, the guided replay with a block
power iteration on the U-Net Jacobian (finite differences, 2 x n_evs sample-forwards per iteration) at each of the 40 window
timesteps, then the drift applied per PC.  Prints ONE JSON line: seconds per run, U-Net sample-forwards per second, and
size-independent checks of the result (finiteness, positive descending eigenvalues, orthonormal PCs, power-iteration
convergence, the drift moves the sample).

    PYTHONPATH=. python tools/bench_config4.py [--iters 50] [--n_evs 4] [--T 200] [--drift_start 120] [--drift_end 80]"""
import argparse
import json
import sys
import time
from argparse import Namespace

import torch

ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=200)
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--n_evs", type=int, default=4)
ap.add_argument("--drift_start", type=int, default=120)
ap.add_argument("--drift_end", type=int, default=80)
ap.add_argument("--model_id", default="cvssp/audioldm2")
a = ap.parse_args()

from audioeditingcode_amd import main_pc_apply_drift as papply, main_pc_extract_inv as pext, models    # noqa: E402
from audioeditingcode_amd.utils import load_audio, synthetic_clip                                      # noqa: E402

dev = "cuda:0"
t0 = time.time()
m = models.load_model(a.model_id, dev, a.T, allow_synthetic=True)
print(f"weights ({m.weights_source}) ready in {time.time() - t0:.1f} s", file=sys.stderr, flush=True)
x0, _, _ = load_audio((synthetic_clip(10.0, seed=1234), 16000), m.get_fn_STFT(), device=dev, stft=True, model_sr=m.get_sr())
with torch.inference_mode():
    w0 = m.vae_encode(x0)
ex = pext.finish_args(Namespace(seed=1, cfg_tar=3, model_id=a.model_id, init_aud=None, num_diffusion_steps=a.T,
                                source_prompt=["a recording of a piano melody"], target_neg_prompt=[""], corr_to_swap=0.8,
                                drift_start=a.drift_start, drift_end=a.drift_end, results_path="unused", const=1e-3,
                                n_evs=a.n_evs, patch=None, iters=a.iters, dry=False))
apa = Namespace(drift_start=a.drift_start, drift_end=a.drift_end, amount=1.5, use_specific_ts_pc=None, fix_alpha=None,
                fade_length=0.0, evs=list(range(1, a.n_evs + 1)), combine_evs=False, evals_pt=None, rand_v=False,
                shift_x0_for_np=True, sub_iters=None)
keys = ("eigdata", "args", "corrs", "in_corrs", "latents", "in_norms", "xts")
torch.manual_seed(1)
torch.cuda.synchronize()
t0 = time.perf_counter()
ck = pext.extract_pcs(m, w0, ex)
torch.cuda.synchronize()
t_ext = time.perf_counter() - t0
t1 = time.perf_counter()
out = papply.apply_pcs(m, {k: ck[k] for k in keys}, apa, torch.device(dev))
torch.cuda.synchronize()
t_app = time.perf_counter() - t1

n_win = len(ck["eigdata"])
fwd = 2 * a.T + 2 * a.T + n_win * a.iters * 2 * a.n_evs + 2 * a.T * a.n_evs       # inversion + replay + power iteration + apply
vals = torch.stack([e["eigval"].reshape(-1) for e in ck["eigdata"].values()])      # [window, n_evs]
vecs = torch.stack([e["eigvec"].reshape(a.n_evs, -1) for e in ck["eigdata"].values()])
gram = vecs @ vecs.transpose(1, 2)
eye = torch.eye(a.n_evs)
last_corr = torch.stack([c[-1].abs().cpu() for c in ck["in_corrs"]])               # |<v_it, v_it-1>| of the last iteration
rel_move = ((out.cpu() - ck["final"].cpu()).flatten(1).norm(dim=1) / ck["final"].cpu().norm()).tolist()
# the chain between timesteps (main_pc_extract_inv.py:204-212): the stored corrs are the per-index correlations of the STORED
# (post-flip) consecutive PCs, and the sign rule left none of them at or below -corr_to_swap
corrs = torch.stack([c.cpu() for c in ck["corrs"]]) if ck["corrs"] else None
chain_err = float(((vecs[:-1] * vecs[1:]).sum(-1) - corrs).abs().max()) if corrs is not None else None
checks = dict(
    window_timesteps=n_win, finite=bool(torch.isfinite(vals).all() and torch.isfinite(vecs).all() and torch.isfinite(out).all()),
    eigenvalues_positive=bool((vals > 0).all()), eigenvalues_descending=bool((vals[:, :-1] >= vals[:, 1:] * (1 - 1e-4)).all()),
    eigval_first_last=[float(vals[0, 0]), float(vals[-1, 0])],
    orthonormality_max_err=float((gram - eye).abs().max()),
    last_iteration_cosine_min=float(last_corr.min()), last_iteration_cosine_median=float(last_corr.median()),
    sign_continuity_min_corr=float(corrs.min()) if corrs is not None else None,
    stored_corrs_vs_stored_pcs_max_err=chain_err,
    sign_rule_holds=bool((corrs > -getattr(a, "corr_to_swap", 0.8)).all()) if corrs is not None else None,
    drift_moves_sample_rel_l2=rel_move)
# pass = the size-independent invariants.  Ordering / convergence of the returned eigenvalues are REPORTED only: the values
# come from the last iteration before its sort (pc_drift.py:146-171, as in the reference), and seeded-random weights have
# no dominant Jacobian directions for 50 power iterations to converge to.
ok = (checks["finite"] and checks["eigenvalues_positive"] and checks["orthonormality_max_err"] < 1e-3
      and min(rel_move) > 1e-4 and (chain_err is None or (chain_err < 1e-3 and checks["sign_rule_holds"])))
print(json.dumps(dict(
    metric="seconds per PC extract + apply run (config 4)", value=t_ext + t_app, unit="s", higher_is_better=False,
    seconds=dict(extract=t_ext, apply=t_app), unet_sample_forwards=fwd, unet_sample_forwards_per_s=fwd / (t_ext + t_app),
    config=dict(workload=f"BASELINE configs[3]: AudioLDM2 ({a.model_id}, seeded-random weights), 10 s clip, T={a.T}, "
                         f"n_evs={a.n_evs}, iters={a.iters}, drift window {a.drift_start}->{a.drift_end}, amount 1.5, "
                         f"PCs applied one by one"),
    checks=checks, checks_pass=ok)))
sys.exit(0 if ok else 3)

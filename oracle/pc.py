"""Oracle restatement of the unsupervised-PC utilities (BASELINE config 4).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Pinned by tests/golden/pc_drift.npz, produced by running the
reference's own /root/reference/code/pc_drift.py here with a synthetic eps-model (oracle/make_golden.py pc).

Follows pc_drift.py: forward_directional :29-93, get_eigenvectors :96-198, apply_drift :201-278.
`w` is an oracle.loops.OracleWrapper; conditioning objects are the opaque `cond` of its unet callable.
"""
import torch


def forward_directional(w, xt, t, latent, cond_uncond, cond_text, cfg_tar, eta=1.0, eigvecs=0.0, amount=0.0,
                        mode="both"):
    """One CFG DDIM/DDPM step through scheduler.step on the (optionally shifted) input; returns (x_{t-1}, x0_hat)."""
    s = w.model.scheduler
    inp = xt + amount * eigvecs * torch.sqrt(s.alphas_cumprod[t])
    eps_u = w.unet(inp if mode in ("both", "uncond") else xt, t, cond_uncond)
    eps_c = w.unet(inp if mode in ("both", "text") else xt, t, cond_text)
    eps = eps_u + cfg_tar * (eps_c - eps_u)
    res = s.step(eps, t, inp, eta=eta, variance_noise=latent)
    return res.prev_sample, res.pred_original_sample


def get_sigma(w, t):
    return torch.sqrt(1.0 / w.model.scheduler.alphas_cumprod - 1)[t]


def get_eigenvectors(w, xt, cond_text, cond_uncond, latents, mask, t, x0_pred, init, const=1e-3, cfg_tar=3.0,
                     iters=50, eta=1.0, n_ev=1, mode="both"):
    """Subspace (power) iteration on the Jacobian of the posterior mean by finite differences.
    `init`: the randn tensor the reference draws with randn_like(xt) (after expansion to n_ev)."""
    if n_ev > 1:
        x0_pred = x0_pred.repeat(n_ev, 1, 1, 1)
        xt = xt.repeat(n_ev, 1, 1, 1)
    eig = init * mask * const
    prev = eig.clone()
    in_corr, in_norm = [], []
    for i in range(iters):
        _, out = forward_directional(w, xt, t, latents, cond_uncond, cond_text, cfg_tar, eta=eta, eigvecs=eig, amount=1,
                                     mode=mode)
        Ab = out * mask - x0_pred
        if n_ev > 1:
            nrm = Ab[:, mask[0].to(torch.bool)].norm(dim=1)
            eig = (Ab / nrm.reshape(n_ev, 1, 1, 1)) * mask
            Q, R = torch.linalg.qr(eig.permute(1, 2, 3, 0).reshape(-1, n_ev), mode="reduced")
            if torch.prod(torch.linalg.diagonal(R)) < 0:
                Q = Q * -1
            eig = (Q / Q.norm(dim=0)).T.reshape(Ab.shape)
            _, order = (nrm / const * (get_sigma(w, t) ** 2)).reshape(n_ev).sort(descending=True, stable=True)
            eig = eig[order, ...]
        else:
            nrm = Ab[mask.to(torch.bool)].norm()
            eig = (Ab / nrm) * mask
        if i > 0:
            in_corr.append((prev.reshape(n_ev, -1) @ eig.reshape(n_ev, -1).T).diag())
        in_norm.append(nrm)
        prev = eig.clone()
        eig = eig * const
    eigval = nrm / const * (get_sigma(w, t) ** 2)
    return eig / const, eigval, in_corr, in_norm


def apply_drift(w, xt_m1, x0_pred, t, eigvec, eigval, latent, amount=1.0, eta=1.0, ev_nums=(1,),
                use_shifted_x0_for_noisepred=True):
    """Shift x0_hat along sqrt(lambda)*v and re-derive x_{t-1} (pc_drift.py:236-278)."""
    s = w.model.scheduler
    shift = 0
    for ev in ev_nums:
        shift = shift + amount * (eigval[ev - 1].unsqueeze(0).sqrt() * eigvec[ev - 1].unsqueeze(0))
    x0_drift = x0_pred.clone() + shift
    prev_t = t - s.config.num_train_timesteps // s.num_inference_steps
    var = s._get_variance(t, prev_t)
    std = eta * var ** 0.5
    a_prev = s.alphas_cumprod[prev_t] if prev_t >= 0 else s.final_alpha_cumprod
    a_t = s.alphas_cumprod[t]
    b_t = 1 - a_t
    if eta > 0:
        xt_m1 = xt_m1 - std * latent
    pred_eps = (xt_m1 - a_prev ** 0.5 * x0_pred) / ((1 - a_prev - std ** 2) ** 0.5)
    if use_shifted_x0_for_noisepred:
        pred_eps = pred_eps - (a_t ** 0.5) / (b_t ** 0.5) * shift
    xt_m1 = a_prev ** 0.5 * x0_drift + (1 - a_prev - std ** 2) ** 0.5 * pred_eps
    if eta > 0:
        xt_m1 = xt_m1 + std * latent
    return xt_m1


def sdedit_loop(w, w0, cond_text, cond_uncond, cfg_tar, skip, eta=1.0):
    """SDEdit baseline (main_run_sdedit.py:78-100): draws len(timesteps)+1 latents, THEN the add_noise noise, from the
    global torch RNG in the reference's order; noises w0 to timesteps[skip] and samples down with
    forward_directional."""
    s = w.model.scheduler
    ts = s.timesteps
    latents = [torch.randn(w0.shape, dtype=w0.dtype) * s.init_noise_sigma for _ in range(len(ts) + 1)]
    ts = ts[skip:]
    latents = latents[skip + 1:]
    noise = torch.randn_like(w0)
    xt = s.add_noise(w0, noise, ts[:1].unsqueeze(0))
    for it, t in enumerate(ts):
        xt, _ = forward_directional(w, xt, t, latents[it], cond_uncond, cond_text, cfg_tar, eta=eta)
    return xt

"""Unsupervised principal-component editing utilities (BASELINE config 4), same call
signatures as the reference's code/pc_drift.py: forward_directional :29-93, get_eigenvectors :96-198,
apply_drift :201-278.

The expensive part -- `iters x 2 x n_ev` U-Net sample-forwards per timestep -- runs on the native U-Net tape:
the unconditional and the conditional pass of one power-iteration step are evaluated as ONE batch of 2*n_ev
samples (the reference issues two batch-n_ev calls).  The small dense algebra around it (masked norms, the
tall-skinny QR of the n_ev directions, the eigenvalue sort) is elementwise / LAPACK-style plumbing on device
tensors.
"""
from enum import Enum
from typing import Dict, List, Optional, Tuple

import torch

from .editing import Conditioning
from .utils import PromptEmbeddings


class PCStreamChoice(Enum):
    BOTH = 1
    TEXT = 2
    UNCOND = 3


def expand_for_evs(x: Optional[torch.Tensor], n_ev: int) -> Optional[torch.Tensor]:
    if x is None:
        return x
    return x.repeat(n_ev, *[1] * (len(x.shape) - 1)).to(x.device)


def _expand_emb(e: PromptEmbeddings, n: int) -> PromptEmbeddings:
    return PromptEmbeddings(embedding_hidden_states=expand_for_evs(e.embedding_hidden_states, n),
                            boolean_prompt_mask=expand_for_evs(e.boolean_prompt_mask, n),
                            embedding_class_lables=expand_for_evs(e.embedding_class_lables, n))


def _to_cond(model, e: PromptEmbeddings) -> Conditioning:
    if model.kind == "audioldm2":
        return Conditioning(ehs0=e.embedding_hidden_states, ehs1=e.embedding_class_lables, mask1=e.boolean_prompt_mask)
    if model.kind == "audioldm":
        return Conditioning(class_labels=e.embedding_class_lables)
    return Conditioning(ehs0=e.embedding_hidden_states, mask0=e.boolean_prompt_mask)


def forward_directional(ldm_stable, xt: torch.Tensor, timestep: torch.Tensor, latent: torch.Tensor,
                        uncond_emb: PromptEmbeddings, text_emb: PromptEmbeddings, cfg_tar, eta: float = 1,
                        eigvecs=0, amount: float = 0, double_precision: bool = False,
                        mode: PCStreamChoice = PCStreamChoice.BOTH):
    """One classifier-free-guided scheduler step from `xt` displaced by amount*eigvecs*sqrt(abar_t) on the chosen
    stream(s) (pc_drift.py:29-93).  The uncond and cond rows go through the U-Net as ONE batch.
    Returns (x_{t-1}, x0_hat)."""
    if double_precision:
        raise NotImplementedError("double_precision=True: the native path is fp32")
    sched = ldm_stable.model.scheduler
    with torch.no_grad():
        displaced = xt + amount * eigvecs * torch.sqrt(sched.alphas_cumprod[int(timestep)])
    rows = len(xt)
    if rows > 1:                              # one embedding row per sample (batched PCs)
        single = lambda e: any(v is not None and len(v) == 1 for v in e)      # noqa: E731
        uncond_emb = _expand_emb(uncond_emb, rows) if single(uncond_emb) else uncond_emb
        text_emb = _expand_emb(text_emb, rows) if single(text_emb) else text_emb
    on_uncond = mode in (PCStreamChoice.BOTH, PCStreamChoice.UNCOND)
    on_text = mode in (PCStreamChoice.BOTH, PCStreamChoice.TEXT)
    eps_u, eps_c = ldm_stable.unet_forward_pair(displaced if on_uncond else xt, displaced if on_text else xt, timestep,
                                                _to_cond(ldm_stable, uncond_emb), _to_cond(ldm_stable, text_emb))
    step = sched.step(eps_u + cfg_tar * (eps_c - eps_u), timestep, displaced, eta=eta, variance_noise=latent)
    return step.prev_sample, step.pred_original_sample


# ---------------------------------------------------------------------------------------------- subspace iteration
def _masked_lengths(v: torch.Tensor, mask: torch.Tensor, k: int) -> torch.Tensor:
    """Euclidean length of each of the k directions over the masked region (mask is shared by all directions)."""
    if k > 1:
        return v[:, mask[0].to(torch.bool)].norm(dim=1)
    return v[mask.to(torch.bool)].norm()


def _orthonormal_rows(dirs: torch.Tensor) -> torch.Tensor:
    """Re-orthonormalise k directions [k, ...] with a thin QR of the (numel x k) matrix; sign convention: the product
    of R's diagonal is kept positive (pc_drift.py:162-169)."""
    k = dirs.shape[0]
    q, r = torch.linalg.qr(dirs.reshape(k, -1).T, mode="reduced")
    if torch.prod(torch.linalg.diagonal(r)) < 0:
        q = -q
    return (q / q.norm(dim=0)).T.reshape(dirs.shape)


def get_eigenvectors(ldm_stable, xt: torch.Tensor, text_emb: PromptEmbeddings, uncond_emb: PromptEmbeddings,
                     latents: torch.Tensor, mask: torch.Tensor, t: torch.Tensor, x0_pred: torch.Tensor,
                     pc_mode: PCStreamChoice = PCStreamChoice.BOTH, const: float = 1e-3, cfg_tar: float = 3,
                     iters: int = 50, double_precision: bool = False, eta: float = 1, n_ev: int = 1,
                     init_eigvecs: Optional[torch.Tensor] = None) -> Tuple:
    """Leading `n_ev` principal components of the posterior covariance at timestep t (pc_drift.py:96-198): block power
    iteration on the Jacobian of x0_hat, applied by finite differences of step `const`.  All n_ev directions -- and
    their uncond + cond evaluations -- are ONE U-Net batch per iteration.  `init_eigvecs` (not in the reference
    signature) replaces the randn_like draw so a run can be reproduced across devices.
    Returns (eigvecs, eigvals, in_corr, in_norm, interm_eigvecs, interm_eigvals)."""
    k = n_ev
    if k > 1:
        xt, x0_pred = expand_for_evs(xt, k), expand_for_evs(x0_pred, k)
        uncond_emb, text_emb = _expand_emb(uncond_emb, k), _expand_emb(text_emb, k)
    tail = [1] * (xt.dim() - 1)
    to_eigval = ldm_stable.get_sigma(int(t)) ** 2 / const        # |J d| / const * sigma_t^2
    start = torch.randn_like(xt) if init_eigvecs is None else init_eigvecs.to(xt.device)
    probe = start * mask * const
    previous = probe.detach().clone()
    in_corr, in_norm, interm_eigvecs, interm_eigvals = [], [], {}, {}
    unit = lengths = None
    with torch.no_grad():
        for it in range(iters):
            _, x0_moved = forward_directional(ldm_stable, xt, t, latents, uncond_emb, text_emb, cfg_tar, eta=eta,
                                              eigvecs=probe, amount=1, mode=pc_mode)
            jd = x0_moved * mask - x0_pred                      # J . probe (masked)
            lengths = _masked_lengths(jd, mask, k)
            if k > 1:
                unit = _orthonormal_rows((jd / lengths.reshape(k, *tail)) * mask)
                unit = unit[(lengths * to_eigval).reshape(k).sort(descending=True, stable=True)[1], ...]
            else:
                unit = (jd / lengths) * mask
            if it > 0:
                in_corr.append((previous.reshape(k, -1) @ unit.reshape(k, -1).T).diag())
            in_norm.append(lengths)
            previous = unit.detach().clone()
            if it > 15 and it % 10 == 0:
                interm_eigvecs[it] = unit
                interm_eigvals[it] = lengths * to_eigval
            probe = unit * const                                # keep the finite difference in its linear regime
    return probe / const, lengths * to_eigval, in_corr, in_norm, interm_eigvecs, interm_eigvals


# ---------------------------------------------------------------------------------------------- applying a drift
def _stored_pc(eigdata, t, timesteps, num_diff_steps, use_specific_ts_pc, sub_iters, evals, device):
    """Direction(s) and eigenvalue(s) to use at timestep t: vectors may come from another timestep
    (`use_specific_ts_pc`) or an intermediate iterate (`sub_iters`), values from an external table (`evals`)."""
    vec_t = int(t) if use_specific_ts_pc is None else int(timesteps[num_diff_steps - use_specific_ts_pc])
    if sub_iters is not None:
        if evals is not None:
            raise ValueError("evals should be None if sub_iters is not None")
        return (eigdata[vec_t]["interm_eigvecs"][sub_iters].to(device),
                eigdata[int(t)]["interm_eigvals"][sub_iters].to(device))
    vals = eigdata[int(t)]["eigval"].to(device) if evals is None else torch.from_numpy(evals[int(t)]).to(device)
    return eigdata[vec_t]["eigvec"].to(device), vals


def apply_drift(ldm_stable, xt_m1: torch.Tensor, x0_pred: torch.Tensor, t: torch.Tensor, timesteps: torch.Tensor,
                num_diff_steps: int, eigdata: Dict[int, Dict[str, torch.Tensor]], latent: torch.Tensor,
                device: torch.device, use_shifted_x0_for_noisepred: bool = True,
                use_specific_ts_pc: Optional[int] = None, amount: float = 1, sub_iters: Optional[int] = None,
                eta: float = 1, ev_nums: List[int] = [1], evals: Optional[Dict[int, torch.Tensor]] = None):
    """Move x0_hat by amount * sum_e sqrt(lambda_e) v_e and rebuild x_{t-1} around it (pc_drift.py:201-278): strip the
    step's noise term, recover the direction term implied by (x_{t-1}, x0_hat), optionally correct it for the moved
    x0, and recompose."""
    vecs, vals = _stored_pc(eigdata, t, timesteps, num_diff_steps, use_specific_ts_pc, sub_iters, evals, device)
    shift = 0
    for e in ev_nums:
        shift = shift + amount * (vals[e - 1].unsqueeze(0).sqrt() * vecs[e - 1].unsqueeze(0))
    sched = ldm_stable.model.scheduler
    prev_t = int(t) - sched.config.num_train_timesteps // sched.num_inference_steps
    std = float(eta * sched._get_variance(int(t), prev_t) ** 0.5)
    abar_prev = float(sched.alphas_cumprod[prev_t] if prev_t >= 0 else sched.final_alpha_cumprod)
    abar_t = float(sched.alphas_cumprod[int(t)])
    dir_scale = (1 - abar_prev - std ** 2) ** 0.5
    moved_x0 = x0_pred.clone() + shift
    mean = xt_m1 - std * latent if eta > 0 else xt_m1
    eps_hat = (mean - abar_prev ** 0.5 * x0_pred) / dir_scale
    if use_shifted_x0_for_noisepred:
        eps_hat = eps_hat - (abar_t ** 0.5) / ((1 - abar_t) ** 0.5) * shift
    out = abar_prev ** 0.5 * moved_x0 + dir_scale * eps_hat
    return out + std * latent if eta > 0 else out

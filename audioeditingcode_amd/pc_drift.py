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
    if double_precision:
        raise NotImplementedError("double_precision=True: the native path is fp32")
    sched = ldm_stable.model.scheduler
    with torch.no_grad():
        inp = xt + amount * eigvecs * torch.sqrt(sched.alphas_cumprod[int(timestep)])
    n = len(xt)
    def needs(e):
        return any(v is not None and len(v) == 1 for v in e)
    if n > 1 and needs(uncond_emb):
        uncond_emb = _expand_emb(uncond_emb, n)
    if n > 1 and needs(text_emb):
        text_emb = _expand_emb(text_emb, n)
    x_u = inp if mode in (PCStreamChoice.BOTH, PCStreamChoice.UNCOND) else xt
    x_c = inp if mode in (PCStreamChoice.BOTH, PCStreamChoice.TEXT) else xt
    eps_u, eps_c = ldm_stable.unet_forward_pair(x_u, x_c, timestep, _to_cond(ldm_stable, uncond_emb),
                                                _to_cond(ldm_stable, text_emb))
    noise_pred = eps_u + cfg_tar * (eps_c - eps_u)
    res = sched.step(noise_pred, timestep, inp, eta=eta, variance_noise=latent)
    return res.prev_sample, res.pred_original_sample


def get_eigenvectors(ldm_stable, xt: torch.Tensor, text_emb: PromptEmbeddings, uncond_emb: PromptEmbeddings,
                     latents: torch.Tensor, mask: torch.Tensor, t: torch.Tensor, x0_pred: torch.Tensor,
                     pc_mode: PCStreamChoice = PCStreamChoice.BOTH, const: float = 1e-3, cfg_tar: float = 3,
                     iters: int = 50, double_precision: bool = False, eta: float = 1, n_ev: int = 1,
                     init_eigvecs: Optional[torch.Tensor] = None) -> Tuple:
    """Subspace iteration on the posterior-mean Jacobian by finite differences.  `init_eigvecs` (optional, not in
    the reference signature) replaces the randn_like draw so a run can be reproduced across devices."""
    if n_ev > 1:
        x0_pred = expand_for_evs(x0_pred, n_ev)
        xt = expand_for_evs(xt, n_ev)
        uncond_emb, text_emb = _expand_emb(uncond_emb, n_ev), _expand_emb(text_emb, n_ev)
    eigvecs = (torch.randn_like(xt) if init_eigvecs is None else init_eigvecs.to(xt.device)) * mask * const
    prev_ev = eigvecs.detach().clone()
    in_corr, in_norm, interm_eigvecs, interm_eigvals = [], [], {}, {}
    sigma2 = ldm_stable.get_sigma(int(t)) ** 2
    with torch.no_grad():
        for i in range(iters):
            _, unmasked = forward_directional(ldm_stable, xt, t, latents, uncond_emb, text_emb, cfg_tar, eta=eta,
                                              eigvecs=eigvecs, amount=1, mode=pc_mode)
            Ab = unmasked * mask - x0_pred
            if n_ev > 1:
                perm = {4: (1, 2, 3, 0), 3: (1, 2, 0), 2: (1, 0)}[len(xt.shape)]
                norm_of_Ab = Ab[:, mask[0].to(torch.bool)].norm(dim=1)
                eigvecs = (Ab / norm_of_Ab.reshape(n_ev, *[1] * (len(xt.shape) - 1))) * mask
                Q, R = torch.linalg.qr(eigvecs.permute(*perm).reshape(-1, n_ev), mode="reduced")
                if torch.prod(torch.linalg.diagonal(R)) < 0:
                    Q = Q * -1
                eigvecs = (Q / Q.norm(dim=0)).T.reshape(Ab.shape)
                _, order = (norm_of_Ab / const * sigma2).reshape(n_ev).sort(descending=True, stable=True)
                eigvecs = eigvecs[order, ...]
            else:
                norm_of_Ab = Ab[mask.to(torch.bool)].norm()
                eigvecs = (Ab / norm_of_Ab) * mask
            if i > 0:
                in_corr.append((prev_ev.reshape(n_ev, -1) @ eigvecs.reshape(n_ev, -1).T).diag())
            in_norm.append(norm_of_Ab)
            prev_ev = eigvecs.detach().clone()
            if not (i % 10) and i > 15:
                interm_eigvecs[i] = eigvecs
                interm_eigvals[i] = norm_of_Ab / const * sigma2
            eigvecs = eigvecs * const
    eigval = norm_of_Ab / const * sigma2
    return eigvecs / const, eigval, in_corr, in_norm, interm_eigvecs, interm_eigvals


def apply_drift(ldm_stable, xt_m1: torch.Tensor, x0_pred: torch.Tensor, t: torch.Tensor, timesteps: torch.Tensor,
                num_diff_steps: int, eigdata: Dict[int, Dict[str, torch.Tensor]], latent: torch.Tensor,
                device: torch.device, use_shifted_x0_for_noisepred: bool = True,
                use_specific_ts_pc: Optional[int] = None, amount: float = 1, sub_iters: Optional[int] = None,
                eta: float = 1, ev_nums: List[int] = [1], evals: Optional[Dict[int, torch.Tensor]] = None):
    use_t = int(t) if use_specific_ts_pc is None else int(timesteps[num_diff_steps - use_specific_ts_pc])
    eigvec = eigdata[use_t]["eigvec"].to(device)
    eigval = eigdata[int(t)]["eigval"].to(device) if evals is None else torch.from_numpy(evals[int(t)]).to(device)
    if sub_iters is not None:
        if evals is not None:
            raise ValueError("evals should be None if sub_iters is not None")
        eigvec = eigdata[use_t]["interm_eigvecs"][sub_iters].to(device)
        eigval = eigdata[int(t)]["interm_eigvals"][sub_iters].to(device)
    shift_by = 0
    for ev_num in ev_nums:
        shift_by = shift_by + amount * (eigval[ev_num - 1].unsqueeze(0).sqrt() * eigvec[ev_num - 1].unsqueeze(0))
    x0_drift = x0_pred.clone() + shift_by
    sched = ldm_stable.model.scheduler
    prev_t = int(t) - sched.config.num_train_timesteps // sched.num_inference_steps
    std = float(eta * sched._get_variance(int(t), prev_t) ** 0.5)
    a_prev = float(sched.alphas_cumprod[prev_t] if prev_t >= 0 else sched.final_alpha_cumprod)
    a_t = float(sched.alphas_cumprod[int(t)])
    if eta > 0:
        xt_m1 = xt_m1 - std * latent
    pred_eps = (xt_m1 - a_prev ** 0.5 * x0_pred) / ((1 - a_prev - std ** 2) ** 0.5)
    if use_shifted_x0_for_noisepred:
        pred_eps = pred_eps - (a_t ** 0.5) / ((1 - a_t) ** 0.5) * shift_by
    xt_m1 = a_prev ** 0.5 * x0_drift + (1 - a_prev - std ** 2) ** 0.5 * pred_eps
    if eta > 0:
        xt_m1 = xt_m1 + std * latent
    return xt_m1

"""SDEdit baseline: the loop body of the reference's code/main_run_sdedit.py:78-100 --
noise the encoded clip to timesteps[skip] with scheduler.add_noise, then run the classifier-free-guided
DDPM sampler (`forward_directional` = scheduler.step with fresh noise) down to t=0.

On this engine that is the same device-resident reverse loop as the edit (editing.EditEngine.edit): the
"recorded" noise maps are simply fresh draws.  Valid for eta in {0, 1}, where the reference's step
(eta*var under the root, models.py:148) and scheduler.step (std^2 under the root) coincide.
"""
from typing import List, Optional

import torch

from .ddm_inversion.inversion_utils import conditioning_from_text


@torch.no_grad()
def sdedit(ldm_stable, w0: torch.Tensor, target_prompt: List[str], target_neg_prompt: List[str], cfg_tar: float,
           skip: int, eta: float = 1.0, latents: Optional[List[torch.Tensor]] = None,
           noise: Optional[torch.Tensor] = None) -> torch.Tensor:
    """w0 [1,C,H,W] encoded clip -> edited latent [1,C,H,W].  `latents` (one per remaining step, in sampling
    order) and `noise` default to CPU-generator draws like everything else on this path."""
    if eta not in (0, 0.0, 1, 1.0):
        raise NotImplementedError("sdedit on the native loop supports eta in {0, 1}")
    sched = ldm_stable.model.scheduler
    ts = sched.timesteps[skip:]
    Z = len(ts)
    if latents is None:
        # the reference's draw order (main_run_sdedit.py:79-92): one latent per scheduler timestep plus one, sliced
        # to the remaining steps, THEN the add_noise draw -- on the CPU generator like every draw of this path
        T = len(sched.timesteps)
        latents = [torch.randn(w0.shape) * sched.init_noise_sigma for _ in range(T + 1)][skip + 1:]
    if noise is None:
        noise = torch.randn(w0.shape)
    xt = sched.add_noise(w0.to(ldm_stable.device), noise.to(ldm_stable.device), ts[:1].unsqueeze(0))
    ed = ldm_stable.editor(w0.shape[-2], w0.shape[-1])
    tgt = conditioning_from_text(ldm_stable, ldm_stable.encode_text(target_prompt))
    neg = conditioning_from_text(ldm_stable, ldm_stable.encode_text(target_neg_prompt))
    xt_c = ed.to_nhwc(xt.reshape(1, *xt.shape[-3:]))                         # [1,H,W,C]
    xts_like = xt_c.unsqueeze(0).expand(Z + 1, *xt_c.shape)                  # edit() starts from index Z
    # edit() consumes zs[Z - it - 1] at step it; the reference uses latents[it]
    zs = ed.to_nhwc(torch.stack([latents[Z - 1 - i].reshape(1, *w0.shape[-3:]) for i in range(Z)]))
    out = ed.edit(xts_like, zs, Z, tgt, neg, [cfg_tar], eta=eta)
    return ed.to_nchw(out)

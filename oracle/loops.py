"""Oracle restatement of the editing loops and step math.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Pinned by golden vectors produced
by the reference itself (oracle/make_golden.py -> tests/golden/loop_*.npz).

Follows (all paths relative to /root/reference/code):
  models.py:67-83     sample_xts_from_x0      -> OracleWrapper.sample_xts_from_x0
  models.py:85-117    get_zs_from_xts         -> OracleWrapper.get_zs_from_xts
  models.py:119-158   reverse_step_with_custom_noise
  models.py:679-689   get_variance / get_alpha_prod_t_prev
  ddm_inversion/inversion_utils.py:8-144    inversion_forward_process -> invert()
  ddm_inversion/inversion_utils.py:147-323  inversion_reverse_process -> edit()
  ddm_inversion/ddim_inversion.py:10-84     next_step / ddim_inversion / text2image_ldm_stable
The multi-prompt mask blur (inversion_utils.py:49,197-198) uses torchvision's
gaussian_blur, absent here: restated from its documented definition (reflect pad,
separable normalised Gaussian), PARITY UNPINNED for P>1 masks.
"""
from types import SimpleNamespace

import torch
import torch.nn.functional as F


def gaussian_blur(x, kernel_size=15, sigma=1.0):
    half = (kernel_size - 1) * 0.5
    g = torch.linspace(-half, half, kernel_size)
    pdf = torch.exp(-0.5 * (g / sigma) ** 2)
    k1 = pdf / pdf.sum()
    k2 = (k1[:, None] * k1[None, :]).to(x.dtype)
    c = x.shape[-3]
    pad = kernel_size // 2
    xp = F.pad(x, (pad, pad, pad, pad), mode="reflect")
    return F.conv2d(xp, k2.expand(c, 1, kernel_size, kernel_size), groups=c)


def segment_scales(batch, shape, scales, cutoff_points, dtype, zero_empty=None):
    """cfg / mask tensors of inversion_utils.py:29-51 and :177-200."""
    cfg = torch.ones((batch, *shape), dtype=dtype)
    masks = torch.ones((batch, *shape), dtype=dtype)
    if batch > 1:
        if cutoff_points is None:
            cutoff_points = [i * 1 / batch for i in range(1, batch)]
        scales = list(scales)
        if len(scales) == 1:
            scales = scales * batch
        elif len(scales) < batch:
            raise ValueError("Not enough target CFG scales")
        cuts = [0, *[int(c * cfg.shape[2]) for c in cutoff_points], cfg.shape[2]]
        for i, (s, e) in enumerate(zip(cuts[:-1], cuts[1:])):
            cfg[i, :, e:] = 0
            cfg[i, :, :s] = 0
            masks[i, :, e:] = 0
            masks[i, :, :s] = 0
            cfg[i] *= scales[i]
            if zero_empty is not None and zero_empty[i]:
                cfg[i] = 0
        cfg = gaussian_blur(cfg)
        masks = gaussian_blur(masks)
    else:
        cfg *= scales[0]
    return cfg, masks


class OracleWrapper:
    """Duck-type of models.PipelineWrapper's scheduler-facing methods.

    `unet` is any callable (x[B,C,H,W], t, cond) -> eps[B,C,H,W]; `cond` is the
    opaque conditioning triple the loops thread through.
    """

    def __init__(self, scheduler, unet, in_channels=8):
        self.model = SimpleNamespace(scheduler=scheduler)
        self.unet = unet
        self.in_channels = in_channels

    # models.py:679-689
    def get_alpha_prod_t_prev(self, prev_t):
        s = self.model.scheduler
        return s.alphas_cumprod[prev_t] if prev_t >= 0 else s.final_alpha_cumprod

    def get_variance(self, t, prev_t):
        s = self.model.scheduler
        a_t = s.alphas_cumprod[t]
        a_p = self.get_alpha_prod_t_prev(prev_t)
        return ((1 - a_p) / (1 - a_t)) * (1 - a_t / a_p)

    def _prev(self, t):
        s = self.model.scheduler
        return t - s.config.num_train_timesteps // s.num_inference_steps

    # models.py:67-83 -- independent noise per timestep, ascending t
    def sample_xts_from_x0(self, x0, num_inference_steps, generator=None):
        s = self.model.scheduler
        abar = s.alphas_cumprod
        sq1m = (1 - abar) ** 0.5
        ts = s.timesteps
        t_to_idx = {int(v): k for k, v in enumerate(ts)}
        xts = torch.zeros((num_inference_steps + 1, self.in_channels, x0.shape[-2], x0.shape[-1]))
        xts[0] = x0
        for t in reversed(ts):
            idx = num_inference_steps - t_to_idx[int(t)]
            noise = torch.randn(x0.shape, generator=generator, dtype=x0.dtype)
            xts[idx] = x0 * (abar[t] ** 0.5) + noise * sq1m[t]
        return xts

    # models.py:85-117
    def get_zs_from_xts(self, xt, xtm1, eps, t, eta=0, numerical_fix=True):
        s = self.model.scheduler
        abar = s.alphas_cumprod
        if s.config.prediction_type == "epsilon":
            x0_hat = (xt - (1 - abar[t]) ** 0.5 * eps) / abar[t] ** 0.5
            direction_eps = eps
        else:  # v_prediction
            x0_hat = (abar[t] ** 0.5) * xt - ((1 - abar[t]) ** 0.5) * eps
            direction_eps = (abar[t] ** 0.5) * eps + ((1 - abar[t]) ** 0.5) * xt
        prev_t = self._prev(t)
        a_prev = self.get_alpha_prod_t_prev(prev_t)
        var = self.get_variance(t, prev_t)
        mu = a_prev ** 0.5 * x0_hat + (1 - a_prev - eta * var) ** 0.5 * direction_eps
        z = (xtm1 - mu) / (eta * var ** 0.5)
        if numerical_fix:
            xtm1 = mu + (eta * var ** 0.5) * z
        return z, xtm1

    # models.py:119-158
    def reverse_step_with_custom_noise(self, eps, t, sample, variance_noise=None, eta=0):
        s = self.model.scheduler
        prev_t = self._prev(t)
        a_t = s.alphas_cumprod[t]
        a_prev = self.get_alpha_prod_t_prev(prev_t)
        b_t = 1 - a_t
        if s.config.prediction_type == "epsilon":
            x0_hat = (sample - b_t ** 0.5 * eps) / a_t ** 0.5
            direction_eps = eps
        else:
            x0_hat = (a_t ** 0.5) * sample - (b_t ** 0.5) * eps
            direction_eps = (a_t ** 0.5) * eps + (b_t ** 0.5) * sample
        var = self.get_variance(t, prev_t)
        prev = a_prev ** 0.5 * x0_hat + (1 - a_prev - eta * var) ** 0.5 * direction_eps
        if eta > 0:
            prev = prev + eta * var ** 0.5 * variance_noise
        return prev


def cfg_combine(eps_u, eps_c, cfg_tensor):
    """inversion_utils.py:97-102 / :276-281."""
    b = eps_c.shape[0]
    return eps_u + (cfg_tensor * (eps_c - eps_u.expand(b, -1, -1, -1))).sum(axis=0).unsqueeze(0)


def invert(w, x0, cond_src, cond_uncond, cfg_scales, num_inference_steps, eta=1.0,
           numerical_fix=True, src_is_empty=False, n_prompts=1, cutoff_points=None,
           prompt_empty=None, generator=None, xts=None):
    """inversion_forward_process (inversion_utils.py:8-144). Returns (xt, zs, xts)."""
    s = w.model.scheduler
    T = num_inference_steps
    if not src_is_empty:
        cfg, _ = segment_scales(n_prompts, x0.shape[1:], cfg_scales, cutoff_points, x0.dtype,
                                zero_empty=prompt_empty)
    ts = s.timesteps
    if xts is None:
        xts = w.sample_xts_from_x0(x0, T, generator=generator)
    zs = torch.zeros((T, w.in_channels, x0.shape[-2], x0.shape[-1]))
    t_to_idx = {int(v): k for k, v in enumerate(ts)}
    xt = x0
    for t in ts:
        idx = T - t_to_idx[int(t)] - 1
        xt = xts[idx + 1][None]
        eps_u = w.unet(xt, t, cond_uncond)
        if not src_is_empty:
            eps_c = w.unet(xt.expand(n_prompts, -1, -1, -1), t, cond_src)
            eps = cfg_combine(eps_u, eps_c, cfg)
        else:
            eps = eps_u
        eta_i = eta[idx] if isinstance(eta, (list, tuple)) else eta          # `eta=etas[idx]`, inversion_utils.py:124
        z, xtm1 = w.get_zs_from_xts(xt, xts[idx][None], eps, t, eta=eta_i, numerical_fix=numerical_fix)
        zs[idx] = z
        xts[idx] = xtm1
    zs[0] = torch.zeros_like(zs[0])
    return xt, zs, xts


def edit(w, xT, tstart, cond_tgt, cond_neg, cfg_scales, zs, eta=1.0, n_prompts=1,
         cutoff_points=None, fix_alpha=0.1):
    """inversion_reverse_process (inversion_utils.py:147-323). `tstart` LongTensor[P]."""
    s = w.model.scheduler
    T = s.num_inference_steps
    cfg, masks = segment_scales(n_prompts, xT.shape[1:], cfg_scales, cutoff_points, xT.dtype)
    xt = xT[tstart.max()].unsqueeze(0)
    Z = zs.shape[0]
    ts = s.timesteps[-Z:]
    t_to_idx = {int(v): k for k, v in enumerate(ts)}
    for it, t in enumerate(ts):
        idx = T - t_to_idx[int(t)] - (T - Z + 1)
        eps_u = w.unet(xt, t, cond_neg)
        eps_c = w.unet(xt.expand(n_prompts, -1, -1, -1), t, cond_tgt)
        eps = cfg_combine(eps_u, eps_c, cfg)
        eta_i = eta[idx] if isinstance(eta, (list, tuple)) else eta          # inversion_utils.py:302
        xt = w.reverse_step_with_custom_noise(eps, t, xt, variance_noise=zs[idx].unsqueeze(0), eta=eta_i)
        apply_fix = (tstart.max() - tstart) > it
        if apply_fix.any():
            af = (apply_fix * fix_alpha)[:, None, None, None]
            xt = (masks * (xt.expand(n_prompts, -1, -1, -1) * (1 - af)
                           + af * xT[tstart.max() - it - 1].expand(n_prompts, -1, -1, -1))
                  ).sum(axis=0).unsqueeze(0)
    return xt


def ddim_next_step(w, eps, t, sample):
    """next_step (ddim_inversion.py:10-20)."""
    s = w.model.scheduler
    t_cur, t_next = min(t - s.config.num_train_timesteps // s.num_inference_steps, 999), t
    a_t = s.alphas_cumprod[t_cur] if t_cur >= 0 else s.final_alpha_cumprod
    a_next = s.alphas_cumprod[t_next]
    x0_hat = (sample - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
    return a_next ** 0.5 * x0_hat + (1 - a_next) ** 0.5 * eps


def ddim_invert(w, w0, cond_src, cond_uncond, cfg_scale, num_inference_steps, skip):
    """ddim_inversion (ddim_inversion.py:44-56)."""
    s = w.model.scheduler
    latent = w0.clone()
    for i in range(num_inference_steps):
        if num_inference_steps - i <= skip:
            break
        t = s.timesteps[len(s.timesteps) - i - 1]
        eps_u = w.unet(latent, t, cond_uncond)
        eps_c = w.unet(latent, t, cond_src)
        latent = ddim_next_step(w, eps_u + cfg_scale * (eps_c - eps_u), t, latent)
    return latent


def ddim_sample(w, xt, cond_tgt, cond_uncond, guidance_scale, skip=0):
    """text2image_ldm_stable (ddim_inversion.py:59-84)."""
    s = w.model.scheduler
    for t in s.timesteps[skip:]:
        eps_u = w.unet(xt, t, cond_uncond)
        eps_c = w.unet(xt, t, cond_tgt)
        xt = s.step(eps_u + guidance_scale * (eps_c - eps_u), t, xt, eta=0).prev_sample
    return xt

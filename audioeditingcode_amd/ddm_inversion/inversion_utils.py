"""Edit-friendly DDPM inversion + edit, same call signatures as the reference's
code/ddm_inversion/inversion_utils.py (inversion_forward_process :8-144, inversion_reverse_process :147-323).

Two execution paths, both entirely on HIP kernels:
  * fast path (default): the device-resident hipGraph loops of editing.EditEngine;
  * hook path: when the caller asks for h-space / skip-connection taps or replacements, or prompts
    have unequal `tstart` (multi-prompt trajectory blend), the loop is driven step by step through the
    wrapper's own methods exactly like the reference does.
"""
from typing import Dict, List, Optional, Tuple, Union

import torch

from ..editing import Conditioning


def gaussian_blur(x, kernel_size=15, sigma=1.0):
    """torchvision.transforms.functional.gaussian_blur restated (reflect pad, separable normalised kernel);
    used only to soften multi-prompt segment masks (inversion_utils.py:49,197-198)."""
    half = (kernel_size - 1) * 0.5
    g = torch.linspace(-half, half, kernel_size)
    pdf = torch.exp(-0.5 * (g / sigma) ** 2)
    k1 = pdf / pdf.sum()
    k2 = (k1[:, None] * k1[None, :]).to(x.dtype)
    c = x.shape[-3]
    pad = kernel_size // 2
    xp = torch.nn.functional.pad(x, (pad, pad, pad, pad), mode="reflect")
    return torch.nn.functional.conv2d(xp, k2.expand(c, 1, kernel_size, kernel_size), groups=c)


def _segment_tensors(batch_size, shape, cfg_scales, cutoff_points, dtype, prompts=None):
    """cfg-scale and mask tensors of inversion_utils.py:29-51 / :177-200 (host side, once per call)."""
    cfg = torch.ones((batch_size, *shape), dtype=dtype)
    masks = torch.ones((batch_size, *shape), dtype=dtype)
    if batch_size > 1:
        if cutoff_points is None:
            cutoff_points = [i * 1 / batch_size for i in range(1, batch_size)]
        if len(cfg_scales) == 1:
            cfg_scales = list(cfg_scales) * batch_size
        elif len(cfg_scales) < batch_size:
            raise ValueError("Not enough target CFG scales")
        cuts = [0, *[int(x * cfg.shape[2]) for x in cutoff_points], cfg.shape[2]]
        for i, (start, end) in enumerate(zip(cuts[:-1], cuts[1:])):
            cfg[i, :, end:] = 0
            cfg[i, :, :start] = 0
            masks[i, :, end:] = 0
            masks[i, :, :start] = 0
            cfg[i] *= cfg_scales[i]
            if prompts is not None and prompts[i] == "":
                cfg[i] = 0
        cfg = gaussian_blur(cfg, kernel_size=15, sigma=1)
        masks = gaussian_blur(masks, kernel_size=15, sigma=1)
    else:
        cfg *= cfg_scales[0]
    return cfg, masks


def conditioning_from_text(model, triple):
    """Wrap encode_text's (hidden_states, class_labels, mask) triple for the loop engine."""
    hs, cl, mask = triple
    if model.kind == "audioldm2":
        return Conditioning(ehs0=hs, ehs1=cl, mask1=mask)
    if model.kind == "audioldm":
        return Conditioning(class_labels=cl)
    return Conditioning(ehs0=hs, mask0=mask)


def inversion_forward_process(model, x0: torch.Tensor, etas: Optional[float] = None, prog_bar: bool = False,
                              prompts: List[str] = [""], cfg_scales: List[float] = [3.5],
                              num_inference_steps: int = 50, cutoff_points: Optional[List[float]] = None,
                              numerical_fix: bool = False, extract_h_space: bool = False,
                              extract_skipconns: bool = False, duration: Optional[float] = None,
                              first_order: bool = False, schedule: str = "sequential",
                              timestep_group: int = 8) -> Tuple:
    if len(prompts) > 1 and extract_h_space:
        raise NotImplementedError("How do you split cfg_scales for hspace? TODO")
    if getattr(model, "kind", None) == "stable_audio":
        return _sa_forward(model, x0, prompts, cfg_scales, num_inference_steps, numerical_fix, duration, first_order,
                           schedule, timestep_group, extract_h_space or extract_skipconns)
    if extract_h_space or extract_skipconns:
        return _forward_with_taps(model, x0, etas, prompts, cfg_scales, num_inference_steps, cutoff_points,
                                  numerical_fix, extract_h_space, extract_skipconns)
    prepared = prepare_forward(model, x0, prompts, cfg_scales, num_inference_steps, cutoff_points)
    return run_forward(model, x0, prepared, etas, cfg_scales, numerical_fix, schedule, timestep_group)


def prepare_forward(model, x0, prompts, cfg_scales, num_inference_steps, cutoff_points=None):
    """Everything of the fast-path forward process that does not touch the loop engine: text conditioning, the per-prompt
    cfg tensor, the x_t draws (models.py:67-83).  Split from `run_forward` so that a clip pipeline can issue it on a side
    stream while the loop engine is still busy with the previous clip (pipeline.ClipPipeline); called back to back on one
    stream it is exactly the first half of inversion_forward_process."""
    has_src = len(prompts) > 1 or prompts[0] != ""
    cond_src, cfg_tensor = None, None
    P = len(prompts)
    if has_src:
        cond_src = conditioning_from_text(model, model.encode_text(prompts))
        if P > 1:
            cfg_tensor, _ = _segment_tensors(P, x0.shape[1:], cfg_scales, cutoff_points, x0.dtype, prompts)
    cond_unc = conditioning_from_text(model, model.encode_text([""], negative=True))
    xts0 = model.sample_xts_from_x0(x0, num_inference_steps=num_inference_steps).unsqueeze(1)
    return dict(cond_src=cond_src, cond_unc=cond_unc, cfg_tensor=cfg_tensor, xts0=xts0)


def run_forward(model, x0, prepared, etas, cfg_scales, numerical_fix, schedule="sequential", timestep_group=8):
    """The loop half of the fast-path forward process: EditEngine.invert + the NCHW views the callers expect."""
    sched = model.model.scheduler
    if type(etas) in [int, float]:
        etas = [etas] * sched.num_inference_steps
    # one eta, or the reference's per-step list (`eta=etas[idx]`, inversion_utils.py:124): the device loop reads a
    # coefficient row per step either way
    eta = 0.0 if etas is None else (float(etas[0]) if all(float(e) == float(etas[0]) for e in etas) else list(etas))
    ed = model.editor(x0.shape[-2], x0.shape[-1])
    zs, xts = ed.invert(x0, prepared["cond_src"], prepared["cond_unc"], cfg_scales, eta=eta, numerical_fix=numerical_fix,
                        xts=prepared["xts0"], cfg_tensor=prepared["cfg_tensor"], mode=schedule, group=timestep_group)
    zs_n = ed.to_nchw(zs)[:, 0]
    xts_n = ed.to_nchw(xts)[:, 0]
    xt = xts_n[1][None]
    return xt, zs_n, xts_n, [None] * len(zs_n)


def _forward_with_taps(model, x0, etas, prompts, cfg_scales, T, cutoff_points, numerical_fix, extract_h_space,
                       extract_skipconns):
    """Step-by-step variant that returns h-space / skip taps (inversion_utils.py:103-121)."""
    has_src = len(prompts) > 1 or prompts[0] != ""
    sched = model.model.scheduler
    if type(etas) in [int, float]:
        etas = [etas] * sched.num_inference_steps
    if has_src:
        hs, cl, mk = model.encode_text(prompts)
        cfg_t, _ = _segment_tensors(len(prompts), x0.shape[1:], cfg_scales, cutoff_points, x0.dtype, prompts)
        cfg_t = cfg_t.to(model.device)
    uhs, ucl, umk = model.encode_text([""], negative=True)
    timesteps = sched.timesteps
    xts = model.sample_xts_from_x0(x0, num_inference_steps=T)
    zs = torch.zeros(size=model.get_noise_shape(x0, T), device=model.device)
    t_to_idx = {int(v): k for k, v in enumerate(timesteps)}
    hspaces, skipconns = [], []
    xt = x0
    for t in timesteps:
        idx = T - t_to_idx[int(t)] - 1
        xt = xts[idx + 1][None]
        out, out_h, out_s = model.unet_forward(xt, timestep=t, encoder_hidden_states=uhs, class_labels=ucl,
                                               encoder_attention_mask=umk)
        if has_src:
            cout, cout_h, cout_s = model.unet_forward(xt.expand(len(prompts), -1, -1, -1), timestep=t,
                                                      encoder_hidden_states=hs, class_labels=cl,
                                                      encoder_attention_mask=mk)
            noise_pred = out.sample + (cfg_t * (cout.sample - out.sample.expand(len(prompts), -1, -1, -1))
                                       ).sum(axis=0).unsqueeze(0)
            noise_h = out_h + cfg_scales[0] * (cout_h - out_h)
            noise_s = {k: [out_s[k][j] + cfg_scales[0] * (cout_s[k][j] - out_s[k][j]) for j in range(len(out_s[k]))]
                       for k in out_s} if extract_skipconns else None
        else:
            noise_pred, noise_h, noise_s = out.sample, out_h, out_s
        hspaces.append(noise_h)
        if extract_skipconns:
            skipconns.append(noise_s)
        z, xtm1, _ = model.get_zs_from_xts(xt, xts[idx][None], noise_pred, t, eta=etas[idx],
                                           numerical_fix=numerical_fix)
        zs[idx] = z
        xts[idx] = xtm1
    zs[0] = torch.zeros_like(zs[0])
    hspaces = torch.concat(hspaces, axis=0)
    if extract_h_space:
        return xt, zs, xts, [None] * len(zs), hspaces
    return xt, zs, xts, [None] * len(zs), hspaces, skipconns


def inversion_reverse_process(model, xT: torch.Tensor, tstart: torch.Tensor, fix_alpha: float = 0.1,
                              etas: float = 0, prompts: List[str] = [""], neg_prompts: List[str] = [""],
                              cfg_scales: Optional[List[float]] = None, prog_bar: bool = False,
                              zs: Optional[torch.Tensor] = None, cutoff_points: Optional[List[float]] = None,
                              hspace_add: Optional[torch.Tensor] = None,
                              hspace_replace: Optional[torch.Tensor] = None,
                              skipconns_replace: Optional[Dict[int, torch.Tensor]] = None,
                              zero_out_resconns: Optional[Union[int, List]] = None, extract_h_space: bool = False,
                              extract_skipconns: bool = False, duration: Optional[float] = None,
                              first_order: bool = False, extra_info: Optional[List] = None) -> Tuple:
    batch_size = len(prompts)
    tstart = torch.as_tensor(tstart).reshape(-1).cpu()
    if getattr(model, "kind", None) == "stable_audio":
        return _sa_reverse(model, xT, tstart, prompts, neg_prompts, cfg_scales, zs, duration, first_order, extra_info,
                           any(v is not None for v in (hspace_add, hspace_replace, skipconns_replace,
                                                       zero_out_resconns)) or extract_h_space or extract_skipconns)
    hooks = any(v is not None for v in (hspace_add, hspace_replace, skipconns_replace, zero_out_resconns)) \
        or extract_h_space or extract_skipconns
    uneven = bool((tstart.max() - tstart).any())
    sched = model.model.scheduler
    if etas is None:
        etas = 0
    if type(etas) in [int, float]:
        etas = [etas] * sched.num_inference_steps
    assert len(etas) == sched.num_inference_steps
    # the device-resident loop starts at xT[len(zs)]; a start index that differs from the number of noise maps takes the
    # step-by-step path, which follows the reference literally.  Per-step eta lists run on the device loop (a coefficient
    # row per step); the list is indexed like the reference's `etas[idx]`, idx = number of the noise map
    general = int(tstart.max()) != (zs.shape[0] if zs is not None else int(tstart.max()))
    if zs is not None:
        used = [float(e) for e in etas[:zs.shape[0]]]
        # a list that switches the noise term off at SOME steps: the reference skips `+ eta*sigma*z` there (models.py:152),
        # the fused kernel would multiply a possibly non-finite z by zero -- take the literal path
        general = general or (any(e == 0 for e in used) and any(e > 0 for e in used))
    if hooks or uneven or general:
        return _reverse_with_hooks(model, xT, tstart, fix_alpha, etas, prompts, neg_prompts, cfg_scales, zs,
                                   cutoff_points, hspace_add, hspace_replace, skipconns_replace, zero_out_resconns,
                                   extract_h_space, extract_skipconns)
    Zn = zs.shape[0]
    eta = float(etas[0]) if all(float(e) == float(etas[0]) for e in etas) else [float(e) for e in etas[:Zn]]
    cond_tgt = conditioning_from_text(model, model.encode_text(prompts))
    cond_neg = conditioning_from_text(model, model.encode_text(neg_prompts, negative=True))
    cfg_tensor = None
    if batch_size > 1:
        cfg_tensor, _ = _segment_tensors(batch_size, xT.shape[1:], cfg_scales, cutoff_points, xT.dtype)
    ed = model.editor(xT.shape[-2], xT.shape[-1])
    Z = zs.shape[0]
    xts_c = ed.to_nhwc(xT.unsqueeze(1))
    zs_c = ed.to_nhwc(zs.unsqueeze(1))
    w = ed.edit(xts_c, zs_c, Z, cond_tgt, cond_neg, cfg_scales, eta=eta, cfg_tensor=cfg_tensor)
    return ed.to_nchw(w), zs


def _reverse_with_hooks(model, xT, tstart, fix_alpha, etas, prompts, neg_prompts, cfg_scales, zs, cutoff_points,
                        hspace_add, hspace_replace, skipconns_replace, zero_out_resconns, extract_h_space,
                        extract_skipconns):
    """Step-by-step reverse process with the reference's hooks and multi-prompt trajectory blend
    (inversion_utils.py:221-315)."""
    batch_size = len(prompts)
    sched = model.model.scheduler
    hs, cl, mk = model.encode_text(prompts)
    uhs, ucl, umk = model.encode_text(neg_prompts, negative=True)
    cfg_t, masks = _segment_tensors(batch_size, xT.shape[1:], cfg_scales, cutoff_points, xT.dtype)
    cfg_t, masks = cfg_t.to(model.device), masks.to(model.device)
    xt = xT[tstart.max()].unsqueeze(0)
    Z = zs.shape[0]
    timesteps = sched.timesteps[-Z:]
    t_to_idx = {int(v): k for k, v in enumerate(timesteps)}
    hspaces, skipconns = [], []

    def pick(v, it, unsq=False):
        if v is None:
            return None
        if hasattr(v, "shape") and v.shape[0] > 1:
            r = v[-Z:][it]
            return r.unsqueeze(0) if unsq else r
        return v
    for it, t in enumerate(timesteps):
        idx = sched.num_inference_steps - t_to_idx[int(t)] - (sched.num_inference_steps - Z + 1)
        add = pick(hspace_add, it)
        kw = dict(replace_h_space=pick(hspace_replace, it, unsq=True), zero_out_resconns=zero_out_resconns,
                  replace_skip_conns=(None if skipconns_replace is None else
                                      (skipconns_replace[-Z:][it] if len(skipconns_replace) > 1
                                       else skipconns_replace)))
        uo, uo_h, uo_s = model.unet_forward(xt, timestep=t, encoder_hidden_states=uhs, class_labels=ucl,
                                            encoder_attention_mask=umk,
                                            mid_block_additional_residual=None if add is None else
                                            (1 / (cfg_scales[0] + 1)) * add, **kw)
        co, co_h, co_s = model.unet_forward(xt.expand(batch_size, -1, -1, -1), timestep=t, encoder_hidden_states=hs,
                                            class_labels=cl, encoder_attention_mask=mk,
                                            mid_block_additional_residual=None if add is None else
                                            (cfg_scales[0] / (cfg_scales[0] + 1)) * add, **kw)
        noise_pred = uo.sample + (cfg_t * (co.sample - uo.sample.expand(batch_size, -1, -1, -1))
                                  ).sum(axis=0).unsqueeze(0)
        if extract_h_space or extract_skipconns:
            hspaces.append(uo_h + cfg_scales[0] * (co_h - uo_h))
        if extract_skipconns:
            skipconns.append({k: [uo_s[k][j] + cfg_scales[0] * (co_s[k][j] - uo_s[k][j])
                                  for j in range(len(uo_s[k]))] for k in uo_s})
        z = zs[idx].unsqueeze(0)
        xt = model.reverse_step_with_custom_noise(noise_pred, t, xt, variance_noise=z, eta=etas[idx])
        apply_fix = ((tstart.max() - tstart) > it)
        if apply_fix.any():
            af = (apply_fix * fix_alpha).unsqueeze(1).unsqueeze(2).unsqueeze(3).to(xT.device)
            xt = (masks * (xt.expand(batch_size, -1, -1, -1) * (1 - af)
                           + af * xT[tstart.max() - it - 1].expand(batch_size, -1, -1, -1))).sum(axis=0).unsqueeze(0)
    if extract_h_space:
        return xt, zs, torch.concat(hspaces, axis=0)
    if extract_skipconns:
        return xt, zs, torch.concat(hspaces, axis=0), skipconns
    return xt, zs


# ------------------------------------------------------------------------------------------------ Stable Audio Open
def _sa_single_prompt(prompts, what):
    if len(prompts) != 1:
        # the reference's DiT call concatenates a batch-1 global token to the batch-P sequence (models.py:1345-1349 into
        # StableAudioDiTModel.forward): more than one prompt fails there as well
        raise NotImplementedError(f"Stable Audio: one {what} prompt per call (got {len(prompts)})")


def _sa_forward(model, x0, prompts, cfg_scales, T, numerical_fix, duration, first_order, schedule, group, taps):
    """inversion_forward_process for StableAudWrapper (inversion_utils.py:52-144 on the 3-D latent): returns
    (xt, zs [T,C,L], xts [T+1,C,L], extra_info) with extra_info[idx] = the data prediction of the previous step
    ([1,C,L]; None for idx = T-1), which inversion_reverse_process needs to re-seed the second-order solver."""
    if taps:
        raise NotImplementedError("Stable Audio has no h-space / skip-connection taps (models.py:1354 returns None)")
    _sa_single_prompt(prompts, "source")
    sched = model.model.scheduler
    has_src = prompts[0] != ""
    if has_src:
        hs, _, mask = model.encode_text(prompts)
    uhs, _, umask = model.encode_text([""], negative=True)
    model.setup_extra_inputs(x0, init_timestep=sched.timesteps[0], audio_end_in_s=duration)
    ctx_src = model.assemble_context(hs, mask) if has_src else None
    ctx_unc = model.assemble_context(uhs, umask)
    ed = model.editor()
    xts0 = model.sample_xts_from_x0(x0, num_inference_steps=T).unsqueeze(1)
    zs, xts, extra = ed.invert(x0.reshape(1, *x0.shape[-2:]), ctx_src, ctx_unc, model.audio_duration_embeds,
                               float(cfg_scales[0]), numerical_fix=numerical_fix, first_order=first_order, xts=xts0,
                               mode=schedule, group=group)
    zs_n, xts_n, extra_n = ed.to_cl(zs), ed.to_cl(xts), ed.to_cl(extra)
    extra_info = [extra_n[i][None] for i in range(T - 1)] + [None]
    return xts_n[1][None], zs_n, xts_n, extra_info


def _sa_reverse(model, xT, tstart, prompts, neg_prompts, cfg_scales, zs, duration, first_order, extra_info, hooks):
    """inversion_reverse_process for StableAudWrapper (inversion_utils.py:200-316 on the 3-D latent)."""
    if hooks:
        raise NotImplementedError("Stable Audio has no h-space / skip-connection hooks")
    _sa_single_prompt(prompts, "target")
    _sa_single_prompt(neg_prompts, "negative")
    sched = model.model.scheduler
    Z = zs.shape[0]
    if int(tstart.max()) != Z:
        raise NotImplementedError("Stable Audio: the edit starts at x_{tstart} with exactly tstart noise maps")
    hs, _, mask = model.encode_text(prompts)
    uhs, _, umask = model.encode_text(neg_prompts, negative=True)
    model.setup_extra_inputs(xT[Z].unsqueeze(0), extra_info=extra_info, init_timestep=sched.timesteps[-Z],
                             audio_end_in_s=duration)
    ctx_tgt = model.assemble_context(hs, mask)
    ctx_neg = model.assemble_context(uhs, umask)
    ed = model.editor()
    m1 = None
    if extra_info is not None and extra_info[Z - 1] is not None:
        m1 = ed.to_lc(extra_info[Z - 1].reshape(*xT.shape[-2:]))
    w = ed.edit(ed.to_lc(xT), ed.to_lc(zs), Z, ctx_tgt, ctx_neg, model.audio_duration_embeds, float(cfg_scales[0]),
                first_order=first_order, m1=m1)
    return ed.to_cl(w)[None], zs

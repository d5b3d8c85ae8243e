"""DDIM baseline (`--mode ddim`), same call signatures as the reference's
code/ddm_inversion/ddim_inversion.py (ddim_inversion :44-56, text2image_ldm_stable :59-84), executed by the
device-resident loops of editing.EditEngine (deterministic: consumes no RNG)."""
from typing import List, Optional

import torch

from .inversion_utils import conditioning_from_text


@torch.no_grad()
def ddim_inversion(ldm_model, w0, prompts, cfg_scale, num_inference_steps, skip):
    ed = ldm_model.editor(w0.shape[-2], w0.shape[-1])
    src = conditioning_from_text(ldm_model, ldm_model.encode_text(prompts))
    unc = conditioning_from_text(ldm_model, ldm_model.encode_text([""]))
    wT = ed.ddim_invert(w0, src, unc, cfg_scale, skip=skip)
    return ed.to_nchw(wT)


@torch.no_grad()
def text2image_ldm_stable(ldm_model, prompt: List[str], num_inference_steps: int = 50, guidance_scale: float = 7.5,
                          xt: Optional[torch.Tensor] = None, skip: int = 0):
    ed = ldm_model.editor(xt.shape[-2], xt.shape[-1])
    tgt = conditioning_from_text(ldm_model, ldm_model.encode_text(prompt))
    unc = conditioning_from_text(ldm_model, ldm_model.encode_text([""]))
    out = ed.ddim_sample(ed.to_nhwc(xt), tgt, unc, guidance_scale, skip=skip)
    return ed.to_nchw(out)

"""Apply extracted principal components to a clip: the body of the reference's
code/main_pc_apply_drift.py:68-199 without wandb / torchaudio -- replay the recorded trajectory with
`forward_directional` and, inside the drift window, shift x0 along the stored PCs with `apply_drift`
(one sample per requested PC, or all of them combined), with the optional "fix_alpha" blend towards the
un-drifted parallel trajectory outside the extraction patch.  Reads the `.pt` layout written by
main_pc_extract_inv (this package's or the reference's).
"""
import argparse
import os
import time
from types import SimpleNamespace
from typing import List, Optional

import torch


def _default_fns():
    from . import pc_drift
    from .utils import get_text_embeddings
    return SimpleNamespace(forward_directional=pc_drift.forward_directional, apply_drift=pc_drift.apply_drift,
                           PCStreamChoice=pc_drift.PCStreamChoice, get_text_embeddings=get_text_embeddings)


def drift_mask(latent0: torch.Tensor, patch, fade_length: int) -> torch.Tensor:
    """main_pc_apply_drift.py:110-121: 1 inside the extraction patch, linear fades of `fade_length` rows beside it."""
    mask = torch.zeros_like(latent0)
    if patch is not None:
        mask[:, :, patch[0]:patch[1], :] = 1
        if fade_length > 0:
            mask[:, :, patch[0] - fade_length:patch[0], :] = \
                torch.linspace(0, 1, fade_length, device=latent0.device)[None, None, :, None]
            mask[:, :, patch[1]:patch[1] + fade_length, :] = \
                torch.linspace(1, 0, fade_length, device=latent0.device)[None, None, :, None]
    else:
        mask[:, :, :, :] = 1
    return mask


def apply_pcs(ldm_stable, load_dict: dict, args, device, fns=None) -> torch.Tensor:
    """main_pc_apply_drift.py:71-199.  `args`: drift_start, drift_end, amount, evs, combine_evs,
    use_specific_ts_pc, sub_iters, shift_x0_for_np, fix_alpha, fade_length, rand_v, evals_pt (dict or None).
    Returns the final latents [len(evs) or 1, C, H, W]."""
    fns = fns or _default_fns()
    ex = load_dict["args"]
    eigdata = load_dict["eigdata"]
    if args.rand_v:
        for k in eigdata:
            norm = eigdata[k]["eigvec"].norm()
            eigdata[k]["eigvec"] = torch.randn_like(eigdata[k]["eigvec"])
            eigdata[k]["eigvec"] = eigdata[k]["eigvec"] / eigdata[k]["eigvec"].norm() * norm
    latents = [x.to(device) for x in load_dict["latents"]]
    xts = None
    if args.fix_alpha is not None:
        xts = load_dict.get("xts", None)
    fade = int(args.fade_length * latents[0].shape[2] / (ex.length if hasattr(ex, "length") else 15))
    timesteps = ldm_stable.model.scheduler.timesteps
    # the reference reads extraction_args.target_prompt, which its own extractor never sets (it records
    # source_prompt): accept either
    prompt = getattr(ex, "target_prompt", None) or ex.source_prompt
    _, text_emb, uncond_emb = fns.get_text_embeddings(prompt, ex.target_neg_prompt, ldm_stable)
    mask = drift_mask(latents[0], ex.patch, fade) if args.fix_alpha is not None else None
    drift_start_it = ex.num_diffusion_steps - args.drift_start
    drift_end_it = ex.num_diffusion_steps - args.drift_end

    xt = latents[0]
    parallel_xt = None
    if args.fix_alpha is not None:
        parallel_xt = xts[0].to(device) if xts is not None else latents[0]
    for it, t in enumerate(timesteps):
        xt_m1, x0_pred = fns.forward_directional(ldm_stable, xt, t, latents[it + 1], uncond_emb, text_emb, ex.cfg_tar,
                                                 eta=ex.eta, double_precision=ex.double_precision)
        if args.fix_alpha is not None:
            if xts is not None:
                parallel_xt = xts[it + 1].to(device)
            else:
                parallel_xt, _ = fns.forward_directional(ldm_stable, parallel_xt, t, latents[it + 1], uncond_emb,
                                                         text_emb, ex.cfg_tar, eta=ex.eta,
                                                         double_precision=ex.double_precision)
        if drift_start_it <= it < drift_end_it:
            kw = dict(use_shifted_x0_for_noisepred=args.shift_x0_for_np, use_specific_ts_pc=args.use_specific_ts_pc,
                      amount=args.amount, sub_iters=args.sub_iters, eta=ex.eta, evals=args.evals_pt)
            if args.combine_evs:
                xt_m1 = fns.apply_drift(ldm_stable, xt_m1, x0_pred, t, timesteps, ex.num_diffusion_steps, eigdata,
                                        latents[it + 1], device, ev_nums=args.evs, **kw)
            else:
                per_ev = []
                for ev_idx, ev_num in enumerate(args.evs):
                    # Reference quirk kept (main_pc_apply_drift.py:174): while the batch is still 1 the script passes
                    # xt_m1 -- not x0_pred -- as apply_drift's `x0_pred`.  apply_drift is affine in that argument and it
                    # cancels (x_{t-1} + (sqrt(a_prev) - c*k) * shift), so this only matters at rounding level; the
                    # reference's operand is used so that rounding matches too.
                    x0_arg = x0_pred[ev_idx].unsqueeze(0) if len(x0_pred) > 1 else xt_m1
                    per_ev.append(fns.apply_drift(
                        ldm_stable, xt_m1[ev_idx].unsqueeze(0) if len(xt_m1) > 1 else xt_m1, x0_arg, t, timesteps,
                        ex.num_diffusion_steps, eigdata, latents[it + 1], device, ev_nums=[ev_num], **kw))
                xt_m1 = torch.cat(per_ev, dim=0).to(device)
            if args.fix_alpha is not None:
                xt_m1 = mask * xt_m1 + (1 - mask) * (args.fix_alpha * parallel_xt + (1 - args.fix_alpha) * xt_m1)
        xt = xt_m1
    return xt


def output_name(args, ex, ev=None) -> str:
    """Output file stem of main_pc_apply_drift.py:201-232."""
    head = f'pcs{"".join(str(x) for x in args.evs)}_' if ev is None else f"pc{ev}_"
    return (head + f"drift{args.drift_start}-{args.drift_end}"
            f'{"_spts" + str(args.use_specific_ts_pc) if args.use_specific_ts_pc is not None else ""}'
            f"_it{ex.iters if args.sub_iters is None else args.sub_iters}_shiftednp{args.shift_x0_for_np}"
            f'{"_fade" + str(args.fade_length) if args.fade_length > 0 else ""}'
            f'{f"_fix{args.fix_alpha}" if args.fix_alpha is not None else ""}'
            f'{"_avgeval" if args.evals_pt is not None else ""}{"_RAND" if args.rand_v else ""}_a{args.amount}')


def build_parser():
    p = argparse.ArgumentParser("Apply extracted PCs to audio")
    p.add_argument("--device_num", type=int, default=0)
    p.add_argument("-s", "--seed", type=int, default=None)
    p.add_argument("--extraction_path", type=str, required=True)
    p.add_argument("--drift_start", type=int, required=True)
    p.add_argument("--drift_end", type=int, required=True)
    p.add_argument("--amount", type=float, required=True)
    p.add_argument("--use_specific_ts_pc", type=int, default=None)
    p.add_argument("--fix_alpha", type=float, default=None)
    p.add_argument("--fade_length", type=float, default=0.0)
    p.add_argument("--evs", type=int, nargs="+", default=[1])
    p.add_argument("--combine_evs", action="store_true")
    p.add_argument("--evals_pt", type=str, default=None)
    p.add_argument("--rand_v", action="store_true")
    p.add_argument("--allow_synthetic", action="store_true",
                   help="run with seeded-random weights / stand-in text embeddings when no checkpoint is on disk "
                        "(benchmarking only: the output is noise)")
    return p


def main(argv: Optional[List[str]] = None):
    from .models import load_model
    from .utils import set_reproducability, write_wav
    args = build_parser().parse_args(argv)
    args.shift_x0_for_np = True
    args.sub_iters = None
    if args.drift_start < args.drift_end:
        raise ValueError("Drift start must be greater than drift end")
    set_reproducability(args.seed, extreme=False)
    path = args.extraction_path[:-3] if args.extraction_path.endswith(".pt") else args.extraction_path
    device = f"cuda:{args.device_num}"
    torch.cuda.set_device(args.device_num)
    load_dict = torch.load(path + ".pt", map_location=device, weights_only=False)
    ex = load_dict["args"]
    if args.evals_pt is not None:
        args.evals_pt = torch.load(args.evals_pt, weights_only=False)
    ldm_stable = load_model(ex.model_id, device, ex.num_diffusion_steps, ex.double_precision,
                            allow_synthetic=getattr(args, "allow_synthetic", False) or None)
    print(f"weights: {ldm_stable.weights_source}; text conditioning: {ldm_stable.conditioning_source}")
    t0 = time.time()
    xt = apply_pcs(ldm_stable, load_dict, args, device)
    with torch.inference_mode():
        x0_dec = torch.cat([ldm_stable.vae_decode(xt[i].unsqueeze(0)) for i in range(len(xt))], dim=0)
        if x0_dec.dim() < 4:
            x0_dec = x0_dec[None]
        audio = ldm_stable.decode_to_mel(x0_dec)
    out_dir = path + "_driftgens"
    os.makedirs(out_dir, exist_ok=True)
    if args.combine_evs:
        write_wav(os.path.join(out_dir, output_name(args, ex) + ".wav"), audio[0].numpy())
    else:
        for i, ev in enumerate(args.evs):
            write_wav(os.path.join(out_dir, output_name(args, ex, ev) + ".wav"), audio[i].numpy())
    print(f"applied PCs {args.evs} in {time.time() - t0:.1f} s -> {out_dir}")


if __name__ == "__main__":
    main()

"""Unsupervised PC extraction for a real clip (BASELINE config 4): the body of the reference's
code/main_pc_extract_inv.py:95-262 without wandb / plotting -- invert the clip with the edit-friendly DDPM
inversion, replay the recorded noise maps with `forward_directional`, and inside the drift window run the
subspace iteration (`get_eigenvectors`) at every timestep.  The checkpoint written by `save_extraction` has the
reference's `.pt` layout (keys eigdata / args / corrs / in_corrs / latents / in_norms / xts; per timestep
eigvec / eigval / interm_eigvecs / interm_eigvals / it / ts / norm_factor), so main_pc_apply_drift of either
code base can read it.

The loop is host orchestration over the wrapper API; every U-Net evaluation inside the injected functions runs on
the native tape.  `fns` (default: this package's pc_drift + ddm_inversion) is injectable so the control flow is
unit-tested on CPU with a stub model (tests/test_host_cpu.py).
"""
import argparse
import os
import time
from types import SimpleNamespace
from typing import List, Optional

import torch


def _default_fns():
    from . import pc_drift
    from .ddm_inversion.inversion_utils import inversion_forward_process
    from .utils import get_text_embeddings
    return SimpleNamespace(forward_directional=pc_drift.forward_directional, get_eigenvectors=pc_drift.get_eigenvectors,
                           PCStreamChoice=pc_drift.PCStreamChoice, inversion_forward_process=inversion_forward_process,
                           get_text_embeddings=get_text_embeddings)


def extract_pcs(ldm_stable, w0: torch.Tensor, args, fns=None, checkpoint_cb=None):
    """main_pc_extract_inv.py:100-256.  `args` carries the reference's argparse fields (source_prompt,
    target_neg_prompt, cfg_tar, num_diffusion_steps, drift_start, drift_end, const, n_evs, iters, patch,
    corr_to_swap, dry, eta, numerical_fix, double_precision, pc_mode).  Returns the checkpoint dict."""
    fns = fns or _default_fns()
    timesteps = ldm_stable.model.scheduler.timesteps
    if args.drift_start is None:
        args.drift_start = args.num_diffusion_steps
    if args.drift_end is None:
        args.drift_end = -1
    drift_start_it = args.num_diffusion_steps - args.drift_start
    drift_end_it = args.num_diffusion_steps - args.drift_end

    _, text_emb, uncond_emb = fns.get_text_embeddings(args.source_prompt, args.target_neg_prompt, ldm_stable)
    _, zs, wts, _ = fns.inversion_forward_process(ldm_stable, w0, etas=args.eta, prompts=args.source_prompt,
                                                  cfg_scales=[args.cfg_tar], prog_bar=False,
                                                  num_inference_steps=args.num_diffusion_steps,
                                                  numerical_fix=args.numerical_fix)
    wts = wts.flip(0)
    latents = [wts[0].unsqueeze(0), *[z.unsqueeze(0) for z in zs.flip(0)]]
    del wts, zs

    mask = torch.zeros_like(latents[0])
    if args.patch is not None:
        mask[:, :, args.patch[0]:args.patch[1], :] = 1
    else:
        mask[:, :, :, :] = 1
    pc_mode = {"text": fns.PCStreamChoice.TEXT, "uncond": fns.PCStreamChoice.UNCOND}.get(args.pc_mode,
                                                                                           fns.PCStreamChoice.BOTH)
    xt = latents[0]
    prev_pc = None
    corrs, in_corrs, in_norms = [], [], []
    xts = [xt.detach().clone()]
    eigdata = {}

    def state():
        return {"eigdata": eigdata, "args": args, "corrs": corrs, "in_corrs": in_corrs, "latents": latents,
                "in_norms": in_norms, "xts": xts}

    for it, t in enumerate(timesteps):
        xt_m1, x0_pred = fns.forward_directional(ldm_stable, xt, t, latents[it + 1], uncond_emb, text_emb, args.cfg_tar,
                                                 eta=args.eta, double_precision=args.double_precision)
        if not args.dry and drift_start_it <= it < drift_end_it:
            eigvecs, eigval, in_corr, in_norm, interm_eigvecs, interm_eigvals = fns.get_eigenvectors(
                ldm_stable, xt, text_emb, uncond_emb, latents[it + 1], mask, t, x0_pred, pc_mode, args.const,
                args.cfg_tar, args.iters, args.double_precision, args.eta, args.n_evs)
            if it > drift_start_it:
                # keep the sign of every PC consistent along the trajectory (main_pc_extract_inv.py:204-212)
                corr = (prev_pc.reshape(args.n_evs, -1) @ eigvecs.reshape(args.n_evs, -1).T).diag()
                for ev_num in range(args.n_evs):
                    if corr[ev_num] <= -args.corr_to_swap:
                        eigvecs[ev_num] *= -1
                        corr[ev_num] *= -1
                corrs.append(corr)
            prev_pc = eigvecs
            in_corrs.append(in_corr)
            in_norms.append(in_norm)
            eigdata[t.item()] = {
                "eigvec": eigvecs.detach().cpu(),
                "eigval": eigval.detach().cpu(),
                "interm_eigvecs": {k: v.detach().cpu() for k, v in interm_eigvecs.items()},
                "interm_eigvals": {k: v.detach().cpu() for k, v in interm_eigvals.items()},
                "it": it,
                "ts": args.num_diffusion_steps - it,
                "norm_factor": torch.sqrt(ldm_stable.model.scheduler.alphas_cumprod[t])}
        xt = xt_m1
        xts.append(xt.detach().clone())
        if checkpoint_cb is not None and it % 10 == 0:
            checkpoint_cb(state())
    out = state()
    out["final"] = xt
    return out


def extraction_name(args, time_stamp=None) -> str:
    """File stem of main_pc_extract_inv.py:72-78 (`time_stamp` = calendar.timegm(time.gmtime()) there)."""
    return (f"s{args.seed}_" + (f"p{args.patch[0]}-{args.patch[1]}_" if args.patch is not None else "") +
            f"pc-{args.pc_mode}_cfgd{args.cfg_tar}_drift{args.drift_start}-{args.drift_end}_it{args.iters}"
            f"_c{args.const:.1e}" + ("_dp" if args.double_precision else "") +
            (f"_{time_stamp}" if time_stamp is not None else ""))


def save_extraction(ckpt: dict, path: str):
    d = {k: ckpt[k] for k in ("eigdata", "args", "corrs", "in_corrs", "latents", "in_norms", "xts")}
    torch.save(d, path if path.endswith(".pt") else path + ".pt")


def build_parser():
    p = argparse.ArgumentParser(description="Extract PCs for a real audio signal")
    p.add_argument("--device_num", type=int, default=0)
    p.add_argument("-s", "--seed", type=int, default=None)
    p.add_argument("--cfg_tar", type=float, default=3)
    p.add_argument("--model_id", type=str, default="cvssp/audioldm2-music")
    p.add_argument("--init_aud", type=str, default=None, help="wav to invert (default: the synthetic benchmark clip)")
    p.add_argument("--num_diffusion_steps", type=int, default=200)
    p.add_argument("--source_prompt", type=str, nargs="+", default=[""])
    p.add_argument("--target_neg_prompt", type=str, nargs="+", default=[""])
    p.add_argument("--corr_to_swap", type=float, default=0.8)
    p.add_argument("--drift_start", type=int, default=None)
    p.add_argument("--drift_end", type=int, default=None)
    p.add_argument("--results_path", default="pc_extractions")
    p.add_argument("-c", "--const", type=float, default=1e-3)
    p.add_argument("--n_evs", type=int, default=1)
    p.add_argument("-p", "--patch", nargs=2, default=None, type=int)
    p.add_argument("-t", "--iters", type=int, default=50)
    p.add_argument("-d", "--dry", action="store_true")
    p.add_argument("--allow_synthetic", action="store_true",
                   help="run with seeded-random weights / stand-in text embeddings when no checkpoint is on disk "
                        "(benchmarking only: the output is noise)")
    return p


def finish_args(args):
    """The fields the reference sets after parsing (main_pc_extract_inv.py:63-68)."""
    args.pc_mode = "both"
    args.eta = 1.0
    args.numerical_fix = True
    args.double_precision = False
    args.test_rand_gen = False
    return args


def main(argv: Optional[List[str]] = None):
    from .models import load_model
    from .utils import load_audio, set_reproducability, synthetic_clip, write_wav
    args = finish_args(build_parser().parse_args(argv))
    set_reproducability(args.seed, extreme=False)
    device = f"cuda:{args.device_num}"
    torch.cuda.set_device(args.device_num)
    ldm_stable = load_model(args.model_id, device, args.num_diffusion_steps, args.double_precision,
                            allow_synthetic=getattr(args, "allow_synthetic", False) or None)
    print(f"weights: {ldm_stable.weights_source}; text conditioning: {ldm_stable.conditioning_source}")
    src = args.init_aud if args.init_aud else (synthetic_clip(), 16000)
    # (the reference's call site is stale here -- no stft=True, 3-tuple bound to x0, main_pc_extract_inv.py:110; SURVEY
    # quirk 10 -- the evident intent is implemented)
    x0, _, _ = load_audio(src, ldm_stable.get_fn_STFT(), device=device, stft=True, model_sr=ldm_stable.get_sr())
    with torch.inference_mode():
        w0 = ldm_stable.vae_encode(x0)
    clip = os.path.basename(args.init_aud).split(".")[0] if args.init_aud else "synthetic"
    save_path = os.path.join(args.results_path, args.model_id.split("/")[-1], clip,
                             "pmt_" + "__".join(x.replace(" ", "_") for x in args.source_prompt) + "__neg__" +
                             "__".join(x.replace(" ", "_") for x in args.target_neg_prompt))
    os.makedirs(save_path, exist_ok=True)
    stem = os.path.join(save_path, extraction_name(args))
    t0 = time.time()
    ckpt = extract_pcs(ldm_stable, w0, args, checkpoint_cb=lambda st: save_extraction(st, stem))
    save_extraction(ckpt, stem)
    with torch.inference_mode():
        x0_dec = ldm_stable.vae_decode(ckpt["final"])
        if x0_dec.dim() < 4:
            x0_dec = x0_dec[None]
        audio = ldm_stable.decode_to_mel(x0_dec)
    write_wav(stem + ".wav", audio[0].numpy())
    print(f"extracted {len(ckpt['eigdata'])} timesteps x {args.n_evs} PCs in {time.time() - t0:.1f} s -> {stem}.pt")


if __name__ == "__main__":
    main()

"""Text-based edit CLI: the call pattern of the reference's code/main_run.py (:113-185), on the native path.

python -m audioeditingcode_amd.main_run --init_aud clip.wav --source_prompt "..." --target_prompt "..." \
       --model_id cvssp/audioldm2-music --tstart 100 --num_diffusion_steps 200
Without --init_aud a synthetic 10 s clip is edited (no dataset exists in this container)."""
import argparse
import os
import time
import warnings

import torch

from .ddm_inversion.ddim_inversion import ddim_inversion, text2image_ldm_stable
from .ddm_inversion.inversion_utils import inversion_forward_process, inversion_reverse_process
from .models import load_model
from .utils import load_audio, set_reproducability, synthetic_clip, write_wav


def edit_clip(ldm_stable, x0, source_prompt, target_prompt, target_neg_prompt, cfg_src, cfg_tar, T, tstart,
              mode="ours", eta=1.0, schedule="sequential", timestep_group=8, cutoff_points=None, fix_alpha=0.1,
              duration=None):
    """main_run.py:104-185 for one mel `x0` [1,1,T_mel,64] (Stable Audio: one waveform [channels, n]): returns (edited
    waveform, original-vocoded waveform, edited latent).  `tstart`: int or one value per target prompt
    (main_run.py:104-110)."""
    tstart = [int(tstart)] if isinstance(tstart, (int, float)) else [int(t) for t in tstart]
    if len(tstart) != len(target_prompt):
        if len(tstart) == 1:
            tstart = tstart * len(target_prompt)
        else:
            raise ValueError("T-start amount and target prompt amount don't match.")
    tstart_t = torch.tensor(tstart, dtype=torch.int)
    skip = T - tstart_t
    with torch.inference_mode():
        w0 = ldm_stable.vae_encode(x0)
        if mode == "ddim":
            if len(cfg_src) > 1:
                raise ValueError("DDIM only supports one cfg_scale_src value")
            if len(cfg_tar) > 1:
                raise ValueError("DDIM only supports one cfg_scale_tar value")
            if len(source_prompt) > 1:
                raise ValueError("DDIM only supports one args.source_prompt value")
            if len(target_prompt) > 1:
                raise ValueError("DDIM only supports one args.target_prompt value")
            if int(skip[0]) != 0:
                warnings.warn("Plain DDIM Inversion should be run with t_start == num_diffusion_steps. "
                              "You are now running partial DDIM inversion.", RuntimeWarning)
            wT = ddim_inversion(ldm_stable, w0, source_prompt, cfg_src[0], num_inference_steps=T, skip=int(skip[0]))
            w_edit = text2image_ldm_stable(ldm_stable, target_prompt, T, cfg_tar[0], wT, skip=int(skip[0]))
        else:
            _, zs, wts, extra_info = inversion_forward_process(
                ldm_stable, w0, etas=eta, prompts=source_prompt, cfg_scales=cfg_src, num_inference_steps=T,
                numerical_fix=True, cutoff_points=cutoff_points, schedule=schedule, timestep_group=timestep_group,
                duration=duration)
            w_edit, _ = inversion_reverse_process(ldm_stable, xT=wts, tstart=tstart_t, fix_alpha=fix_alpha, etas=eta,
                                                  prompts=target_prompt, neg_prompts=target_neg_prompt,
                                                  cfg_scales=cfg_tar, zs=zs[:int(T - min(skip))],
                                                  cutoff_points=cutoff_points, duration=duration,
                                                  extra_info=extra_info)
        x0_dec = ldm_stable.vae_decode(w_edit)
        if "stable-audio" in ldm_stable.model_id:                       # main_run.py:187-190: the VAE output IS the audio
            return x0_dec.detach().clone().cpu().squeeze(0), x0.detach().clone().cpu(), w_edit
        if x0_dec.dim() < 4:
            x0_dec = x0_dec[None, :, :, :]
        audio = ldm_stable.decode_to_mel(x0_dec)
        orig_audio = ldm_stable.decode_to_mel(x0)
    return audio, orig_audio, w_edit


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--device_num", type=int, default=0)
    p.add_argument("-s", "--seed", type=int, default=None)
    p.add_argument("--model_id", type=str, default="cvssp/audioldm2-music")
    p.add_argument("--init_aud", type=str, default=None)
    p.add_argument("--cfg_src", type=float, nargs="+", default=[3])
    p.add_argument("--cfg_tar", type=float, nargs="+", default=[12])
    p.add_argument("--num_diffusion_steps", type=int, default=200)
    p.add_argument("--target_prompt", type=str, nargs="+", default=[""])
    p.add_argument("--source_prompt", type=str, nargs="*", default=[""])
    p.add_argument("--target_neg_prompt", type=str, nargs="*", default=[""])
    p.add_argument("--tstart", type=int, nargs="+", default=[100])
    p.add_argument("--results_path", default="results")
    p.add_argument("--mode", default="ours", choices=["ours", "ddim"])
    p.add_argument("--cutoff_points", type=float, nargs="*", default=None)
    p.add_argument("--fix_alpha", type=float, default=0.1)
    p.add_argument("--schedule", default="sequential", choices=["sequential", "batched"])
    p.add_argument("--allow_synthetic", action="store_true",
                   help="run with seeded-random weights / stand-in text embeddings when no checkpoint is on disk "
                        "(benchmarking only: the output is noise)")
    args = p.parse_args(argv)
    args.eta = 1.0
    set_reproducability(args.seed, extreme=False)
    device = f"cuda:{args.device_num}"
    torch.cuda.set_device(args.device_num)
    ldm_stable = load_model(args.model_id, device, args.num_diffusion_steps, allow_synthetic=args.allow_synthetic or None)
    src = args.init_aud if args.init_aud else (synthetic_clip(), 16000)
    sa = "stable-audio" in args.model_id
    x0, sr, duration = load_audio(src, ldm_stable.get_fn_STFT(), device=device, stft=not sa, model_sr=ldm_stable.get_sr())
    t0 = time.time()
    audio, orig, _ = edit_clip(ldm_stable, x0, args.source_prompt, args.target_prompt, args.target_neg_prompt,
                               args.cfg_src, args.cfg_tar, args.num_diffusion_steps, args.tstart, args.mode,
                               args.eta, args.schedule, cutoff_points=args.cutoff_points, fix_alpha=args.fix_alpha,
                               duration=duration)
    torch.cuda.synchronize()
    print(f"edited {duration:.1f} s clip in {time.time() - t0:.2f} s (weights: {ldm_stable.weights_source}; "
          f"text conditioning: {ldm_stable.conditioning_source})")
    os.makedirs(args.results_path, exist_ok=True)
    # main_run.py:223-224 saves the whole [channels, n] tensor: mono for the mel families, stereo for Stable Audio
    audio, orig = (a if a.dim() == 2 else a.reshape(-1, a.shape[-1]) for a in (audio, orig))
    write_wav(os.path.join(args.results_path, "edited.wav"), audio.numpy(), sr=sr)
    write_wav(os.path.join(args.results_path, "orig.wav"), orig.numpy(), sr=sr)


if __name__ == "__main__":
    main()

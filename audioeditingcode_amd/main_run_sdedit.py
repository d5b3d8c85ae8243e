"""SDEdit baseline CLI: the reference's code/main_run_sdedit.py without wandb / plotting --
encode the clip, noise it to `tstart`, run the classifier-free-guided DDPM sampler down to 0 (sdedit.sdedit: the
device-resident reverse loop), decode, vocode."""
import argparse
import os
import time
from typing import List, Optional

import torch


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--device_num", type=int, default=0)
    p.add_argument("-s", "--seed", type=int, default=None)
    p.add_argument("--model_id", type=str, default="cvssp/audioldm2-music")
    p.add_argument("--init_aud", type=str, default=None, help="wav to edit (default: the synthetic benchmark clip)")
    p.add_argument("--cfg_tar", type=float, default=12)
    p.add_argument("--num_diffusion_steps", type=int, default=200)
    p.add_argument("--target_prompt", type=str, nargs="+", default=[""])
    p.add_argument("--target_neg_prompt", type=str, nargs="+", default=[""])
    p.add_argument("--results_path", default="sdedit")
    p.add_argument("--tstart", type=int, default=100)
    p.add_argument("--allow_synthetic", action="store_true",
                   help="run with seeded-random weights / stand-in text embeddings when no checkpoint is on disk "
                        "(benchmarking only: the output is noise)")
    return p


def main(argv: Optional[List[str]] = None):
    from .models import load_model
    from .sdedit import sdedit
    from .utils import load_audio, set_reproducability, synthetic_clip, write_wav
    args = build_parser().parse_args(argv)
    args.eta = 1.0
    set_reproducability(args.seed, extreme=False)
    skip = args.num_diffusion_steps - args.tstart
    name = f"s{args.seed}_skip{skip}_cfg{args.cfg_tar}"
    device = f"cuda:{args.device_num}"
    torch.cuda.set_device(args.device_num)
    ldm_stable = load_model(args.model_id, device, args.num_diffusion_steps,
                            allow_synthetic=getattr(args, "allow_synthetic", False) or None)
    print(f"weights: {ldm_stable.weights_source}; text conditioning: {ldm_stable.conditioning_source}")
    src = args.init_aud if args.init_aud else (synthetic_clip(), 16000)
    # (stale call site in the reference, main_run_sdedit.py:69 -- SURVEY quirk 10; the evident intent is implemented)
    x0, _, _ = load_audio(src, ldm_stable.get_fn_STFT(), device=device, stft=True, model_sr=ldm_stable.get_sr())
    t0 = time.time()
    with torch.inference_mode():
        w0 = ldm_stable.vae_encode(x0)
        xt = sdedit(ldm_stable, w0, args.target_prompt, args.target_neg_prompt, args.cfg_tar, skip, eta=args.eta)
        x0_dec = ldm_stable.vae_decode(xt)
        if x0_dec.dim() < 4:
            x0_dec = x0_dec[None]
        audio = ldm_stable.decode_to_mel(x0_dec)
        orig_audio = ldm_stable.decode_to_mel(x0)
    torch.cuda.synchronize()
    clip = os.path.basename(args.init_aud).split(".")[0] if args.init_aud else "synthetic"
    save_path = os.path.join(args.results_path, args.model_id.split("/")[-1], clip,
                             "pmt_" + "__".join(x.replace(" ", "_") for x in args.target_prompt) + "__neg__" +
                             "__".join(x.replace(" ", "_") for x in args.target_neg_prompt))
    os.makedirs(save_path, exist_ok=True)
    write_wav(os.path.join(save_path, name + ".wav"), audio[0].numpy())
    write_wav(os.path.join(save_path, "orig.wav"), orig_audio[0].numpy())
    print(f"SDEdit from t={args.tstart} in {time.time() - t0:.2f} s -> {save_path}")


if __name__ == "__main__":
    main()

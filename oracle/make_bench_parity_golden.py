"""TEST INFRASTRUCTURE: the oracle side of bench.py's parity leg at the BENCHED schedule (AudioLDM2 full size with the bench's
seeded-random weights, its synthetic clip #4242, its prompts, cfg 3 / 12, T=200, tstart=100, reference step order) as a
fixture: log-mel of the clip (oracle STFT), edited latent, decoded mel, vocoded waveform.  bench.py feeds the fixture's mel to
the HIP path with the same seed and reports `parity_T200` (latent / mel / waveform rel L2 vs this file) next to the live T=8
parity leg -- the metric of BASELINE.json is "clips/s + mel-L2 vs ref", and ~7 minutes of CPU oracle do not fit a bench run.

The conditioning is the wrapper's synthetic stand-in (no text-encoder checkpoints exist here): the same seeded CPU generators
(models._prompt_generator / _SyntheticText), so both sides see identical tensors.

    PYTHONPATH=. python oracle/make_bench_parity_golden.py       -> tests/golden/bench_parity_T200.npz (~7 min on 8 cores)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audioeditingcode_amd import configs, models, weights                               # noqa: E402
from audioeditingcode_amd.utils import synthetic_clip                                    # noqa: E402
from oracle import audio as oaudio, hifigan as ohifi, loops as oloops, unet as ounet, vae as ovae       # noqa: E402
from oracle.scheduler import OracleDDIMScheduler                                        # noqa: E402

T, TSTART, SEED, CLIP = 200, 100, 77, 4242
SRC, TGT, NEG = ["a recording of a piano melody"], ["a recording of an electric guitar melody"], [""]


def synthetic_conditioning(fam, prompts):
    """AudioLDM2Wrapper.encode_text without text encoders (models.py of this package), on the CPU."""
    c = fam["ctx"]
    gen = torch.stack([torch.randn(c["gpt2_len"], c["gpt2_dim"], generator=models._prompt_generator(p, "gpt2"))
                       for p in prompts])
    t5, mask = models._SyntheticText.t5(prompts, c["t5_dim"])
    return gen, t5, mask


def main():
    fam = configs.get_family("cvssp/audioldm2")
    shapes = dict(unet=weights.unet_param_shapes(fam["unet"]), vae=weights.vae_param_shapes(fam["vae"]),
                  vocoder=weights.vocoder_param_shapes(fam["vocoder"]))
    sds = {k: weights.random_state_dict(shapes[k], seed=i) for i, k in enumerate(("unet", "vae", "vocoder"))}     # bench.py's
    wave = oaudio.prepare_waveform(synthetic_clip(10.0, seed=1234 + CLIP), 1024 * 160)
    wave = torch.clip(torch.as_tensor(wave, dtype=torch.float32)[None], -1, 1)
    mel, _, _ = oaudio.mel_spectrogram(wave)                            # [1, 64, frames]
    x0 = mel[0].T[:1024][None, None].contiguous()
    cfg, sd = fam["unet"], sds["unet"]
    osched = OracleDDIMScheduler()
    osched.set_timesteps(T)

    def unet_fn(x, t, cond):
        hs, cl, mk = (v.expand(x.shape[0], *v.shape[1:]) for v in cond)
        return ounet.unet_forward(cfg, sd, x, t, encoder_hidden_states=hs, encoder_hidden_states_1=cl,
                                  encoder_attention_mask_1=mk)[0]
    t0 = time.time()
    with torch.no_grad():
        ow = oloops.OracleWrapper(osched, unet_fn)
        w0 = ovae.vae_encode(fam["vae"], sds["vae"], x0)
        xts0 = ow.sample_xts_from_x0(w0, T, generator=torch.Generator().manual_seed(SEED))
        enc = lambda p: synthetic_conditioning(fam, p)                  # noqa: E731
        _, zs, xts = oloops.invert(ow, w0, enc(SRC), enc([""]), [3.0], T, xts=xts0)
        print(f"inversion {time.time() - t0:.0f} s", flush=True)
        w_o = oloops.edit(ow, xts, torch.tensor([TSTART]), enc(TGT), enc(NEG), [12.0], zs[:TSTART], eta=1.0)
        mel_o = ovae.vae_decode(fam["vae"], sds["vae"], w_o)
        wav_o = ohifi.hifigan_forward(fam["vocoder"], sds["vocoder"], mel_o[:, 0])
    print(f"done {time.time() - t0:.0f} s", flush=True)
    out = os.path.join(ROOT, "tests", "golden", "bench_parity_T200.npz")
    np.savez_compressed(out, x0=x0.numpy(), w0=w0.numpy(), w_edit=w_o.numpy(), mel=mel_o.numpy(), wav=wav_o.numpy(),
                        T=np.array(T), tstart=np.array(TSTART), seed=np.array(SEED), clip=np.array(CLIP),
                        prompts=np.array([SRC[0], TGT[0], NEG[0]]))
    print("wrote", out, os.path.getsize(out), "bytes; latent", tuple(w_o.shape), "mel", tuple(mel_o.shape), "wav", tuple(wav_o.shape))


if __name__ == "__main__":
    main()

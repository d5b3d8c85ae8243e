"""GPU parity of the STFT/mel front end, the VAE and the HiFi-GAN vocoder tapes."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from audioeditingcode_amd import configs, weights                                     # noqa: E402
from audioeditingcode_amd.codec import STFTEngine, VAEDecoder, VAEEncoder, VocoderEngine  # noqa: E402
from oracle import audio as oaudio                                                     # noqa: E402
from oracle import hifigan as ohifi                                                    # noqa: E402
from oracle import vae as ovae                                                         # noqa: E402
from oracle.synth import chirp_waveform                                                # noqa: E402

DEV = "cuda:0"


def rel(a, b):
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def test_stft_mel_matches_reference_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "stft_mel_64f.npz"))
    wav = torch.from_numpy(g["wav"])[None]
    eng = STFTEngine(configs.STFT_AUDIOLDM, DEV, 1, wav.shape[1])
    mel = eng(wav)
    torch.cuda.synchronize()
    ref = torch.from_numpy(g["mel"])[0].T                     # reference layout [n_mels, frames] -> [frames, n_mels]
    assert mel.shape == (1, 65, 64)
    np.testing.assert_allclose(mel[0].cpu().numpy(), ref.numpy(), atol=2e-3, rtol=1e-4)
    np.testing.assert_allclose(eng.mag.cpu().numpy()[:, :513], g["mag"][0].T, atol=3e-4, rtol=1e-4)
    np.testing.assert_allclose(eng.mel_basis.numpy(), g["mel_basis"], atol=1e-7, rtol=1e-5)


def test_stft_full_clip_batch2_matches_oracle():
    wavs = torch.stack([torch.from_numpy(oaudio.prepare_waveform(chirp_waveform(seed=s).numpy(), 163840))
                        for s in (1234, 1235)])
    eng = STFTEngine(configs.STFT_AUDIOLDM, DEV, 2, 163840)
    mel = eng(wavs)
    torch.cuda.synchronize()
    ref, _, _ = oaudio.mel_spectrogram(wavs)
    assert mel.shape == (2, 1025, 64)
    assert (mel.cpu() - ref.transpose(1, 2)).abs().max() < 5e-3
    assert rel(mel.cpu(), ref.transpose(1, 2)) < 1e-5


def _vae(cfg, T, F, seed=0):
    sd = weights.random_state_dict(weights.vae_param_shapes(cfg), seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    mel = torch.randn(1, 1, T, F, generator=g) * 2 - 4
    enc = VAEEncoder(cfg, sd, DEV, 1, T, F)
    lat = enc(mel)
    torch.cuda.synchronize()
    ref_lat = ovae.vae_encode(cfg, sd, mel)
    got_lat = lat.cpu().permute(0, 3, 1, 2)
    dec = VAEDecoder(cfg, sd, DEV, 1, enc.h, enc.w)
    rec = dec(lat)
    torch.cuda.synchronize()
    ref_rec = ovae.vae_decode(cfg, sd, ref_lat)
    return got_lat, ref_lat, rec.cpu().permute(0, 3, 1, 2), ref_rec


def test_vae_tiny_matches_oracle():
    cfg = configs.tiny_family("audioldm2")["vae"]
    got_lat, ref_lat, rec, ref_rec = _vae(cfg, 64, 32)
    assert got_lat.shape == ref_lat.shape == (1, 8, 16, 8)
    assert rel(got_lat, ref_lat) < 1e-4 and rel(rec, ref_rec) < 2e-4


def test_vae_full_size_matches_oracle():
    cfg = configs.VAE_AUDIOLDM
    got_lat, ref_lat, rec, ref_rec = _vae(cfg, 1024, 64)
    assert got_lat.shape == (1, 8, 256, 16) and rec.shape == (1, 1, 1024, 64)
    assert rel(got_lat, ref_lat) < 1e-4, rel(got_lat, ref_lat)
    assert rel(rec, ref_rec) < 2e-4, rel(rec, ref_rec)


def test_vocoder_matches_transformers_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "hifigan_c64.npz"))
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}
    cfg = dict(configs.VOCODER_AUDIOLDM, upsample_initial_channel=64)
    mel = torch.from_numpy(g["mel"])
    eng = VocoderEngine(cfg, sd, DEV, mel.shape[0], mel.shape[1])
    wav = eng(mel)
    torch.cuda.synchronize()
    assert tuple(wav.shape) == g["wav"].shape
    np.testing.assert_allclose(wav.cpu().numpy(), g["wav"], atol=2e-6, rtol=1e-4)


def test_vocoder_full_size_matches_oracle():
    cfg = configs.VOCODER_AUDIOLDM
    sd = weights.random_state_dict(weights.vocoder_param_shapes(cfg), seed=0)
    g = torch.Generator().manual_seed(1)
    mel = torch.randn(1, 256, 64, generator=g) * 2 - 4          # 2.5 s: the oracle finishes in ~1 s
    eng = VocoderEngine(cfg, sd, DEV, 1, 256)
    wav = eng(mel)
    torch.cuda.synchronize()
    ref = ohifi.hifigan_forward(cfg, sd, mel)
    assert wav.shape == ref.shape == (1, 256 * 160 + 32)
    assert rel(wav.cpu(), ref) < 2e-4, rel(wav.cpu(), ref)


def test_vocoder_ten_second_clip_matches_oracle():
    """The benchmark clip length: 1024 mel frames -> 163 872 samples (main_run.py:184-185 calls the vocoder twice)."""
    cfg = configs.VOCODER_AUDIOLDM
    sd = weights.random_state_dict(weights.vocoder_param_shapes(cfg), seed=0)
    g = torch.Generator().manual_seed(2)
    mel = torch.randn(1, 1024, 64, generator=g) * 2 - 4
    eng = VocoderEngine(cfg, sd, DEV, 1, 1024)
    wav = eng(mel)
    torch.cuda.synchronize()
    ref = ohifi.hifigan_forward(cfg, sd, mel)
    assert wav.shape == ref.shape == (1, 1024 * 160 + 32)
    assert torch.isfinite(wav).all()
    assert rel(wav.cpu(), ref) < 2e-4, rel(wav.cpu(), ref)

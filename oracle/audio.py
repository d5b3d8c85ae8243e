"""Oracle restatement of the waveform -> log-mel front end (numpy/torch CPU, fp32).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Pinned by tests/golden/stft_mel_64f.npz, produced
by running the reference's own audioldm/audio/stft.py here.

Follows (relative to /root/reference/code/audioldm/audio):
  tools.py:46-49   normalize_wav           tools.py:34-44  pad_wav
  tools.py:52-64   read_wav_file (second normalisation :61-62 -- the waveform is normalised TWICE)
  tools.py:18-31   _pad_spec               tools.py:67-85  wav_to_fbank
  stft.py:15-50    STFT.__init__ (DFT basis [1026,1024], periodic Hann)
  stft.py:52-81    STFT.transform (reflect pad 512, stride-160 correlation, magnitude)
  stft.py:159-180  TacotronSTFT.mel_spectrogram;  audio_processing.py:85-91 log(clamp(x,1e-5))
  stft.py:145-149  librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) -- librosa 0.9.2 defaults
                   (Slaney scale, Slaney area norm): THIRD PARTY, absent here; restated from the
                   published Slaney definition; values pinned only to transformers'
                   mel_filter_bank(norm="slaney", mel_scale="slaney").
"""
import numpy as np
import torch
import torch.nn.functional as F


def normalize_wav(w):
    w = w - np.mean(w)
    w = w / (np.max(np.abs(w)) + 1e-8)
    return w * 0.5


def prepare_waveform(w, segment_length):
    """read_wav_file after resampling: normalise, pad/crop to segment_length, normalise again."""
    w = normalize_wav(np.asarray(w))
    n = w.shape[-1]
    # n > segment_length: the reference slices `waveform[:segment_length]` on the [1, N] array (tools.py:39-40 after
    # :58), i.e. the FIRST axis -- the waveform is NOT cropped; the surplus frames are cut later by _pad_spec.
    # (Pinned by tests/golden/waveform_prep.npz "long"; unreachable from load_audio for clips > 0.42 s, where
    # int(duration*102.4)*160 >= duration*16000.)
    if n < segment_length:
        tmp = np.zeros(segment_length)          # float64, as the reference's pad_wav
        tmp[:n] = w
        w = tmp
    w = w / np.max(np.abs(w))
    return (0.5 * w).astype(np.float32)


def stft_basis(n_fft=1024):
    """Windowed forward DFT basis [2*(n_fft/2+1), n_fft] fp32 (real rows then imag rows)."""
    k = np.arange(n_fft // 2 + 1)[:, None].astype(np.float64)
    n = np.arange(n_fft)[None, :].astype(np.float64)
    ang = 2.0 * np.pi * k * n / n_fft
    basis = np.vstack([np.cos(ang), -np.sin(ang)]).astype(np.float32)
    win = (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n_fft) / n_fft)).astype(np.float32)  # hann, fftbins=True
    return basis * win[None, :]


def _hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def _mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_basis(sr=16000, n_fft=1024, n_mels=64, fmin=0.0, fmax=8000.0):
    """Slaney-scale triangular filters with Slaney (area) normalisation -> fp32 [n_mels, n_fft/2+1]."""
    fft_freqs = np.linspace(0, sr / 2, n_fft // 2 + 1)
    mel_pts = np.linspace(_hz_to_mel_slaney(fmin), _hz_to_mel_slaney(fmax), n_mels + 2)
    hz = _mel_to_hz_slaney(mel_pts)
    fdiff = np.diff(hz)
    ramps = hz[:, None] - fft_freqs[None, :]
    w = np.zeros((n_mels, n_fft // 2 + 1))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (hz[2:n_mels + 2] - hz[:n_mels])
    return (w * enorm[:, None]).astype(np.float32)


def stft_magnitude(y, basis, hop=160):
    """y [B, N] fp32 -> magnitude [B, n_fft/2+1, frames]."""
    n_fft = basis.shape[1]
    x = F.pad(y[:, None, None, :], (n_fft // 2, n_fft // 2, 0, 0), mode="reflect")[:, 0]
    ft = F.conv1d(x, torch.from_numpy(basis)[:, None, :], stride=hop)
    cut = n_fft // 2 + 1
    return torch.sqrt(ft[:, :cut] ** 2 + ft[:, cut:] ** 2)


def mel_spectrogram(y, basis=None, melb=None, hop=160):
    """TacotronSTFT.mel_spectrogram: returns (log-mel [B,n_mels,F], log-magnitude, energy)."""
    assert torch.min(y) >= -1 and torch.max(y) <= 1
    basis = stft_basis() if basis is None else basis
    melb = mel_basis() if melb is None else melb
    mag = stft_magnitude(y, basis, hop)
    mel = torch.log(torch.clamp(torch.matmul(torch.from_numpy(melb), mag), min=1e-5))
    return mel, torch.log(torch.clamp(mag, min=1e-5)), torch.norm(mag, dim=1)


def pad_spec(fbank, target_length):
    n = fbank.shape[0]
    if target_length > n:
        fbank = F.pad(fbank, (0, 0, 0, target_length - n))
    elif target_length < n:
        fbank = fbank[:target_length]
    if fbank.shape[-1] % 2 != 0:
        fbank = fbank[..., :-1]
    return fbank


def wav_to_fbank(wave, target_length, basis=None, melb=None):
    """wav_to_fbank with the file read replaced by an in-memory 16 kHz waveform -> [target_length, 64]."""
    w = torch.from_numpy(prepare_waveform(wave, target_length * 160))
    mel, _, _ = mel_spectrogram(torch.clip(w[None], -1, 1), basis, melb)
    return pad_spec(mel[0].T.contiguous(), target_length), w

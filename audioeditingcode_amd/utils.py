"""Audio I/O + helpers mirroring the reference's code/utils.py (load_audio :53-95, get_spec :49-50,
set_reproducability :98-116, get_text_embeddings :217-231) and audioldm/audio/tools.py
(normalize_wav :46-49, pad_wav :34-44, read_wav_file :52-64, _pad_spec :18-31, wav_to_fbank :67-85)."""
import math
import random
import wave
from typing import List, NamedTuple, Optional, Tuple

import numpy as np
import torch


class PromptEmbeddings(NamedTuple):           # pc_drift.py:10-13
    embedding_hidden_states: torch.Tensor
    embedding_class_lables: torch.Tensor
    boolean_prompt_mask: torch.Tensor


def normalize_wav(waveform):
    waveform = waveform - np.mean(waveform)
    waveform = waveform / (np.max(np.abs(waveform)) + 1e-8)
    return waveform * 0.5


def pad_wav(waveform, segment_length):
    n = waveform.shape[-1]
    assert n > 100, "Waveform is too short, %s" % n
    if segment_length is None or n == segment_length:
        return waveform
    if n > segment_length:
        # the reference slices the FIRST axis of the [1, N] array here (tools.py:39-40): a no-op, the waveform is not
        # cropped and the surplus frames are cut by _pad_spec; kept as is (pinned by tests/golden/waveform_prep.npz)
        return waveform[:segment_length]
    tmp = np.zeros((1, segment_length))          # float64 like the reference
    tmp[:, :n] = waveform
    return tmp


def prepare_waveform(waveform_16k, segment_length):
    """read_wav_file after loading/resampling: normalise, pad/crop, normalise AGAIN (tools.py:57,61-62)."""
    w = normalize_wav(np.asarray(waveform_16k))[None, ...]
    w = pad_wav(w, segment_length)
    w = w / np.max(np.abs(w))
    return (0.5 * w)[0].astype(np.float32)


def read_wav(path) -> Tuple[np.ndarray, int]:
    """Mono float waveform + sample rate.  torchaudio when present, else the stdlib wave module (PCM16/32)."""
    try:
        import torchaudio
        w, sr = torchaudio.load(path)
        return w.numpy()[0], sr
    except ImportError:
        with wave.open(path, "rb") as f:
            sr, nch, sw, n = f.getframerate(), f.getnchannels(), f.getsampwidth(), f.getnframes()
            raw = f.readframes(n)
        dt = {2: np.int16, 4: np.int32}[sw]
        x = np.frombuffer(raw, dtype=dt).reshape(-1, nch).astype(np.float32) / float(np.iinfo(dt).max + 1)
        return x[:, 0], sr


def read_wav_channels(path) -> Tuple[np.ndarray, int]:
    """[channels, n] float waveform + sample rate (`torchaudio.load`'s layout, utils.py:78)."""
    try:
        import torchaudio
        w, sr = torchaudio.load(path)
        return w.numpy(), sr
    except ImportError:
        with wave.open(path, "rb") as f:
            sr, nch, sw, n = f.getframerate(), f.getnchannels(), f.getsampwidth(), f.getnframes()
            raw = f.readframes(n)
        if sw == 1:                                     # 8-bit PCM is unsigned, centred on 128
            x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
        elif sw == 3:                                   # 24-bit little-endian: widen to int32 (sign in the top byte)
            b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
            x = ((b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)) << 8 >> 8).astype(np.float32) / float(1 << 23)
        elif sw in (2, 4):
            dt = {2: np.int16, 4: np.int32}[sw]
            x = np.frombuffer(raw, dtype=dt).astype(np.float32) / float(np.iinfo(dt).max + 1)
        else:
            raise ValueError(f"{path}: unsupported PCM sample width {sw} bytes (torchaudio is not installed)")
        return np.ascontiguousarray(x.reshape(-1, nch).T), sr


def resample(w, sr, new_sr):
    if sr == new_sr:
        return w
    try:
        import torchaudio
        return torchaudio.functional.resample(torch.from_numpy(w)[None], orig_freq=sr, new_freq=new_sr)[0].numpy()
    except ImportError:
        from scipy.signal import resample_poly
        g = math.gcd(sr, new_sr)
        return resample_poly(w, new_sr // g, sr // g).astype(np.float32)


def get_duration(path):
    with wave.open(path, "rb") as f:           # audioldm/utils.py:17-21
        return f.getnframes() / float(f.getframerate())


def pad_spec(fbank, target_length=1024):
    n = fbank.shape[0]
    p = target_length - n
    if p > 0:
        fbank = torch.nn.functional.pad(fbank, (0, 0, 0, p))
    elif p < 0:
        fbank = fbank[0:target_length, :]
    if fbank.size(-1) % 2 != 0:
        fbank = fbank[..., :-1]
    return fbank


def wav_to_fbank(waveform_16k, target_length, fn_STFT):
    """In-memory waveform -> (fbank [target_length, n_mels] on the STFT's device, waveform tensor)."""
    w = torch.from_numpy(prepare_waveform(waveform_16k, target_length * 160))
    mel, _, _ = fn_STFT.mel_spectrogram(torch.clip(w[None], -1, 1))
    fbank = pad_spec(mel[0].T, target_length)
    return fbank, w


def get_spec(wav: torch.Tensor, fn_STFT) -> torch.Tensor:
    return fn_STFT.mel_spectrogram(torch.clip(wav[:, 0], -1, 1).cpu())[0]


def load_audio(audio_path, fn_STFT, left: int = 0, right: int = 0, device: Optional[torch.device] = None,
               return_wav: bool = False, stft: bool = False, model_sr: Optional[int] = None):
    """code/utils.py:53-95: the AudioLDM/TANGO spectrogram branch (stft=True) and the Stable Audio raw-waveform branch
    (stft=False -> (waveform [channels, n] normalised to +-0.5 over all channels, model_sr, duration)).  `audio_path` may
    also be a (waveform, sample_rate) pair for in-memory clips."""
    if not stft:
        if isinstance(audio_path, str):
            wav, sr = read_wav_channels(audio_path)
        else:
            wav, sr = audio_path
            wav = np.asarray(wav, dtype=np.float32)
            wav = wav[None] if wav.ndim == 1 else wav
        if sr != model_sr:
            wav = np.stack([resample(c, sr, model_sr) for c in wav])
        w = torch.from_numpy(np.ascontiguousarray(wav)).float()
        w = w - torch.mean(w)                                   # utils.py:83-88
        w = w / (torch.max(torch.abs(w)) + 1e-8)
        w = w * 0.5
        return w, model_sr, w.shape[-1] / model_sr
    if isinstance(audio_path, str):
        wav, sr = read_wav(audio_path)
        duration = get_duration(audio_path)
    elif isinstance(audio_path, tuple):
        wav, sr = audio_path
        duration = len(wav) / float(sr)
    else:
        mel, duration, wav = audio_path, None, None
    if wav is not None:
        wav16 = resample(np.asarray(wav, dtype=np.float32), sr, 16000)
        fbank, wav_t = wav_to_fbank(wav16, target_length=int(duration * 102.4), fn_STFT=fn_STFT)
        mel = fbank.unsqueeze(0)
    c, h, w = mel.shape
    left = min(left, w - 1)
    right = min(right, w - left - 1)
    mel = mel[:, :, left:w - right]
    mel = mel.unsqueeze(0).to(device)
    if return_wav:
        return mel, 16000, duration, wav_t
    return mel, model_sr, duration


def set_reproducability(seed: int, extreme: bool = True) -> None:
    if seed is not None:
        torch.manual_seed(seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed_all(seed)
        random.seed(seed)
        np.random.seed(seed)


def get_text_embeddings(target_prompt: List[str], target_neg_prompt: List[str], ldm_stable):
    a = ldm_stable.encode_text(target_prompt)
    b = ldm_stable.encode_text(target_neg_prompt)
    text_emb = PromptEmbeddings(embedding_hidden_states=a[0], boolean_prompt_mask=a[2], embedding_class_lables=a[1])
    uncond_emb = PromptEmbeddings(embedding_hidden_states=b[0], boolean_prompt_mask=b[2], embedding_class_lables=b[1])
    return a[1], text_emb, uncond_emb


def write_wav(path, wav, sr=16000):
    """16-bit PCM.  wav: [n] (mono) or [channels, n] (the layout `torchaudio.save` takes in main_run.py:223-224; the
    Stable Audio path hands over [2, n]) -- channels are interleaved frame by frame, none is dropped."""
    x = np.clip(np.asarray(wav, dtype=np.float32), -1, 1)
    if x.ndim == 1:
        x = x[None]
    if x.ndim != 2:
        raise ValueError(f"write_wav expects [n] or [channels, n], got shape {x.shape}")
    with wave.open(path, "wb") as f:
        f.setnchannels(x.shape[0])
        f.setsampwidth(2)
        f.setframerate(sr)
        f.writeframes(np.ascontiguousarray((x * 32767.0).astype(np.int16).T).tobytes())


def synthetic_clip(seconds=10.0, sr=16000, seed=1234):
    """SURVEY 8(d) synthetic benchmark clip: 100->4000 Hz chirp + 0.05 randn (CPU generator)."""
    n = int(seconds * sr)
    g = torch.Generator().manual_seed(seed)
    tt = torch.arange(n, dtype=torch.float64) / sr
    dur = n / sr
    phase = 2 * math.pi * (100.0 * tt + 0.5 * (4000.0 - 100.0) / dur * tt * tt)
    return (0.5 * torch.sin(phase).float() + 0.05 * torch.randn(n, generator=g)).numpy()

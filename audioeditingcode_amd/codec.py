"""Front / back end of the path as op tapes: STFT + log-mel (A2-A4), AutoencoderKL (A5, A13) and the
HiFi-GAN vocoder (A14).  Same kernels as the U-Net (conv_gemm / GroupNorm / softmax), other graphs.

Reference call sites (relative to /root/reference/code):
  audioldm/audio/stft.py:52-81,159-180      STFT.transform + TacotronSTFT.mel_spectrogram
  models.py:495-503, :581-589               vae_encode / vae_decode
  models.py:505-509, :591-597               decode_to_mel -> SpeechT5HifiGan
"""

import numpy as np
import torch

from . import _lib as L
from .tape import Tape


# =============================================================================================== STFT
def stft_basis(n_fft=1024):
    """Windowed forward DFT basis [2*(n_fft/2+1), n_fft] (stft.py:26-47): rows 0..cut-1 real, then imaginary;
    periodic Hann window (scipy get_window('hann', fftbins=True))."""
    k = np.arange(n_fft // 2 + 1)[:, None].astype(np.float64)
    n = np.arange(n_fft)[None, :].astype(np.float64)
    ang = 2.0 * np.pi * k * n / n_fft
    basis = np.vstack([np.cos(ang), -np.sin(ang)]).astype(np.float32)
    win = (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n_fft) / n_fft)).astype(np.float32)
    return torch.from_numpy(basis * win[None, :])


def mel_filterbank(sr=16000, n_fft=1024, n_mels=64, fmin=0.0, fmax=8000.0):
    """librosa.filters.mel defaults of 0.9.2 (stft.py:145-149): Slaney mel scale, Slaney area normalisation."""
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0

    def hz2mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, f / f_sp)

    def mel2hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    freqs = np.linspace(0, sr / 2, n_fft // 2 + 1)
    hz = mel2hz(np.linspace(hz2mel(fmin), hz2mel(fmax), n_mels + 2))
    fdiff = np.diff(hz)
    ramps = hz[:, None] - freqs[None, :]
    w = np.zeros((n_mels, n_fft // 2 + 1))
    for i in range(n_mels):
        w[i] = np.maximum(0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    w *= (2.0 / (hz[2:n_mels + 2] - hz[:n_mels]))[:, None]
    return torch.from_numpy(w.astype(np.float32))


class STFTEngine:
    """wav [B, N] -> log-mel [B, frames, n_mels] (frames = N // hop + 1).  Dense windowed DFT on the matrix
    cores (the reference's own formulation: a stride-`hop` conv1d with the DFT basis), magnitude, mel, log."""

    def __init__(self, cfg, device, batch, n_samples, with_aux=False):
        self.device = torch.device(device)
        n_fft, hop = cfg["filter_length"], cfg["hop_length"]
        self.n_fft, self.hop, self.n_mels = n_fft, hop, cfg["n_mel_channels"]
        cut = n_fft // 2 + 1
        B, N = batch, n_samples
        self.frames = N // hop + 1
        F = self.frames
        tp = self.tape = Tape(device)
        self.basis = stft_basis(n_fft).to(self.device)
        self.mel_basis = mel_filterbank(cfg["sampling_rate"], n_fft, self.n_mels, cfg["mel_fmin"], cfg["mel_fmax"])
        Kp = ((cut + 31) // 32) * 32
        melw = torch.zeros(self.n_mels, Kp)
        melw[:, :cut] = self.mel_basis
        self.melw = melw.to(self.device)
        ldp = ((N + n_fft + 3) // 4) * 4
        self.wav = tp.alloc(B, N, zero=True)
        self.padded = tp.alloc(B, ldp, zero=True)
        tp.reflect_pad(self.wav, self.padded, B=B, N=N, pad=n_fft // 2, ldd=ldp)
        self.ft = tp.alloc(B * F, 2 * cut)
        tp.conv(self.padded, self.basis, None, self.ft, B=B, IH=F, IW=1, Cin=n_fft, OH=F, OW=1, N=2 * cut, lda=hop,
                a_bs=ldp, ldc=2 * cut, name="stft.dft")
        self.mag = tp.alloc(B * F, Kp)
        tp.magnitude(self.ft, self.mag, F=B * F, cut=cut, ld_ft=2 * cut, ld_mag=Kp)
        self.mel = tp.alloc(B, F, self.n_mels)
        tp.linear(self.mag, self.melw, None, self.mel.view(B * F, self.n_mels), M=B * F, K=Kp, N=self.n_mels,
                  out_act=L.ACT_LOGCLAMP, out_p=1e-5, name="stft.mel")
        tp.finalize()

    @torch.inference_mode()
    def __call__(self, wav):
        """wav: [B, N] float tensor in [-1, 1] (asserted like stft.py:169-170)."""
        assert float(wav.min()) >= -1 and float(wav.max()) <= 1, "waveform outside [-1, 1]"
        self.wav.copy_(wav.to(self.device, torch.float32))
        self.tape.run()
        return self.mel


# =============================================================================================== VAE
class _ConvNet:
    """Shared block emitters for the VAE graphs (GroupNorm eps 1e-6, SiLU, 3x3 convs; no time embedding)."""

    def __init__(self, device, sd):
        self.device = torch.device(device)
        self.tape = Tape(device)
        self._tmp = {}
        self.wd = {}
        for k, v in sd.items():
            if v.dim() == 4:
                v = v.permute(0, 2, 3, 1).reshape(v.shape[0], -1)
            self.wd[k] = v.contiguous().to(self.device, torch.float32)

    def tmp(self, tag, *shape):
        key = (tag, tuple(shape))
        if key not in self._tmp:
            self._tmp[key] = self.tape.alloc(*shape)
        return self._tmp[key]

    def conv3(self, p, x, B, H, W, Cin, Cout, out=None, stride=1, pad=1, up=0, res=None, OH=None, OW=None):
        OH = (H << up) if OH is None else OH
        OW = (W << up) if OW is None else OW
        out = self.tape.alloc(B, OH, OW, Cout) if out is None else out
        self.tape.conv(x, self.wd[p + ".weight"], self.wd[p + ".bias"], out, B=B, IH=H, IW=W, Cin=Cin, OH=OH, OW=OW,
                       N=Cout, KH=3, KW=3, stride=stride, pad_h=pad, pad_w=pad, up=up, res=res, name=p)
        return out

    def resnet(self, p, x, B, H, W, Cin, Cout, groups):
        tp, wd = self.tape, self.wd
        a = self.tmp("a", B, H, W, Cin)
        tp.groupnorm(x, wd[p + ".norm1.weight"], wd[p + ".norm1.bias"], a, B=B, HW=H * W, C=Cin, G=groups, eps=1e-6,
                     act=L.ACT_SILU, name=p + ".norm1")
        h = self.tmp("h", B, H, W, Cout)
        self.conv3(p + ".conv1", a, B, H, W, Cin, Cout, out=h)
        a2 = self.tmp("a2", B, H, W, Cout)
        tp.groupnorm(h, wd[p + ".norm2.weight"], wd[p + ".norm2.bias"], a2, B=B, HW=H * W, C=Cout, G=groups, eps=1e-6,
                     act=L.ACT_SILU, name=p + ".norm2")
        res = x
        if (p + ".conv_shortcut.weight") in wd:
            res = self.tmp("sc", B, H, W, Cout)
            tp.conv(x, wd[p + ".conv_shortcut.weight"], wd[p + ".conv_shortcut.bias"], res, B=B, IH=H, IW=W, Cin=Cin,
                    OH=H, OW=W, N=Cout, name=p + ".conv_shortcut")
        return self.conv3(p + ".conv2", a2, B, H, W, Cout, Cout, res=res)

    def mid_attention(self, p, x, B, H, W, C, groups):
        """Single-head attention over H*W tokens with d = C (variational_autoencoder/modules.py:185-230).
        d = 512 is far beyond the fused kernel's head sizes, and it runs once per encode/decode, so the
        score matrix is materialised: QK^T GEMM -> row softmax -> PV GEMM, per batch item."""
        tp, wd = self.tape, self.wd
        N = H * W
        n = self.tmp("attn_n", B, N, C)
        tp.groupnorm(x, wd[p + ".group_norm.weight"], wd[p + ".group_norm.bias"], n, B=B, HW=N, C=C, G=groups,
                     eps=1e-6, name=p + ".group_norm")
        q, k, v = (self.tmp("attn_" + s, B, N, C) for s in "qkv")
        for nm, buf in (("to_q", q), ("to_k", k), ("to_v", v)):
            tp.linear(n.view(B * N, C), wd[f"{p}.{nm}.weight"], wd[f"{p}.{nm}.bias"], buf.view(B * N, C), M=B * N, K=C,
                      N=C, name=f"{p}.{nm}")
        s = self.tmp("attn_s", N, N)
        vt = self.tmp("attn_vt", C, N)
        o = self.tmp("attn_o", B, N, C)
        for b in range(B):
            tp.linear(q[b], k[b], None, s, M=N, K=C, N=N, name=p + ".qk")
            tp.softmax_rows(s, s, rows=N, cols=N, scale=float(C) ** -0.5, name=p + ".softmax")
            tp.transpose(v[b], vt, Bt=1, R=N, C=C, name=p + ".v_t")
            tp.linear(s, vt, None, o[b], M=N, K=N, N=C, name=p + ".pv")
        out = tp.alloc(B, H, W, C)
        tp.linear(o.view(B * N, C), wd[p + ".to_out.0.weight"], wd[p + ".to_out.0.bias"], out.view(B * N, C),
                  M=B * N, K=C, N=C, res=x.view(B * N, C), name=p + ".to_out")
        return out

    def mid(self, p, x, B, H, W, C, groups):
        x = self.resnet(p + ".resnets.0", x, B, H, W, C, C, groups)
        x = self.mid_attention(p + ".attentions.0", x, B, H, W, C, groups)
        return self.resnet(p + ".resnets.1", x, B, H, W, C, C, groups)


class VAEEncoder(_ConvNet):
    """mel [B,1,T,F] (== channels-last [B,T,F,1]) -> scaled latent mean [B, T/4, F/4, latent] (A5)."""

    def __init__(self, cfg, sd, device, batch, T, F):
        super().__init__(device, {k: v for k, v in sd.items() if k.startswith(("encoder.", "quant_conv."))})
        boc, groups = cfg["block_out_channels"], cfg.get("norm_num_groups", 32)
        lc = cfg.get("latent_channels", 8)
        tp, B = self.tape, batch
        self.x_in = tp.alloc(B, T, F, cfg.get("in_channels", 1), zero=True)
        h = self.conv3("encoder.conv_in", self.x_in, B, T, F, cfg.get("in_channels", 1), boc[0])
        ch, hh, ww = boc[0], T, F
        for i in range(len(boc)):
            for j in range(cfg.get("layers_per_block", 2)):
                h = self.resnet(f"encoder.down_blocks.{i}.resnets.{j}", h, B, hh, ww, ch, boc[i], groups)
                ch = boc[i]
            if i < len(boc) - 1:
                oh, ow = (hh + 1 - 3) // 2 + 1, (ww + 1 - 3) // 2 + 1     # F.pad (0,1,0,1) then stride 2, pad 0
                h = self.conv3(f"encoder.down_blocks.{i}.downsamplers.0.conv", h, B, hh, ww, ch, ch, stride=2, pad=0,
                               OH=oh, OW=ow)
                hh, ww = oh, ow
        h = self.mid("encoder.mid_block", h, B, hh, ww, ch, groups)
        a = self.tmp("a", B, hh, ww, ch)
        tp.groupnorm(h, self.wd["encoder.conv_norm_out.weight"], self.wd["encoder.conv_norm_out.bias"], a, B=B,
                     HW=hh * ww, C=ch, G=groups, eps=1e-6, act=L.ACT_SILU, name="encoder.conv_norm_out")
        mom = self.conv3("encoder.conv_out", a, B, hh, ww, ch, 2 * lc)
        # DiagonalGaussian.mode() = mean = first `lc` moment channels -> only those rows of quant_conv
        self.latent = tp.alloc(B, hh, ww, lc)
        tp.conv(mom, self.wd["quant_conv.weight"][:lc].contiguous(), self.wd["quant_conv.bias"][:lc].contiguous(),
                self.latent, B=B, IH=hh, IW=ww, Cin=2 * lc, OH=hh, OW=ww, N=lc, name="quant_conv(mean)")
        tp.axpby(self.latent, self.latent, numel=self.latent.numel(), a=float(cfg["scaling_factor"]), b=0.0,
                 name="latent*scaling_factor")
        self.h, self.w = hh, ww
        tp.finalize()

    @torch.inference_mode()
    def __call__(self, mel):
        self.x_in.copy_(mel.to(self.device, torch.float32).reshape(self.x_in.shape))
        self.tape.run()
        return self.latent


class VAEDecoder(_ConvNet):
    """latent [B,h,w,latent] channels-last -> mel [B, 4h, 4w, 1] (== NCHW [B,1,4h,4w]) (A13)."""

    def __init__(self, cfg, sd, device, batch, h, w):
        super().__init__(device, {k: v for k, v in sd.items() if k.startswith(("decoder.", "post_quant_conv."))})
        boc, groups = cfg["block_out_channels"], cfg.get("norm_num_groups", 32)
        lc = cfg.get("latent_channels", 8)
        tp, B = self.tape, batch
        self.z_in = tp.alloc(B, h, w, lc, zero=True)
        z = tp.alloc(B, h, w, lc)
        inv = float((torch.tensor(1.0) / torch.tensor(float(cfg["scaling_factor"]))).item())   # `1 / sf * x`
        tp.axpby(self.z_in, z, numel=z.numel(), a=inv, b=0.0, name="latent/scaling_factor")
        pq = tp.alloc(B, h, w, lc)
        tp.conv(z, self.wd["post_quant_conv.weight"], self.wd["post_quant_conv.bias"], pq, B=B, IH=h, IW=w, Cin=lc,
                OH=h, OW=w, N=lc, name="post_quant_conv")
        ch = boc[-1]
        x = self.conv3("decoder.conv_in", pq, B, h, w, lc, ch)
        x = self.mid("decoder.mid_block", x, B, h, w, ch, groups)
        hh, ww = h, w
        nb = len(boc)
        for i in range(nb):
            co = boc[nb - 1 - i]
            for j in range(cfg.get("layers_per_block", 2) + 1):
                x = self.resnet(f"decoder.up_blocks.{i}.resnets.{j}", x, B, hh, ww, ch, co, groups)
                ch = co
            if i < nb - 1:
                x = self.conv3(f"decoder.up_blocks.{i}.upsamplers.0.conv", x, B, hh, ww, ch, ch, up=1)
                hh, ww = 2 * hh, 2 * ww
        a = self.tmp("a", B, hh, ww, ch)
        tp.groupnorm(x, self.wd["decoder.conv_norm_out.weight"], self.wd["decoder.conv_norm_out.bias"], a, B=B,
                     HW=hh * ww, C=ch, G=groups, eps=1e-6, act=L.ACT_SILU, name="decoder.conv_norm_out")
        self.mel = self.conv3("decoder.conv_out", a, B, hh, ww, ch, cfg.get("out_channels", 1))
        tp.finalize()

    @torch.inference_mode()
    def __call__(self, z_nhwc):
        self.z_in.copy_(z_nhwc.reshape(self.z_in.shape))
        self.tape.run()
        return self.mel


# =============================================================================================== vocoder
class VocoderEngine:
    """HiFi-GAN generator (A14): mel [B, T, n_mels] -> waveform [B, T*prod(rates) (+ConvTranspose edge)].

    Every layer is the conv_gemm kernel over a [B, L, C] channels-last sequence: Conv1d = taps along L with
    dilation; ConvTranspose1d(k, stride u, pad p) = u phase convolutions (phase r owns taps r, r+u, ...
    and writes output positions u*q + r - p); LeakyReLU rides in the A-loader, the resblock residual, the
    3-way MRF accumulate/mean and the final tanh ride in the epilogue."""

    def __init__(self, cfg, sd, device, batch, T):
        self.device = torch.device(device)
        tp = self.tape = Tape(device)
        B = batch
        slope = cfg.get("leaky_relu_slope", 0.1)
        rates, ksz = cfg["upsample_rates"], cfg["upsample_kernel_sizes"]
        rks, rds = cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"]
        nk = len(rks)
        dev = lambda t: t.contiguous().to(self.device, torch.float32)      # noqa: E731
        c1d = lambda w: dev(w.permute(0, 2, 1).reshape(w.shape[0], -1))    # noqa: E731  [Co,Ci,k] -> [Co, k*Ci]
        if cfg.get("normalize_before", False):
            raise NotImplementedError("normalize_before=True vocoders are not used by AudioLDM/AudioLDM2")
        nm = cfg["model_in_dim"]
        self.mel_in = tp.alloc(B, T, nm, zero=True)
        c0 = cfg["upsample_initial_channel"]
        h = tp.alloc(B, T, c0)
        tp.conv(self.mel_in, c1d(sd["conv_pre.weight"]), dev(sd["conv_pre.bias"]), h, B=B, IH=T, IW=1, Cin=nm, OH=T,
                OW=1, N=c0, KH=7, KW=1, pad_h=3, name="conv_pre")
        Lc, ch = T, c0
        for i, (u, k) in enumerate(zip(rates, ksz)):
            co = c0 // (2 ** (i + 1))
            p = (k - u) // 2
            Lo = (Lc - 1) * u - 2 * p + k
            up = tp.alloc(B, Lo, co)
            w = sd[f"upsampler.{i}.weight"]                       # [Cin, Cout, k]
            bias = dev(sd[f"upsampler.{i}.bias"])
            for r in range(u):
                taps = list(range(r, k, u))
                if not taps:
                    continue
                wr = dev(torch.stack([w[:, :, j] for j in taps], 0).permute(2, 0, 1).reshape(co, -1))  # [Co, m*Ci]
                Q = Lc + len(taps) - 1
                tp.conv(h, wr, bias, up, B=B, IH=Lc, IW=1, Cin=ch, OH=Q, OW=1, N=co, KH=len(taps), KW=1, pad_h=0,
                        dil_h=-1, in_act=L.ACT_LEAKY, in_slope=slope, o_mul=u, o_add=r - p, o_len=Lo, out_bs=Lo,
                        name=f"upsampler.{i}.phase{r}")
            Lc, ch = Lo, co
            acc = tp.alloc(B, Lc, ch)
            xa, xb, y = tp.alloc(B, Lc, ch), tp.alloc(B, Lc, ch), tp.alloc(B, Lc, ch)
            for j in range(nk):
                pre = f"resblocks.{i * nk + j}"
                x = up
                for m, d in enumerate(rds[j]):
                    kk = rks[j]
                    tp.conv(x, c1d(sd[f"{pre}.convs1.{m}.weight"]), dev(sd[f"{pre}.convs1.{m}.bias"]), y, B=B, IH=Lc,
                            IW=1, Cin=ch, OH=Lc, OW=1, N=ch, KH=kk, KW=1, pad_h=(kk * d - d) // 2, dil_h=d,
                            in_act=L.ACT_LEAKY, in_slope=slope, name=f"{pre}.convs1.{m}")
                    last = m == len(rds[j]) - 1
                    dst = acc if last else (xa if x is not xa else xb)
                    mode = 0 if not last else (0 if j == 0 else (2 if j == nk - 1 else 1))
                    tp.conv(y, c1d(sd[f"{pre}.convs2.{m}.weight"]), dev(sd[f"{pre}.convs2.{m}.bias"]), dst, B=B,
                            IH=Lc, IW=1, Cin=ch, OH=Lc, OW=1, N=ch, KH=kk, KW=1, pad_h=(kk - 1) // 2,
                            in_act=L.ACT_LEAKY, in_slope=slope, res=x, accumulate=mode, out_div=float(nk),
                            name=f"{pre}.convs2.{m}")
                    x = dst
            h = acc
        self.wav = tp.alloc(B, Lc, 1)
        tp.conv(h, c1d(sd["conv_post.weight"]), dev(sd["conv_post.bias"]), self.wav, B=B, IH=Lc, IW=1, Cin=ch, OH=Lc,
                OW=1, N=1, KH=7, KW=1, pad_h=3, in_act=L.ACT_LEAKY, in_slope=0.01, out_act=L.ACT_TANH,
                name="conv_post")
        self.L_out = Lc
        tp.finalize()

    @torch.inference_mode()
    def __call__(self, mel):
        self.mel_in.copy_(mel.to(self.device, torch.float32).reshape(self.mel_in.shape))
        self.tape.run()
        return self.wav.view(self.wav.shape[0], self.L_out)

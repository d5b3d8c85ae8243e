"""Oracle restatement of the HiFi-GAN vocoder forward (torch CPU, fp32).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Pinned (tests/golden/hifigan_c64.npz) against
transformers' SpeechT5HifiGan -- the class the reference reaches through
``pipeline.mel_spectrogram_to_waveform`` (/root/reference/code/models.py:505-509, :591-597) --
and cross-checked against the in-tree twin audioldm/hifigan/models.py:20-174
(conv_pre k7 -> 5x [LeakyReLU 0.1, ConvTranspose1d, mean of 3 MRF resblocks] ->
LeakyReLU(0.01, :161) -> conv_post k7 -> tanh).
State-dict keys: SpeechT5HifiGan's (conv_pre, upsampler.i, resblocks.n.convs1/2.j, conv_post).
"""
import torch
import torch.nn.functional as F


def hifigan_forward(cfg, sd, mel):
    """mel [B, T, n_mels] -> waveform [B, T*prod(rates) (+edge)]."""
    rates = cfg["upsample_rates"]
    ksz = cfg["upsample_kernel_sizes"]
    rks = cfg["resblock_kernel_sizes"]
    rds = cfg["resblock_dilation_sizes"]
    slope = cfg.get("leaky_relu_slope", 0.1)
    if cfg.get("normalize_before", False):
        mel = (mel - sd["mean"]) / sd["scale"]
    h = F.conv1d(mel.transpose(2, 1), sd["conv_pre.weight"], sd["conv_pre.bias"], padding=3)
    nk = len(rks)
    for i, (u, k) in enumerate(zip(rates, ksz)):
        h = F.leaky_relu(h, slope)
        h = F.conv_transpose1d(h, sd[f"upsampler.{i}.weight"], sd[f"upsampler.{i}.bias"], stride=u,
                               padding=(k - u) // 2)
        acc = None
        for j in range(nk):
            p = f"resblocks.{i * nk + j}"
            x = h
            for m, d in enumerate(rds[j]):
                y = F.leaky_relu(x, slope)
                y = F.conv1d(y, sd[f"{p}.convs1.{m}.weight"], sd[f"{p}.convs1.{m}.bias"], dilation=d,
                             padding=(rks[j] * d - d) // 2)
                y = F.leaky_relu(y, slope)
                y = F.conv1d(y, sd[f"{p}.convs2.{m}.weight"], sd[f"{p}.convs2.{m}.bias"],
                             padding=(rks[j] - 1) // 2)
                x = y + x
            acc = x if acc is None else acc + x
        h = acc / nk
    h = F.leaky_relu(h)                       # default slope 0.01 (hifigan/models.py:161)
    h = F.conv1d(h, sd["conv_post.weight"], sd["conv_post.bias"], padding=3)
    return torch.tanh(h).squeeze(1)

"""Oracle restatement of AutoencoderKL encode/decode (torch CPU, fp32, NCHW).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Pinned against the in-tree twin
(tests/golden/vae_twin_c32.npz):
  audioldm/variational_autoencoder/modules.py:118-175 (ResnetBlock, eps 1e-6),
  :185-230 (AttnBlock, single head), :85-100 (Downsample: pad (0,1,0,1) + stride-2 conv),
  :42-56 (Upsample: nearest 2x + conv), :419-543 (Encoder), :546-683 (Decoder);
  distributions.py mode() = mean = first half of the moment channels.
Call sites: /root/reference/code/models.py:495-503, :581-589 (vae_encode front-pads the
time axis to a multiple of 4, scales by scaling_factor; vae_decode divides by it).
Weights use diffusers AutoencoderKL key names (restated; twin_to_diffusers maps the twin).
"""
import torch
import torch.nn.functional as F

EPS = 1e-6


def _conv(sd, p, x, stride=1, padding=1):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], stride=stride, padding=padding)


def _gn(sd, p, x, groups):
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], EPS)


def _res(sd, p, x, groups):
    h = _conv(sd, p + ".conv1", F.silu(_gn(sd, p + ".norm1", x, groups)))
    h = _conv(sd, p + ".conv2", F.silu(_gn(sd, p + ".norm2", h, groups)))
    if (p + ".conv_shortcut.weight") in sd:
        x = _conv(sd, p + ".conv_shortcut", x, padding=0)
    return x + h


def _mid_attn(sd, p, x, groups):
    b, c, hh, ww = x.shape
    h = _gn(sd, p + ".group_norm", x, groups).reshape(b, c, hh * ww).transpose(1, 2)
    q = F.linear(h, sd[p + ".to_q.weight"], sd[p + ".to_q.bias"])
    k = F.linear(h, sd[p + ".to_k.weight"], sd[p + ".to_k.bias"])
    v = F.linear(h, sd[p + ".to_v.weight"], sd[p + ".to_v.bias"])
    a = torch.softmax(torch.bmm(q, k.transpose(1, 2)) * (c ** -0.5), dim=-1)
    o = F.linear(torch.bmm(a, v), sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])
    return x + o.transpose(1, 2).reshape(b, c, hh, ww)


def _mid(sd, p, x, groups):
    x = _res(sd, p + ".resnets.0", x, groups)
    x = _mid_attn(sd, p + ".attentions.0", x, groups)
    return _res(sd, p + ".resnets.1", x, groups)


def encode_moments(cfg, sd, x):
    groups = cfg.get("norm_num_groups", 32)
    nb = len(cfg["block_out_channels"])
    h = _conv(sd, "encoder.conv_in", x)
    for i in range(nb):
        for j in range(cfg.get("layers_per_block", 2)):
            h = _res(sd, f"encoder.down_blocks.{i}.resnets.{j}", h, groups)
        if i < nb - 1:
            h = F.pad(h, (0, 1, 0, 1))
            h = _conv(sd, f"encoder.down_blocks.{i}.downsamplers.0.conv", h, stride=2, padding=0)
    h = _mid(sd, "encoder.mid_block", h, groups)
    h = _conv(sd, "encoder.conv_out", F.silu(_gn(sd, "encoder.conv_norm_out", h, groups)))
    return _conv(sd, "quant_conv", h, padding=0)


def decode(cfg, sd, z):
    groups = cfg.get("norm_num_groups", 32)
    nb = len(cfg["block_out_channels"])
    h = _conv(sd, "post_quant_conv", z, padding=0)
    h = _conv(sd, "decoder.conv_in", h)
    h = _mid(sd, "decoder.mid_block", h, groups)
    for i in range(nb):
        for j in range(cfg.get("layers_per_block", 2) + 1):
            h = _res(sd, f"decoder.up_blocks.{i}.resnets.{j}", h, groups)
        if i < nb - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(sd, f"decoder.up_blocks.{i}.upsamplers.0.conv", h)
    return _conv(sd, "decoder.conv_out", F.silu(_gn(sd, "decoder.conv_norm_out", h, groups)))


def vae_encode(cfg, sd, mel):
    """models.py:581-585."""
    if mel.shape[2] % 4:
        mel = F.pad(mel, (0, 0, 4 - (mel.shape[2] % 4), 0))
    mean = encode_moments(cfg, sd, mel)[:, :cfg.get("latent_channels", 8)]
    return (mean * cfg["scaling_factor"]).float()


def vae_decode(cfg, sd, z):
    """models.py:588-589."""
    return decode(cfg, sd, 1 / cfg["scaling_factor"] * z)


def twin_to_diffusers(tsd, n_levels, num_res_blocks):
    """Twin (Encoder/Decoder, 1x1-conv attention) -> diffusers AutoencoderKL names (pinning only)."""
    out = {}

    def cp(src, dst):
        for s in ("weight", "bias"):
            if f"{src}.{s}" in tsd:
                out[f"{dst}.{s}"] = tsd[f"{src}.{s}"]

    def res(src, dst):
        for a in ("norm1", "conv1", "norm2", "conv2"):
            cp(f"{src}.{a}", f"{dst}.{a}")
        cp(f"{src}.nin_shortcut", f"{dst}.conv_shortcut")

    def attn(src, dst):
        cp(f"{src}.norm", f"{dst}.group_norm")
        for a, b in (("q", "to_q"), ("k", "to_k"), ("v", "to_v"), ("proj_out", "to_out.0")):
            out[f"{dst}.{b}.weight"] = tsd[f"{src}.{a}.weight"][:, :, 0, 0]
            out[f"{dst}.{b}.bias"] = tsd[f"{src}.{a}.bias"]

    for side in ("encoder", "decoder"):
        cp(f"{side}.conv_in", f"{side}.conv_in")
        cp(f"{side}.norm_out", f"{side}.conv_norm_out")
        cp(f"{side}.conv_out", f"{side}.conv_out")
        res(f"{side}.mid.block_1", f"{side}.mid_block.resnets.0")
        attn(f"{side}.mid.attn_1", f"{side}.mid_block.attentions.0")
        res(f"{side}.mid.block_2", f"{side}.mid_block.resnets.1")
    for i in range(n_levels):
        for j in range(num_res_blocks):
            res(f"encoder.down.{i}.block.{j}", f"encoder.down_blocks.{i}.resnets.{j}")
        cp(f"encoder.down.{i}.downsample.conv", f"encoder.down_blocks.{i}.downsamplers.0.conv")
        for j in range(num_res_blocks + 1):
            res(f"decoder.up.{i}.block.{j}", f"decoder.up_blocks.{n_levels - 1 - i}.resnets.{j}")
        cp(f"decoder.up.{i}.upsample.conv", f"decoder.up_blocks.{n_levels - 1 - i}.upsamplers.0.conv")
    cp("quant_conv", "quant_conv")
    cp("post_quant_conv", "post_quant_conv")
    return out

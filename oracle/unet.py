"""Oracle restatement of the eps-prediction U-Net forward (torch CPU, fp32, NCHW).

TEST INFRASTRUCTURE (see oracle/__init__.py).

Graph order and hooks follow the reference's inline forwards
  /root/reference/code/models.py:160-393   (PipelineWrapper.unet_forward: AudioLDM-1 / TANGO)
  /root/reference/code/models.py:691-899   (AudioLDM2Wrapper.unet_forward)
Block internals follow the in-tree AudioLDM-1 twin (which pins them, tests/golden/unet_twin_c32.npz):
  audioldm/latent_diffusion/openaimodel.py:175-286 (ResBlock), :432-851 (UNetModel)
  audioldm/latent_diffusion/attention.py:149-323 (CrossAttention), :370-410 (BasicTransformerBlock),
  :413-469 (SpatialTransformer);  util.py:173-197 (timestep_embedding)
The module *layout* (diffusers UNet2DConditionModel / AudioLDM2UNet2DConditionModel key names, the
3-transformers-per-layer AudioLDM2 stack, additive -10000 key mask) is restated from diffusers'
published semantics: PARITY UNPINNED there (diffusers absent).

Weights: a flat dict of torch tensors with diffusers key names.
"""
import math

import torch
import torch.nn.functional as F


def timestep_embedding(t, dim, flip_sin_to_cos=True, freq_shift=0.0, max_period=10000):
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32) / (half - freq_shift)
    emb = t[:, None].float() * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _conv(sd, p, x, stride=1, padding=1):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding)


def _gn(sd, p, x, groups, eps):
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], eps)


def resnet(sd, p, x, temb, groups, eps):
    h = _conv(sd, p + ".conv1", F.silu(_gn(sd, p + ".norm1", x, groups, eps)))
    if temb is not None and (p + ".time_emb_proj.weight") in sd:
        h = h + _lin(sd, p + ".time_emb_proj", F.silu(temb))[:, :, None, None]
    h = _conv(sd, p + ".conv2", F.silu(_gn(sd, p + ".norm2", h, groups, eps)))
    if (p + ".conv_shortcut.weight") in sd:
        x = _conv(sd, p + ".conv_shortcut", x, padding=0)
    return x + h


def attention(sd, p, x, ctx, heads, bias=None):
    """diffusers Attention / twin CrossAttention: softmax(q k^T / sqrt(d) + bias) v, then to_out."""
    ctx = x if ctx is None else ctx
    q = _lin(sd, p + ".to_q", x)
    k = _lin(sd, p + ".to_k", ctx)
    v = _lin(sd, p + ".to_v", ctx)
    b, n, c = q.shape
    d = c // heads
    q = q.view(b, n, heads, d).transpose(1, 2)
    k = k.view(b, -1, heads, d).transpose(1, 2)
    v = v.view(b, -1, heads, d).transpose(1, 2)
    s = torch.matmul(q, k.transpose(-1, -2)) * (d ** -0.5)
    if bias is not None:
        s = s + bias[:, None, :, :]          # bias [B,1,Mkeys] -> [B,1,1,M]
    a = torch.softmax(s, dim=-1)
    o = torch.matmul(a, v).transpose(1, 2).reshape(b, n, c)
    return _lin(sd, p + ".to_out.0", o)


def transformer_block(sd, p, x, ctx, heads, bias=None, double_self=False):
    x = attention(sd, p + ".attn1", F.layer_norm(x, x.shape[-1:], sd[p + ".norm1.weight"], sd[p + ".norm1.bias"]),
                  None, heads) + x
    h = F.layer_norm(x, x.shape[-1:], sd[p + ".norm2.weight"], sd[p + ".norm2.bias"])
    x = attention(sd, p + ".attn2", h, None if double_self else ctx, heads, None if double_self else bias) + x
    h = F.layer_norm(x, x.shape[-1:], sd[p + ".norm3.weight"], sd[p + ".norm3.bias"])
    g = _lin(sd, p + ".ff.net.0.proj", h)
    a, gate = g.chunk(2, dim=-1)
    return _lin(sd, p + ".ff.net.2", a * F.gelu(gate)) + x


def transformer2d(sd, p, x, ctx, heads, groups, bias=None, double_self=False, linear_proj=False, depth=1):
    b, c, hh, ww = x.shape
    res = x
    h = _gn(sd, p + ".norm", x, groups, 1e-6)
    if not linear_proj:
        h = _conv(sd, p + ".proj_in", h, padding=0)
        h = h.permute(0, 2, 3, 1).reshape(b, hh * ww, -1)
    else:
        h = _lin(sd, p + ".proj_in", h.permute(0, 2, 3, 1).reshape(b, hh * ww, c))
    for d in range(depth):
        h = transformer_block(sd, f"{p}.transformer_blocks.{d}", h, ctx, heads, bias, double_self)
    if not linear_proj:
        h = h.reshape(b, hh, ww, -1).permute(0, 3, 1, 2)
        h = _conv(sd, p + ".proj_out", h, padding=0)
    else:
        h = _lin(sd, p + ".proj_out", h).reshape(b, hh, ww, c).permute(0, 3, 1, 2)
    return h + res


def _per_block(v, n):
    return list(v) if isinstance(v, (list, tuple)) else [v] * n


def _ctx_list(cfg):
    """Normalise cross_attention_dim to: per block, a list of per-transformer dims (None = double self)."""
    n = len(cfg["block_out_channels"])
    cad = cfg.get("cross_attention_dim")
    if isinstance(cad, (list, tuple)) and len(cad) and isinstance(cad[0], (list, tuple)):
        return [list(c) for c in cad], True     # AudioLDM2 form
    return [[c] for c in _per_block(cad, n)], False


def unet_forward(cfg, sd, sample, timestep, encoder_hidden_states=None, class_labels=None,
                 encoder_attention_mask=None, encoder_hidden_states_1=None, encoder_attention_mask_1=None,
                 mid_block_additional_residual=None, replace_h_space=None, replace_skip_conns=None,
                 zero_out_resconns=None):
    """Returns (eps, h_space, extracted_res_conns) like the reference wrappers."""
    boc = cfg["block_out_channels"]
    nb = len(boc)
    groups = cfg.get("norm_num_groups", 32)
    eps = cfg.get("norm_eps", 1e-5)
    lpb = cfg.get("layers_per_block", 2)
    heads_pb = _per_block(cfg.get("num_attention_heads") or cfg.get("attention_head_dim", 8), nb)
    ctx_pb, multi = _ctx_list(cfg)
    linear_proj = cfg.get("use_linear_projection", False)
    depth = cfg.get("transformer_layers_per_block", 1)

    def mask_bias(m):
        return None if m is None else ((1 - m.to(sample.dtype)) * -10000.0).unsqueeze(1)   # models.py:740-755

    bias0, bias1 = mask_bias(encoder_attention_mask), mask_bias(encoder_attention_mask_1)

    def run_attn(prefix, k0, x, dims, heads):
        """One attention site: len(dims) Transformer2D modules starting at index k0."""
        for j, cdim in enumerate(dims):
            if cdim is None:
                ctx, bias, dbl = None, None, True
            elif multi and j > 1:
                ctx, bias, dbl = encoder_hidden_states_1, bias1, False
            else:
                ctx, bias, dbl = encoder_hidden_states, bias0, False
            x = transformer2d(sd, f"{prefix}.attentions.{k0 + j}", x, ctx, heads, groups, bias, dbl,
                              linear_proj, depth)
        return x

    # 1. time (+ class / FiLM) embedding  (models.py:218-256, :757-803)
    t = timestep if torch.is_tensor(timestep) else torch.tensor([timestep])
    t = t.reshape(-1).expand(sample.shape[0])
    t_emb = timestep_embedding(t, boc[0], cfg.get("flip_sin_to_cos", True), cfg.get("freq_shift", 0))
    emb = _lin(sd, "time_embedding.linear_2", F.silu(_lin(sd, "time_embedding.linear_1", t_emb)))
    if cfg.get("class_embed_type") is not None:
        cemb = _lin(sd, "class_embedding", class_labels)
        emb = torch.cat([emb, cemb], dim=-1) if cfg.get("class_embeddings_concat") else emb + cemb

    # 2-3. conv_in + down
    h = _conv(sd, "conv_in", sample)
    skips = [h]
    for i, btype in enumerate(cfg["down_block_types"]):
        has_attn = "CrossAttn" in btype
        for j in range(lpb):
            h = resnet(sd, f"down_blocks.{i}.resnets.{j}", h, emb, groups, eps)
            if has_attn:
                h = run_attn(f"down_blocks.{i}", j * len(ctx_pb[i]), h, ctx_pb[i], heads_pb[i])
            skips.append(h)
        if i < nb - 1:
            h = _conv(sd, f"down_blocks.{i}.downsamplers.0.conv", h, stride=2, padding=1)
            skips.append(h)

    # 4. mid
    h = resnet(sd, "mid_block.resnets.0", h, emb, groups, eps)
    h = run_attn("mid_block", 0, h, ctx_pb[-1], heads_pb[-1])
    h = resnet(sd, "mid_block.resnets.1", h, emb, groups, eps)

    # h-space hooks (models.py:334-343, :840-847)
    if replace_h_space is None:
        h_space = h.clone()
    else:
        h_space = replace_h_space
        h = replace_h_space.clone()
    if mid_block_additional_residual is not None:
        h = h + mid_block_additional_residual

    # 5. up
    extracted = {}
    rev_ch = list(reversed(range(nb)))
    for i, btype in enumerate(cfg["up_block_types"]):
        has_attn = "CrossAttn" in btype
        lvl = rev_ch[i]
        nres = lpb + 1
        res = skips[-nres:]
        skips = skips[:-nres]
        if replace_skip_conns is not None and replace_skip_conns.get(i):
            res = replace_skip_conns.get(i)
        if zero_out_resconns is not None:
            if (type(zero_out_resconns) is int and i >= (zero_out_resconns - 1)) or \
                    (type(zero_out_resconns) is list and i in zero_out_resconns):
                res = [torch.zeros_like(r) for r in res]
        extracted[i] = res
        res = list(res)
        for j in range(nres):
            h = torch.cat([h, res.pop()], dim=1)
            h = resnet(sd, f"up_blocks.{i}.resnets.{j}", h, emb, groups, eps)
            if has_attn:
                h = run_attn(f"up_blocks.{i}", j * len(ctx_pb[lvl]), h, ctx_pb[lvl], heads_pb[lvl])
        if i < nb - 1:
            # forward_upsample_size (models.py:186-188, :361-366): resize to the next skip's spatial size when the
            # input is not a multiple of 2^levels, else plain 2x (diffusers Upsample2D output_size semantics)
            tgt = skips[-1].shape[2:]
            if tuple(tgt) == (2 * h.shape[2], 2 * h.shape[3]):
                h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            else:
                h = F.interpolate(h, size=tuple(tgt), mode="nearest")
            h = _conv(sd, f"up_blocks.{i}.upsamplers.0.conv", h)

    # 6. out
    h = _conv(sd, "conv_out", F.silu(_gn(sd, "conv_norm_out", h, groups, eps)))
    return h, h_space, extracted


# ------------------------------------------------------------------ twin -> diffusers key map (pinning only)
def twin_to_diffusers(tsd, channel_mult, num_res_blocks, attn_levels):
    """Rename the in-tree twin's (openaimodel.UNetModel) state dict to diffusers names.

    Mirrors what diffusers' public AudioLDM conversion does; used only to pin this oracle on the twin.
    """
    out = {}

    def res(src, dst):
        for a, b in (("in_layers.0", "norm1"), ("in_layers.2", "conv1"), ("emb_layers.1", "time_emb_proj"),
                     ("out_layers.0", "norm2"), ("out_layers.3", "conv2"), ("skip_connection", "conv_shortcut")):
            for s in ("weight", "bias"):
                k = f"{src}.{a}.{s}"
                if k in tsd:
                    out[f"{dst}.{b}.{s}"] = tsd[k]

    def attn(src, dst):
        for k, v in tsd.items():
            if k.startswith(src + "."):
                out[dst + k[len(src):]] = v

    for s in ("weight", "bias"):
        out[f"time_embedding.linear_1.{s}"] = tsd[f"time_embed.0.{s}"]
        out[f"time_embedding.linear_2.{s}"] = tsd[f"time_embed.2.{s}"]
        if f"film_emb.{s}" in tsd:
            out[f"class_embedding.{s}"] = tsd[f"film_emb.{s}"]
        out[f"conv_in.{s}"] = tsd[f"input_blocks.0.0.{s}"]
        out[f"conv_norm_out.{s}"] = tsd[f"out.0.{s}"]
        out[f"conv_out.{s}"] = tsd[f"out.2.{s}"]
    n = 1
    nl = len(channel_mult)
    for lvl in range(nl):
        for j in range(num_res_blocks):
            res(f"input_blocks.{n}.0", f"down_blocks.{lvl}.resnets.{j}")
            if lvl in attn_levels:
                attn(f"input_blocks.{n}.1", f"down_blocks.{lvl}.attentions.{j}")
            n += 1
        if lvl != nl - 1:
            for s in ("weight", "bias"):
                out[f"down_blocks.{lvl}.downsamplers.0.conv.{s}"] = tsd[f"input_blocks.{n}.0.op.{s}"]
            n += 1
    res("middle_block.0", "mid_block.resnets.0")
    attn("middle_block.1", "mid_block.attentions.0")
    res("middle_block.2", "mid_block.resnets.1")
    n = 0
    for i, lvl in enumerate(reversed(range(nl))):
        for j in range(num_res_blocks + 1):
            res(f"output_blocks.{n}.0", f"up_blocks.{i}.resnets.{j}")
            k = 1
            if lvl in attn_levels:
                attn(f"output_blocks.{n}.1", f"up_blocks.{i}.attentions.{j}")
                k = 2
            if lvl != 0 and j == num_res_blocks:
                for s in ("weight", "bias"):
                    out[f"up_blocks.{i}.upsamplers.0.conv.{s}"] = tsd[f"output_blocks.{n}.{k}.conv.{s}"]
            n += 1
    return out

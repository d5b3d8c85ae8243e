"""Parameter inventories, seeded random init and checkpoint discovery.

Key names follow the checkpoints the reference loads through `from_pretrained`
(models.py:478, :556-564): diffusers' UNet2DConditionModel / AudioLDM2UNet2DConditionModel and
AutoencoderKL, transformers' SpeechT5HifiGan.  When a real model directory
exists on disk its safetensors are used; otherwise (this container: no network, no checkpoints)
weights are seeded random with the real architecture -- the same state dict feeds the HIP path
and the CPU oracle, so parity does not depend on the values.
"""
import glob
import json
import os

import torch


def _per_block(v, n):
    return list(v) if isinstance(v, (list, tuple)) else [v] * n


def ctx_dims_per_block(cfg):
    n = len(cfg["block_out_channels"])
    cad = cfg.get("cross_attention_dim")
    if isinstance(cad, (list, tuple)) and len(cad) and isinstance(cad[0], (list, tuple)):
        return [list(c) for c in cad], True
    return [[c] for c in _per_block(cad, n)], False


def unet_param_shapes(cfg):
    """name -> shape for every parameter of the (AudioLDM2)UNet2DConditionModel described by cfg."""
    boc = cfg["block_out_channels"]
    nb = len(boc)
    lpb = cfg.get("layers_per_block", 2)
    cin, cout = cfg["in_channels"], cfg["out_channels"]
    ted = boc[0] * 4
    emb_dim = ted
    ctx_pb, _ = ctx_dims_per_block(cfg)
    linear_proj = cfg.get("use_linear_projection", False)
    sh = {}

    def lin(p, o, i, bias=True):
        sh[p + ".weight"] = (o, i)
        if bias:
            sh[p + ".bias"] = (o,)

    def conv(p, o, i, k):
        sh[p + ".weight"] = (o, i, k, k)
        sh[p + ".bias"] = (o,)

    def norm(p, c):
        sh[p + ".weight"] = (c,)
        sh[p + ".bias"] = (c,)

    lin("time_embedding.linear_1", ted, boc[0])
    lin("time_embedding.linear_2", ted, ted)
    if cfg.get("class_embed_type") == "simple_projection":
        lin("class_embedding", ted, cfg["projection_class_embeddings_input_dim"])
        if cfg.get("class_embeddings_concat"):
            emb_dim = 2 * ted
    conv("conv_in", boc[0], cin, 3)

    def resnet(p, ci, co):
        norm(p + ".norm1", ci)
        conv(p + ".conv1", co, ci, 3)
        lin(p + ".time_emb_proj", co, emb_dim)
        norm(p + ".norm2", co)
        conv(p + ".conv2", co, co, 3)
        if ci != co:
            conv(p + ".conv_shortcut", co, ci, 1)

    def transformer(p, c, cdim):
        norm(p + ".norm", c)
        if linear_proj:
            lin(p + ".proj_in", c, c)
            lin(p + ".proj_out", c, c)
        else:
            conv(p + ".proj_in", c, c, 1)
            conv(p + ".proj_out", c, c, 1)
        b = p + ".transformer_blocks.0"
        for a, kd in (("attn1", c), ("attn2", c if cdim is None else cdim)):
            lin(f"{b}.{a}.to_q", c, c, bias=False)
            lin(f"{b}.{a}.to_k", c, kd, bias=False)
            lin(f"{b}.{a}.to_v", c, kd, bias=False)
            lin(f"{b}.{a}.to_out.0", c, c)
        for nrm in ("norm1", "norm2", "norm3"):
            norm(f"{b}.{nrm}", c)
        lin(f"{b}.ff.net.0.proj", 8 * c, c)
        lin(f"{b}.ff.net.2", c, 4 * c)

    ch = boc[0]
    skip_ch = [ch]
    for i, bt in enumerate(cfg["down_block_types"]):
        co = boc[i]
        for j in range(lpb):
            resnet(f"down_blocks.{i}.resnets.{j}", ch, co)
            ch = co
            if "CrossAttn" in bt:
                for k, cd in enumerate(ctx_pb[i]):
                    transformer(f"down_blocks.{i}.attentions.{j * len(ctx_pb[i]) + k}", co, cd)
            skip_ch.append(ch)
        if i < nb - 1:
            conv(f"down_blocks.{i}.downsamplers.0.conv", co, co, 3)
            skip_ch.append(ch)
    resnet("mid_block.resnets.0", ch, ch)
    for k, cd in enumerate(ctx_pb[-1]):
        transformer(f"mid_block.attentions.{k}", ch, cd)
    resnet("mid_block.resnets.1", ch, ch)
    for i, bt in enumerate(cfg["up_block_types"]):
        lvl = nb - 1 - i
        co = boc[lvl]
        for j in range(lpb + 1):
            sc = skip_ch.pop()
            resnet(f"up_blocks.{i}.resnets.{j}", ch + sc, co)
            ch = co
            if "CrossAttn" in bt:
                for k, cd in enumerate(ctx_pb[lvl]):
                    transformer(f"up_blocks.{i}.attentions.{j * len(ctx_pb[lvl]) + k}", co, cd)
        if i < nb - 1:
            conv(f"up_blocks.{i}.upsamplers.0.conv", co, co, 3)
    norm("conv_norm_out", boc[0])
    conv("conv_out", cout, boc[0], 3)
    return sh


def vae_param_shapes(cfg):
    boc = cfg["block_out_channels"]
    nb = len(boc)
    lpb = cfg.get("layers_per_block", 2)
    lc = cfg.get("latent_channels", 8)
    sh = {}

    def conv(p, o, i, k):
        sh[p + ".weight"] = (o, i, k, k)
        sh[p + ".bias"] = (o,)

    def norm(p, c):
        sh[p + ".weight"] = (c,)
        sh[p + ".bias"] = (c,)

    def res(p, ci, co):
        norm(p + ".norm1", ci)
        conv(p + ".conv1", co, ci, 3)
        norm(p + ".norm2", co)
        conv(p + ".conv2", co, co, 3)
        if ci != co:
            conv(p + ".conv_shortcut", co, ci, 1)

    def mid(p, c):
        res(p + ".resnets.0", c, c)
        a = p + ".attentions.0"
        norm(a + ".group_norm", c)
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            sh[f"{a}.{n}.weight"] = (c, c)
            sh[f"{a}.{n}.bias"] = (c,)
        res(p + ".resnets.1", c, c)

    conv("encoder.conv_in", boc[0], cfg.get("in_channels", 1), 3)
    ch = boc[0]
    for i in range(nb):
        for j in range(lpb):
            res(f"encoder.down_blocks.{i}.resnets.{j}", ch, boc[i])
            ch = boc[i]
        if i < nb - 1:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", ch, ch, 3)
    mid("encoder.mid_block", ch)
    norm("encoder.conv_norm_out", ch)
    conv("encoder.conv_out", 2 * lc, ch, 3)
    conv("quant_conv", 2 * lc, 2 * lc, 1)
    conv("post_quant_conv", lc, lc, 1)
    ch = boc[-1]
    conv("decoder.conv_in", ch, lc, 3)
    mid("decoder.mid_block", ch)
    for i in range(nb):
        co = boc[nb - 1 - i]
        for j in range(lpb + 1):
            res(f"decoder.up_blocks.{i}.resnets.{j}", ch, co)
            ch = co
        if i < nb - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", ch, ch, 3)
    norm("decoder.conv_norm_out", ch)
    conv("decoder.conv_out", cfg.get("out_channels", 1), ch, 3)
    return sh


def vocoder_param_shapes(cfg):
    sh = {"mean": (cfg["model_in_dim"],), "scale": (cfg["model_in_dim"],)}
    c0 = cfg["upsample_initial_channel"]
    sh["conv_pre.weight"] = (c0, cfg["model_in_dim"], 7)
    sh["conv_pre.bias"] = (c0,)
    nk = len(cfg["resblock_kernel_sizes"])
    ch = c0
    for i, (u, k) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
        co = c0 // (2 ** (i + 1))
        sh[f"upsampler.{i}.weight"] = (ch, co, k)
        sh[f"upsampler.{i}.bias"] = (co,)
        ch = co
        for j, rk in enumerate(cfg["resblock_kernel_sizes"]):
            for m in range(len(cfg["resblock_dilation_sizes"][j])):
                for cn in ("convs1", "convs2"):
                    sh[f"resblocks.{i * nk + j}.{cn}.{m}.weight"] = (ch, ch, rk)
                    sh[f"resblocks.{i * nk + j}.{cn}.{m}.bias"] = (ch,)
    sh["conv_post.weight"] = (1, ch, 7)
    sh["conv_post.bias"] = (1,)
    return sh


def dit_param_shapes(cfg):
    """diffusers StableAudioDiTModel parameter inventory (names as in its state dict)."""
    C = cfg["num_attention_heads"] * cfg["attention_head_dim"]
    KV = cfg["num_key_value_attention_heads"] * cfg["attention_head_dim"]
    cin, cout, dc = cfg["in_channels"], cfg["out_channels"], cfg["cross_attention_dim"]
    ff = cfg.get("ff_inner_dim") or 4 * C
    sh = {"time_proj.weight": (cfg["time_proj_dim"] // 2,),
          "timestep_proj.0.weight": (C, cfg["time_proj_dim"]), "timestep_proj.0.bias": (C,),
          "timestep_proj.2.weight": (C, C), "timestep_proj.2.bias": (C,),
          "global_proj.0.weight": (C, cfg["global_states_input_dim"]), "global_proj.2.weight": (C, C),
          "cross_attention_proj.0.weight": (dc, cfg["cross_attention_input_dim"]),
          "cross_attention_proj.2.weight": (dc, dc),
          "preprocess_conv.weight": (cin, cin, 1), "proj_in.weight": (C, cin)}
    for i in range(cfg["num_layers"]):
        p = f"transformer_blocks.{i}."
        for n in ("norm1", "norm2", "norm3"):
            sh[p + n + ".weight"] = (C,)
            sh[p + n + ".bias"] = (C,)
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            sh[p + f"attn1.{n}.weight"] = (C, C)
        sh[p + "attn2.to_q.weight"] = (C, C)
        sh[p + "attn2.to_k.weight"] = (KV, dc)
        sh[p + "attn2.to_v.weight"] = (KV, dc)
        sh[p + "attn2.to_out.0.weight"] = (C, C)
        sh[p + "ff.net.0.proj.weight"] = (2 * ff, C)
        sh[p + "ff.net.0.proj.bias"] = (2 * ff,)
        sh[p + "ff.net.2.weight"] = (C, ff)
        sh[p + "ff.net.2.bias"] = (C,)
    sh["proj_out.weight"] = (cout, C)
    sh["postprocess_conv.weight"] = (cout, cout, 1)
    return sh


def oobleck_param_shapes(cfg):
    """diffusers AutoencoderOobleck inventory AFTER weight-norm folding (`fold_weight_norm`): plain conv weights,
    Snake1d alpha/beta [1,C,1].  The encoder's last convolution emits the posterior's (mean | scale) pair, i.e.
    2 * decoder_input_channels channels -- which is `encoder_hidden_size` (128 = 2 * 64) in the released config."""
    sh = {}
    hid, ca = cfg["encoder_hidden_size"], cfg["audio_channels"]
    mult = [1] + list(cfg["channel_multiples"])
    ratios = list(cfg["downsampling_ratios"])

    def conv(p, o, i, k, bias=True):
        sh[p + ".weight"] = (o, i, k)
        if bias:
            sh[p + ".bias"] = (o,)

    def snake(p, c):
        sh[p + ".alpha"] = (1, c, 1)
        sh[p + ".beta"] = (1, c, 1)

    def res(p, c):
        snake(p + ".snake1", c)
        conv(p + ".conv1", c, c, 7)
        snake(p + ".snake2", c)
        conv(p + ".conv2", c, c, 1)

    conv("encoder.conv1", hid, ca, 7)
    for i, st in enumerate(ratios):
        ci, co = hid * mult[i], hid * mult[i + 1]
        for j in range(3):
            res(f"encoder.block.{i}.res_unit{j + 1}", ci)
        snake(f"encoder.block.{i}.snake1", ci)
        conv(f"encoder.block.{i}.conv1", co, ci, 2 * st)
    snake("encoder.snake1", hid * mult[-1])
    conv("encoder.conv2", 2 * cfg["decoder_input_channels"], hid * mult[-1], 3)
    dch = cfg["decoder_channels"]
    conv("decoder.conv1", dch * mult[-1], cfg["decoder_input_channels"], 7)
    for i, st in enumerate(ratios[::-1]):
        ci, co = dch * mult[len(ratios) - i], dch * mult[len(ratios) - i - 1]
        snake(f"decoder.block.{i}.snake1", ci)
        sh[f"decoder.block.{i}.conv_t1.weight"] = (ci, co, 2 * st)          # ConvTranspose1d layout [Cin, Cout, k]
        sh[f"decoder.block.{i}.conv_t1.bias"] = (co,)
        for j in range(3):
            res(f"decoder.block.{i}.res_unit{j + 1}", co)
    snake("decoder.snake1", dch)
    conv("decoder.conv2", ca, dch, 7, bias=False)
    return sh


def projection_param_shapes(cfg):
    """diffusers StableAudioProjectionModel: identity text projection when the dims agree, two number conditioners."""
    sh = {}
    d, k = cfg["conditioning_dim"], cfg["number_embedding_internal_dim"]
    if cfg["text_encoder_dim"] != d:
        sh["text_projection.weight"] = (d, cfg["text_encoder_dim"])
        sh["text_projection.bias"] = (d,)
    for n in ("start_number_conditioner", "end_number_conditioner"):
        sh[n + ".time_positional_embedding.0.weights"] = (k // 2,)
        sh[n + ".time_positional_embedding.1.weight"] = (d, k + 1)
        sh[n + ".time_positional_embedding.1.bias"] = (d,)
    return sh


def random_state_dict(shapes, seed=0, gain=1.0):
    """Fan-in scaled init (keeps activations O(1) so parity tests exercise real dynamic range);
    norm weights 1 + 0.1*randn, biases 0.02*randn.  Deterministic in (shapes order, seed)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shp in shapes.items():
        if name in ("mean",):
            sd[name] = torch.zeros(shp)
        elif name in ("scale",):
            sd[name] = torch.ones(shp)
        elif name.endswith("time_proj.weight") or name.endswith(".weights"):     # learned Fourier frequencies
            sd[name] = torch.randn(shp, generator=g)
        elif len(shp) == 1:
            is_norm_w = name.endswith(".weight")
            t = torch.randn(shp, generator=g)
            sd[name] = (1.0 + 0.1 * t) if is_norm_w else 0.02 * t
        else:
            fan_in = 1
            for s in shp[1:]:
                fan_in *= s
            if "upsampler" in name or "conv_t1" in name:   # ConvTranspose1d [Cin, Cout, k]: fan-in = Cin*k/stride-ish
                fan_in = shp[0] * max(1, shp[2] // 4)
            if name.endswith(".alpha") or name.endswith(".beta"):          # Snake1d log-scale parameters
                sd[name] = 0.3 * torch.randn(shp, generator=g)
                continue
            sd[name] = torch.randn(shp, generator=g) * (gain / fan_in ** 0.5)
    return sd


def count_params(shapes):
    n = 0
    for s in shapes.values():
        k = 1
        for d in s:
            k *= d
        n += k
    return n


# ------------------------------------------------------------------------------------------------
def find_checkpoint(model_id):
    """Return a local HF snapshot directory for model_id, or None.  Never touches the network."""
    cands = []
    if os.path.isdir(model_id):
        cands.append(model_id)
    home = os.environ.get("HF_HOME", os.path.join(os.path.expanduser("~"), ".cache", "huggingface"))
    cands += sorted(glob.glob(os.path.join(home, "hub", "models--" + model_id.replace("/", "--"), "snapshots", "*")))
    for c in cands:
        if os.path.exists(os.path.join(c, "unet", "config.json")) or \
                os.path.exists(os.path.join(c, "transformer", "config.json")):       # Stable Audio: DiT, no U-Net
            return c
    return None


def _load_component(root, sub, fname_candidates):
    from safetensors.torch import load_file
    for f in fname_candidates:
        p = os.path.join(root, sub, f)
        if os.path.exists(p):
            if p.endswith(".safetensors"):
                return load_file(p)
            return torch.load(p, map_location="cpu", weights_only=True)
    raise FileNotFoundError(f"no weights for {sub} under {root}")


def fold_weight_norm(sd):
    """Fold weight_g/weight_v (or parametrizations.weight.original0/1) pairs into plain weights."""
    out = dict(sd)
    for k in list(sd):
        for g_s, v_s in ((".weight_g", ".weight_v"), (".parametrizations.weight.original0",
                                                       ".parametrizations.weight.original1")):
            if k.endswith(g_s):
                base = k[: -len(g_s)]
                g, v = sd[k], sd[base + v_s]
                nrm = v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
                out[base + ".weight"] = v * (g / nrm)
                out.pop(k)
                out.pop(base + v_s)
    return out


def load_checkpoint(root):
    """Read configs + weights of a local diffusers pipeline directory."""
    res = {}
    for sub, names in (("unet", ["diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.bin"]),
                       ("vae", ["diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.bin"]),
                       ("vocoder", ["model.safetensors", "pytorch_model.bin"])):
        with open(os.path.join(root, sub, "config.json")) as f:
            cfg = json.load(f)
        sd = _load_component(root, sub, names)
        if sub == "vocoder":
            sd = fold_weight_norm(sd)
        res[sub] = (cfg, {k: v.float() for k, v in sd.items()})
    sp = os.path.join(root, "scheduler", "scheduler_config.json")
    if os.path.exists(sp):
        with open(sp) as f:
            res["scheduler"] = json.load(f)
    return res


def load_stable_audio_checkpoint(root):
    """Configs + weights of a local StableAudioPipeline directory: transformer/ (DiT), vae/ (Oobleck, weight norm
    folded), projection_model/, scheduler/."""
    names = ["diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.bin"]
    res = {}
    for sub in ("transformer", "vae", "projection_model"):
        with open(os.path.join(root, sub, "config.json")) as f:
            cfg = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
        sd = _load_component(root, sub, names)
        if sub == "vae":
            sd = fold_weight_norm(sd)
        res[sub] = (cfg, {k: v.float() for k, v in sd.items()})
    sp = os.path.join(root, "scheduler", "scheduler_config.json")
    if os.path.exists(sp):
        with open(sp) as f:
            res["scheduler"] = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
    return res

"""Checkpoint discovery / loading (models.load_model's disk path, reference models.py:1357-1374 -> from_pretrained)
and the parameter inventories, without a GPU and without network."""
import json
import os

import pytest
import torch

from audioeditingcode_amd import configs, weights


def test_vocoder_inventory_matches_the_transformers_class():
    """The reference's vocoder IS transformers.SpeechT5HifiGan (models.py:505-509): names and shapes of our inventory
    must be that class's state dict for the same config."""
    from transformers import SpeechT5HifiGan, SpeechT5HifiGanConfig
    cfg = configs.FAMILIES["audioldm2"]["vocoder"]
    keys = ("model_in_dim", "sampling_rate", "upsample_initial_channel", "upsample_rates", "upsample_kernel_sizes",
            "resblock_kernel_sizes", "resblock_dilation_sizes", "leaky_relu_slope", "normalize_before")
    hf = SpeechT5HifiGan(SpeechT5HifiGanConfig(**{k: cfg[k] for k in keys if k in cfg}))
    ref = {k: tuple(v.shape) for k, v in hf.state_dict().items()}
    ours = {k: tuple(v) for k, v in weights.vocoder_param_shapes(cfg).items()}
    assert ours == ref


def test_fold_weight_norm_equals_torch_weight_norm():
    conv = torch.nn.utils.weight_norm(torch.nn.Conv1d(6, 4, 3))
    sd = {"c." + k: v.detach() for k, v in conv.state_dict().items()}
    assert "c.weight_g" in sd and "c.weight_v" in sd
    out = weights.fold_weight_norm(sd)
    assert set(out) == {"c.weight", "c.bias"}
    x = torch.randn(2, 6, 9)
    torch.testing.assert_close(torch.nn.functional.conv1d(x, out["c.weight"], out["c.bias"]), conv(x))
    par = torch.nn.utils.parametrizations.weight_norm(torch.nn.Conv1d(6, 4, 3))
    sd2 = {"c." + k: v.detach() for k, v in par.state_dict().items()}
    out2 = weights.fold_weight_norm(sd2)
    torch.testing.assert_close(torch.nn.functional.conv1d(x, out2["c.weight"], out2["c.bias"]), par(x))


@pytest.mark.parametrize("kind", ["audioldm2", "audioldm", "tango"])
def test_local_snapshot_directory_roundtrip(tmp_path, kind):
    """A diffusers-layout snapshot on disk ({unet,vae,vocoder}/config.json + safetensors, scheduler config) is found
    and read back key for key; the vocoder's weight-norm pairs are folded."""
    from safetensors.torch import save_file
    fam = configs.tiny_family(kind)
    root = tmp_path / "snap"
    sds = {}
    for sub, shapes, fname in (("unet", weights.unet_param_shapes(fam["unet"]), "diffusion_pytorch_model.safetensors"),
                               ("vae", weights.vae_param_shapes(fam["vae"]), "diffusion_pytorch_model.safetensors"),
                               ("vocoder", weights.vocoder_param_shapes(fam["vocoder"]), "model.safetensors")):
        os.makedirs(root / sub)
        sd = weights.random_state_dict(shapes, seed=3)
        sds[sub] = sd
        on_disk = dict(sd)
        if sub == "vocoder":                       # store conv_pre the way a weight-normed checkpoint does
            w = on_disk.pop("conv_pre.weight")
            g = w.flatten(1).norm(dim=1).view(-1, 1, 1)
            on_disk["conv_pre.weight_g"], on_disk["conv_pre.weight_v"] = g, w * 2.0      # v is scale-free
        save_file({k: v.contiguous() for k, v in on_disk.items()}, str(root / sub / fname))
        with open(root / sub / "config.json", "w") as f:
            json.dump(fam[sub], f)
    os.makedirs(root / "scheduler")
    with open(root / "scheduler" / "scheduler_config.json", "w") as f:
        json.dump(fam["scheduler"], f)
    assert weights.find_checkpoint(str(root)) == str(root)
    assert weights.find_checkpoint(str(tmp_path / "nope")) is None
    comp = weights.load_checkpoint(str(root))
    for sub in ("unet", "vae", "vocoder"):
        cfg, sd = comp[sub]
        assert cfg == json.loads(json.dumps(fam[sub]))
        assert set(sd) == set(sds[sub])
        for k in sd:
            torch.testing.assert_close(sd[k], sds[sub][k], rtol=1e-6, atol=1e-6)
    assert comp["scheduler"] == json.loads(json.dumps(fam["scheduler"]))


def test_hf_cache_layout_is_searched(tmp_path, monkeypatch):
    snap = tmp_path / "hub" / "models--cvssp--audioldm2" / "snapshots" / "abc123"
    os.makedirs(snap / "unet")
    (snap / "unet" / "config.json").write_text("{}")
    monkeypatch.setenv("HF_HOME", str(tmp_path))
    assert weights.find_checkpoint("cvssp/audioldm2") == str(snap)
    assert weights.find_checkpoint("cvssp/audioldm2-music") is None

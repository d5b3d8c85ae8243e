"""GPU parity of the compiled U-Net tape against the CPU oracle (same seeded weights, same inputs)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from audioeditingcode_amd import configs, weights              # noqa: E402
from audioeditingcode_amd.unet import UNetEngine                # noqa: E402
from oracle import unet as ounet                                # noqa: E402
from conftest import oracle_run                                 # noqa: E402

DEV = "cuda:0"


def _run_case(fam, B, H, W, L0, L1, t, seed=0, use_ehs=True, heads=None, want_folded=0, oracle_key=None):
    cfg = fam["unet"]
    if heads is not None:
        cfg["attention_head_dim"] = heads
    sd = weights.random_state_dict(weights.unet_param_shapes(cfg), seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(B, cfg["in_channels"], H, W, generator=g)
    kind = fam["ctx"]["kind"]
    kw, okw = {}, {}
    eng = UNetEngine(cfg, sd, DEV, B, H, W, ctx_len0=L0, ctx_len1=L1, use_ehs=use_ehs)
    if kind == "audioldm2":
        e0 = torch.randn(B, L0, fam["ctx"]["gpt2_dim"], generator=g)
        e1 = torch.randn(B, L1, fam["ctx"]["t5_dim"], generator=g)
        m1 = torch.ones(B, L1)
        m1[0, L1 // 2:] = 0                                  # ragged prompt lengths inside one batch
        eng.set_conditioning(ehs0=e0, ehs1=e1, bias1=(1 - m1) * -10000.0)
        okw = dict(encoder_hidden_states=e0, encoder_hidden_states_1=e1, encoder_attention_mask_1=m1)
    elif kind == "audioldm":
        cl = torch.nn.functional.normalize(torch.randn(B, fam["ctx"]["clap_dim"], generator=g), dim=-1)
        eng.set_conditioning(class_labels=cl)
        okw = dict(class_labels=cl)
    else:
        e0 = torch.randn(B, L0, fam["ctx"]["t5_dim"], generator=g)
        m0 = torch.ones(B, L0)
        m0[-1, L0 - 2:] = 0
        eng.set_conditioning(ehs0=e0, bias0=(1 - m0) * -10000.0)
        okw = dict(encoder_hidden_states=e0, encoder_attention_mask=m0)
    eng.x_in.copy_(x.permute(0, 2, 3, 1))
    eng.set_timestep(t)
    eng.forward()
    torch.cuda.synchronize()
    assert sum(1 for o in eng.tape.ops if o.code == 1 and o.i[36] > 0) >= want_folded
    got = eng.eps.cpu().permute(0, 3, 1, 2)
    hs = eng.h_space.cpu().permute(0, 3, 1, 2)
    oracle = lambda: ounet.unet_forward(cfg, sd, x, torch.tensor(t), **okw)[:2]                  # noqa: E731
    # (oracle_key: the CPU forward of a full-size model is a committed oracle run, tests/conftest.py oracle_run)
    ref, ref_h = oracle_run(oracle_key, oracle, x) if oracle_key else oracle()
    return got, ref, hs, ref_h, eng


@pytest.mark.parametrize("kind,use_ehs", [("audioldm2", True), ("audioldm", False), ("tango", True)])
def test_tiny_unet_matches_oracle(kind, use_ehs):
    fam = configs.tiny_family(kind)
    got, ref, hs, ref_h, _ = _run_case(fam, B=2, H=32, W=16, L0=8 if kind == "audioldm2" else 6, L1=5, t=501,
                                       use_ehs=use_ehs)
    scale = ref.abs().max().item()
    assert (hs - ref_h).abs().max().item() < 2e-4 * max(1.0, ref_h.abs().max().item())
    assert (got - ref).abs().max().item() < 2e-4 * max(1.0, scale), ((got - ref).abs().max().item(), scale)


def test_full_audioldm2_unet_matches_oracle():
    """BASELINE config 2 shape: AudioLDM2 U-Net, latent 8x256x16, cond+uncond batched (B=2)."""
    fam = configs.FAMILIES["audioldm2"]
    got, ref, hs, ref_h, eng = _run_case(fam, B=2, H=256, W=16, L0=8, L1=16, t=996)
    rel = ((got - ref).norm() / ref.norm()).item()
    assert rel < 1e-4, rel
    assert abs(eng.tape.flops / 2 / 1e9 - 172.4) < 3.0, eng.tape.flops / 2 / 1e9     # 172.4 GF / sample


def test_unet_with_sizes_not_multiple_of_8_uses_upsample_size_path():
    """Latent 36x12 -> 18x6 -> 9x3 -> 5x2: odd levels force the reference's forward_upsample_size path."""
    fam = configs.tiny_family("audioldm2")
    got, ref, hs, ref_h, _ = _run_case(fam, B=2, H=36, W=12, L0=8, L1=5, t=301, seed=5)
    assert ref_h.shape[-2:] == (5, 2)
    assert (got - ref).abs().max().item() < 2e-4 * max(1.0, ref.abs().max().item())


def test_full_audioldm_s_unet_matches_oracle():
    """BASELINE config 1's model at its real size: AudioLDM-S U-Net (185 M parameters, CLAP FiLM conditioning through the
    concatenated class embedding, attn2 degenerating to self-attention), latent 8x256x16, cond+uncond batched."""
    fam = configs.get_family("cvssp/audioldm-s-full")
    got, ref, hs, ref_h, eng = _run_case(fam, B=2, H=256, W=16, L0=0, L1=0, t=981, use_ehs=False, oracle_key="unet_full_audioldm_s")
    rel = ((got - ref).norm() / ref.norm()).item()
    assert rel < 1e-4, rel
    assert ((hs - ref_h).norm() / ref_h.norm()).item() < 1e-4
    n_params = sum(v.numel() for v in weights.random_state_dict(weights.unet_param_shapes(fam["unet"]), seed=0).values())
    assert abs(n_params / 1e6 - 185.0) < 1.0, n_params                      # in-tree twin: 185.0 M


def test_full_tango_unet_matches_oracle():
    """The TANGO wrapper's U-Net at its real size (models.py:396-472: SD-2.1-style UNet2DConditionModel, 865.9 M parameters,
    block widths 320 / 640 / 1280 / 1280, linear projections, 64-wide heads, T5 cross-attention with the additive -10000
    key mask, self-attention over all 4096 latent tokens at level 0), latent 8x256x16, cond+uncond batched."""
    fam = configs.FAMILIES["tango"]
    got, ref, hs, ref_h, eng = _run_case(fam, B=2, H=256, W=16, L0=16, L1=0, t=501, oracle_key="unet_full_tango")
    rel = ((got - ref).norm() / ref.norm()).item()
    assert rel < 1e-4, rel
    assert ((hs - ref_h).norm() / ref_h.norm()).item() < 1e-4
    n_params = sum(v.numel() for v in weights.random_state_dict(weights.unet_param_shapes(fam["unet"]), seed=0).values())
    assert abs(n_params / 1e6 - 865.9) < 1.0, n_params


@pytest.mark.parametrize("kind,L0,L1", [("audioldm2", 8, 16), ("tango", 16, 0)])
def test_folded_cross_attention_unet_matches_oracle(kind, L0, L1):
    """the folded cross-attention (two skinny GEMMs with per-batch weights and a grouped softmax, lin_gemm kernels) inside
    a whole tiny U-Net against the oracle's q-proj -> softmax(QK^T)V -> to_out, ragged key masks included"""
    fam = configs.tiny_family(kind)
    # (2 heads: head widths 16..64 are the attention kernels' supported widths; 2 heads x 16 keys = one 32-column tile)
    got, ref, hs, ref_h, _ = _run_case(fam, B=2, H=32, W=16, L0=L0, L1=L1, t=401, want_folded=4)
    assert (hs - ref_h).abs().max().item() < 2e-4 * max(1.0, ref_h.abs().max().item())
    assert (got - ref).abs().max().item() < 2e-4 * max(1.0, ref.abs().max().item())


def test_tape_image_runs_a_forward_without_the_python_graph_compiler(tmp_path):
    """The model-level boundary (include/aed.h, tape images): a U-Net engine's tapes are exported ahead of time
    (image.export_image); then (1) the seven aed_image_* entry points, driven through ctypes the way a C host would, and
    (2) a plain C program (examples/image_host.c, built by __graft_entry__.build(), no Python / no torch in its process)
    reproduce the engine's own forward bit for bit -- for the snapshot inputs and for new inputs copied in by name."""
    import ctypes
    import os
    import subprocess

    import numpy as np

    from audioeditingcode_amd import _lib as L
    from audioeditingcode_amd.image import Image, export_image
    fam = configs.tiny_family("audioldm2")
    cfg = fam["unet"]
    sd = weights.random_state_dict(weights.unet_param_shapes(cfg), seed=0)
    ts = torch.tensor([501, 301], dtype=torch.int64, device=DEV)
    state = torch.zeros(4, dtype=torch.int32, device=DEV)
    eng = UNetEngine(cfg, sd, DEV, 2, 32, 16, ctx_len0=8, ctx_len1=8, timesteps_dev=ts, state_dev=state)
    g = torch.Generator().manual_seed(1)
    mk = lambda: (torch.randn(2, 8, 48, generator=g), torch.randn(2, 8, 64, generator=g), torch.zeros(2, 8),   # noqa: E731
                  torch.randn(2, 32, 16, 8, generator=g))

    def engine_forward(e0, e1, b1, x):
        eng.set_conditioning(ehs0=e0, ehs1=e1, bias1=b1)
        eng.x_in.copy_(x)
        eng.forward()
        torch.cuda.synchronize()
        return eng.eps.cpu().clone()
    first, second = mk(), mk()
    ref1 = engine_forward(*first)
    path = str(tmp_path / "unet_tiny.aedimg")
    info = export_image(path, {"context": eng.ctx_tape, "forward": eng.tape},
                        dict(x_in=eng.x_in, eps=eng.eps, ehs0=eng.ehs0, ehs1=eng.ehs1, bias1=eng.bias1, timesteps=ts,
                             state=state), scratch=[eng.eps, eng.h_space])
    assert info["programs"] == ["context", "forward"]
    ref2 = engine_forward(*second)                 # the engine moves on; the image keeps the snapshot of `first`
    # ---- (1) the C ABI through ctypes
    im = Image(path)
    st = torch.cuda.Stream()
    sp = ctypes.c_void_p(st.cuda_stream)
    im.run("context", sp)
    im.run("forward", sp)
    got = im.copy_out("eps", torch.empty_like(ref1), sp)
    assert torch.equal(got, ref1)
    for name, t in zip(("ehs0", "ehs1", "bias1", "x_in"), second):
        im.copy_in(name, t.float(), sp)
    im.run("context", sp)
    im.run("forward", sp)
    assert torch.equal(im.copy_out("eps", torch.empty_like(ref2), sp), ref2)
    im.close()
    # ---- (2) a host without Python
    pkg = os.path.dirname(L.LIB_PATH)
    host = os.path.join(pkg, "aed_image_host")
    if not os.path.exists(host):                   # normally built by __graft_entry__.build(); plain gcc, seconds
        root = os.path.dirname(pkg)
        subprocess.check_call(["gcc", "-O2", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "image_host.c"),
                               "-L" + pkg, "-laed", "-Wl,-rpath,$ORIGIN", "-o", host])
    files = []
    for name, t in zip(("ehs0", "ehs1", "bias1", "x_in"), second):
        f = str(tmp_path / f"{name}.bin")
        t.float().contiguous().numpy().tofile(f)
        files.append(f"{name}={f}")
    out = str(tmp_path / "eps.bin")
    r = subprocess.run([host, path, out, "3", *files], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    eps = torch.from_numpy(np.fromfile(out, dtype=np.float32)).reshape(ref2.shape)
    assert torch.equal(eps, ref2), float((eps - ref2).abs().max())


def test_wrapper_unet_forward_matches_the_references_inline_forward_graphs():
    """SURVEY A8 at the graph level, on the GPU and through the wrapper API: `unet_forward` of the three wrappers -- plain,
    replace_h_space, mid_block_additional_residual, replace_skip_conns, zero_out_resconns (int / list) and an input size that
    forces forward_upsample_size -- against tests/golden/unet_forward_graph.npz, which the reference's OWN inline forward
    graphs wrote (models.py:160-393 PipelineWrapper.unet_forward, :691-899 AudioLDM2Wrapper.unet_forward, executed by
    oracle/make_golden.py unet_graph on stand-in diffusers blocks).  eps, h_space and every extracted residual."""
    import unet_graph_cases as ugc
    from audioeditingcode_amd import models
    g = ugc.load()
    rel = lambda a, b: float((a.cpu() - b).norm() / b.norm().clamp_min(1e-12))                       # noqa: E731
    n = 0
    for fam in ugc.FAMILIES:
        m = models.load_model(f"tiny/{fam}", DEV, 10, seed=int(g[f"{fam}.seed"]))
        for fam_, size, hook in ugc.cases():
            if fam_ != fam:
                continue
            x, t, cond = ugc.inputs(g, fam, size)
            cond = {k: (None if v is None else v.to(DEV)) for k, v in cond.items()}
            hooks = ugc.hook_kwargs(g, fam, hook)
            out, h, skips = m.unet_forward(x.to(DEV), torch.tensor(t), **cond, **hooks)
            skips = {i: [s.clone() for s in skips[i]] for i in range(4)}
            torch.cuda.synchronize()
            ugc.check(fam, size, hook, (out.sample, h, skips), ugc.expected(g, fam, size, hook), 1e-4, rel)
            n += 1
    assert n == 24

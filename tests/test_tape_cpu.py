"""The product's HOST logic executed on CPU: tapes built by the product's own builders on device="cpu" are run by the
oracle's tape interpreter (oracle/tape_interp.py, an independent plain-torch statement of every opcode) instead of
libaed.so, and compared with the oracle models / reference fixtures.  This checks graph wiring, weight packing, the
algebraic LayerNorm fold, skip/concat strides, padding masks, the device-indexed loops and the timestep-batched
inversion without a GPU.  (The kernels themselves are proven only by the `-m gpu` tests.)"""
import pytest
import torch

from audioeditingcode_amd import configs, weights
from audioeditingcode_amd.tape import Tape
from audioeditingcode_amd.unet import UNetEngine
from oracle import tape_interp
from oracle import unet as ounet


@pytest.fixture(autouse=True)
def _cpu_executor(monkeypatch):
    monkeypatch.setattr(Tape, "run", tape_interp.run_tape)


def _unet_case(kind, B=2, H=16, W=16, L0=6, L1=5, t=501, seed=0, use_ehs=True, heads=None, want_folded=0):
    fam = configs.tiny_family(kind)
    cfg = fam["unet"]
    if heads is not None:
        cfg["attention_head_dim"] = heads
    sd = weights.random_state_dict(weights.unet_param_shapes(cfg), seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(B, cfg["in_channels"], H, W, generator=g)
    eng = UNetEngine(cfg, sd, "cpu", B, H, W, ctx_len0=L0, ctx_len1=L1, use_ehs=use_ehs)
    ctx = fam["ctx"]
    if ctx["kind"] == "audioldm2":
        e0 = torch.randn(B, L0, ctx["gpt2_dim"], generator=g)
        e1 = torch.randn(B, L1, ctx["t5_dim"], generator=g)
        m1 = torch.ones(B, L1)
        m1[0, L1 // 2:] = 0
        eng.set_conditioning(ehs0=e0, ehs1=e1, bias1=(1 - m1) * -10000.0)
        okw = dict(encoder_hidden_states=e0, encoder_hidden_states_1=e1, encoder_attention_mask_1=m1)
    elif ctx["kind"] == "audioldm":
        cl = torch.nn.functional.normalize(torch.randn(B, ctx["clap_dim"], generator=g), dim=-1)
        eng.set_conditioning(class_labels=cl)
        okw = dict(class_labels=cl)
    else:
        e0 = torch.randn(B, L0, ctx["t5_dim"], generator=g)
        m0 = torch.ones(B, L0)
        m0[-1, L0 - 2:] = 0
        eng.set_conditioning(ehs0=e0, bias0=(1 - m0) * -10000.0)
        okw = dict(encoder_hidden_states=e0, encoder_attention_mask=m0)
    eng.x_in.copy_(x.permute(0, 2, 3, 1))
    eng.set_timestep(t)
    eng.forward()
    n_fold = sum(1 for o in eng.tape.ops if o.code == 1 and o.i[36] > 0)
    assert n_fold >= want_folded, (n_fold, want_folded)
    ref, ref_h, _ = ounet.unet_forward(cfg, sd, x, torch.tensor(t), **okw)
    return eng.eps.permute(0, 3, 1, 2), ref, eng.h_space.permute(0, 3, 1, 2), ref_h


@pytest.mark.parametrize("kind,use_ehs", [("audioldm2", True), ("audioldm", False), ("tango", True)])
def test_unet_tape_wiring_matches_oracle(kind, use_ehs):
    got, ref, hs, ref_h = _unet_case(kind, use_ehs=use_ehs)
    assert (hs - ref_h).abs().max().item() < 1e-4 * max(1.0, ref_h.abs().max().item())
    assert (got - ref).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("kind,L0,L1", [("audioldm2", 8, 16), ("tango", 16, 0)])
def test_folded_cross_attention_matches_oracle(kind, L0, L1):
    """Cross-attention over the short, per-prompt-constant text keys as two skinny GEMMs (scores + grouped softmax,
    then P.VO + bias + residual) with per-head operands built on the context tape: exact algebra, so the U-Net output
    still matches the oracle's q-projection -> softmax(QK^T) V -> to_out; ragged key masks included."""
    got, ref, hs, ref_h = _unet_case(kind, H=32, W=16, L0=L0, L1=L1, heads=4, want_folded=4)
    assert (hs - ref_h).abs().max().item() < 1e-4 * max(1.0, ref_h.abs().max().item())
    assert (got - ref).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())


# ------------------------------------------------------------------------------------------------ loops on CPU
from audioeditingcode_amd.editing import Conditioning, EditEngine          # noqa: E402
from audioeditingcode_amd.scheduler import DDIMScheduler                   # noqa: E402
from oracle import loops as oloops                                         # noqa: E402
from oracle.scheduler import OracleDDIMScheduler                           # noqa: E402

LH, LW = 16, 16


def rel(a, b):
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


@pytest.fixture
def cpu_loops(monkeypatch):
    """EditEngine without HIP: graphs become plain Python loops, the one direct C-ABI call becomes torch math."""
    def run_graph(self, body, steps, use_graph=True, plan=None):
        for _ in range(steps):
            body()

    def sample_xts(self, x0, noise=None, generator=None):
        s = self.sched
        T = s.num_inference_steps
        x0 = x0.float()
        if noise is None:
            noise = torch.stack([torch.randn(x0.shape, generator=generator, dtype=torch.float32) for _ in range(T)])
        ts, abar = s.timesteps.cpu(), s.alphas_cumprod
        t_rows = torch.stack([ts[T - (r + 1)] for r in range(T)])
        sa, sb = abar[t_rows] ** 0.5, ((1 - abar) ** 0.5)[t_rows]
        shape = (T, *[1] * x0.dim())
        return torch.cat([x0[None], x0[None] * sa.reshape(shape) + noise * sb.reshape(shape)])
    monkeypatch.setattr(EditEngine, "_run_graph", run_graph)
    monkeypatch.setattr(EditEngine, "sample_xts", sample_xts)


def _loop_setup(kind="audioldm2", T=8, seed=0):
    fam = configs.tiny_family(kind)
    cfg = fam["unet"]
    sd = weights.random_state_dict(weights.unet_param_shapes(cfg), seed=seed)
    g = torch.Generator().manual_seed(seed + 5)
    if kind == "audioldm2":
        mk = lambda L1: dict(encoder_hidden_states=torch.randn(1, 8, 48, generator=g),          # noqa: E731
                             encoder_hidden_states_1=torch.randn(1, L1, 64, generator=g),
                             encoder_attention_mask_1=torch.ones(1, L1))
        conds = dict(src=mk(6), tgt=mk(9), unc=mk(1))
        to_c = lambda d: Conditioning(ehs0=d["encoder_hidden_states"], ehs1=d["encoder_hidden_states_1"],  # noqa: E731
                                      mask1=d["encoder_attention_mask_1"])
    else:
        mk = lambda: dict(class_labels=torch.nn.functional.normalize(torch.randn(1, 24, generator=g), dim=-1))  # noqa: E731
        conds = dict(src=mk(), tgt=mk(), unc=mk())
        to_c = lambda d: Conditioning(class_labels=d["class_labels"])                                      # noqa: E731
    sched, osched = DDIMScheduler(), OracleDDIMScheduler()
    sched.set_timesteps(T)
    osched.set_timesteps(T)

    def unet_fn(x, t, cond):
        kw = {k: (v.expand(x.shape[0], *v.shape[1:]) if torch.is_tensor(v) else v) for k, v in cond.items()}
        return ounet.unet_forward(cfg, sd, x, t, **kw)[0]
    ow = oloops.OracleWrapper(osched, unet_fn)
    eng = EditEngine(cfg, sd, sched, "cpu", LH, LW, kind)
    x0 = torch.randn(1, 8, LH, LW, generator=g) * 0.8
    return eng, ow, conds, to_c, x0


@pytest.mark.parametrize("kind", ["audioldm2", "audioldm"])
def test_device_resident_loops_match_oracle_on_cpu(cpu_loops, kind):
    """invert + edit: device-indexed copy2d / step ops / advance, coefficient tables, cfg, ragged context padding."""
    T, tstart = 8, 5
    eng, ow, conds, to_c, x0 = _loop_setup(kind, T)
    xts0 = ow.sample_xts_from_x0(x0, T, generator=torch.Generator().manual_seed(3))
    _, zs_o, xts_o = oloops.invert(ow, x0, conds["src"], conds["unc"], [3.0], T, eta=1.0, xts=xts0.clone())
    w_o = oloops.edit(ow, xts_o, torch.tensor([tstart]), conds["tgt"], conds["unc"], [12.0], zs_o[:tstart], eta=1.0)
    zs, xts = eng.invert(x0, to_c(conds["src"]), to_c(conds["unc"]), [3.0], eta=1.0, xts=xts0.unsqueeze(1))
    w = eng.edit(xts, zs, tstart, to_c(conds["tgt"]), to_c(conds["unc"]), [12.0], eta=1.0)
    zs_n, xts_n, w_n = eng.to_nchw(zs)[:, 0], eng.to_nchw(xts)[:, 0], eng.to_nchw(w)
    assert torch.equal(zs_n[0], torch.zeros_like(zs_n[0]))
    assert rel(xts_n[1:], xts_o[1:]) < 1e-5
    assert rel(zs_n[1:], zs_o[1:]) < 1e-3, rel(zs_n[1:], zs_o[1:])
    assert rel(w_n, w_o) < 1e-3, rel(w_n, w_o)


def test_per_step_eta_lists_on_the_device_loops_on_cpu(cpu_loops):
    """`etas` as a per-step list (the reference indexes it as etas[idx], inversion_utils.py:124 / :302): one coefficient row
    per step on the device loops, in both schedules, against the oracle loops with the same list."""
    T, tstart = 8, 5
    etas = [1.0, 0.8, 0.6, 1.0, 0.9, 0.7, 1.0, 0.5]
    eng, ow, conds, to_c, x0 = _loop_setup("audioldm2", T)
    xts0 = ow.sample_xts_from_x0(x0, T, generator=torch.Generator().manual_seed(3))
    _, zs_o, xts_o = oloops.invert(ow, x0, conds["src"], conds["unc"], [3.0], T, eta=etas, xts=xts0.clone())
    w_o = oloops.edit(ow, xts_o, torch.tensor([tstart]), conds["tgt"], conds["unc"], [12.0], zs_o[:tstart], eta=etas)
    for mode in ("sequential", "batched"):
        zs, xts = eng.invert(x0, to_c(conds["src"]), to_c(conds["unc"]), [3.0], eta=etas, xts=xts0.unsqueeze(1), mode=mode,
                             group=4)
        w = eng.edit(xts, zs, tstart, to_c(conds["tgt"]), to_c(conds["unc"]), [12.0], eta=etas[:tstart])
        assert rel(eng.to_nchw(xts)[1:, 0], xts_o[1:]) < 1e-5
        assert rel(eng.to_nchw(zs)[1:, 0], zs_o[1:]) < 2e-3
        assert rel(eng.to_nchw(w), w_o) < 2e-3
    # a uniform list equals the scalar; a wrong length is refused
    za, _ = eng.invert(x0, to_c(conds["src"]), to_c(conds["unc"]), [3.0], eta=[1.0] * T, xts=xts0.unsqueeze(1))
    za = za.clone()
    zb, _ = eng.invert(x0, to_c(conds["src"]), to_c(conds["unc"]), [3.0], eta=1.0, xts=xts0.unsqueeze(1))
    assert torch.equal(za, zb)
    with pytest.raises(ValueError):
        eng.invert(x0, to_c(conds["src"]), to_c(conds["unc"]), [3.0], eta=[1.0] * (T - 1), xts=xts0.unsqueeze(1))


def test_timestep_batched_inversion_logic_on_cpu(cpu_loops):
    """The headline schedule: G timesteps per U-Net call (row/timestep index tables, per-group step ops, counter
    stride) gives the sequential result up to the 1-ulp numerical-fix coupling."""
    T = 8
    eng, ow, conds, to_c, x0 = _loop_setup("audioldm2", T)
    xts0 = eng.sample_xts(x0, generator=torch.Generator().manual_seed(2))
    zs_a, xts_a = eng.invert(x0, to_c(conds["src"]), to_c(conds["unc"]), [3.0], xts=xts0.clone())
    zs_a, xts_a = zs_a.clone(), xts_a.clone()
    zs_b, xts_b = eng.invert(x0, to_c(conds["src"]), to_c(conds["unc"]), [3.0], xts=xts0.clone(), mode="batched",
                             group=4)
    assert rel(xts_b[1:], xts_a[1:]) < 1e-5
    assert rel(zs_b[1:], zs_a[1:]) < 1e-3, rel(zs_b[1:], zs_a[1:])
    # against the oracle too
    _, zs_o, _ = oloops.invert(ow, x0, conds["src"], conds["unc"], [3.0], T, eta=1.0, xts=xts0[:, 0].clone())
    assert rel(eng.to_nchw(zs_b)[1:, 0], zs_o[1:]) < 1e-3


def test_two_clips_in_one_batch_on_cpu(cpu_loops):
    """n clips per engine (BASELINE config 3's 8-clips-per-GPU shape): rows [uncond x n | cond x n]."""
    T = 6
    eng, ow, conds, to_c, x0 = _loop_setup("audioldm2", T)
    g = torch.Generator().manual_seed(9)
    x0b = torch.cat([x0, torch.randn(1, 8, LH, LW, generator=g) * 0.8])
    xts0 = eng.sample_xts(x0b, generator=torch.Generator().manual_seed(4))
    src2 = Conditioning(ehs0=conds["src"]["encoder_hidden_states"].repeat(2, 1, 1),
                        ehs1=conds["src"]["encoder_hidden_states_1"].repeat(2, 1, 1),
                        mask1=conds["src"]["encoder_attention_mask_1"].repeat(2, 1))
    zs2, _ = eng.invert(x0b, src2, to_c(conds["unc"]), [3.0], xts=xts0.clone())
    zs2 = zs2.clone()
    for i in range(2):
        zs1, _ = eng.invert(x0b[i:i + 1], to_c(conds["src"]), to_c(conds["unc"]), [3.0], xts=xts0[:, i:i + 1].clone())
        assert rel(zs2[1:, i], zs1[1:, 0]) < 1e-4


# ------------------------------------------------------------------------------------------------ codec tapes on CPU
import os                                                                              # noqa: E402

import numpy as np                                                                     # noqa: E402

from audioeditingcode_amd.codec import STFTEngine, VAEDecoder, VAEEncoder, VocoderEngine  # noqa: E402
from oracle import hifigan as ohifi                                                    # noqa: E402
from oracle import vae as ovae                                                         # noqa: E402

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_stft_tape_reproduces_the_reference_fixture_on_cpu():
    """reflect pad -> DFT as a strided implicit GEMM (scalar-gather path, Cin = 1) -> magnitude -> mel GEMM + log."""
    g = np.load(os.path.join(G, "stft_mel_64f.npz"))
    wav = torch.from_numpy(g["wav"])[None]
    eng = STFTEngine(configs.STFT_AUDIOLDM, "cpu", 1, wav.shape[1])
    mel = eng(wav)
    np.testing.assert_allclose(mel[0].numpy(), torch.from_numpy(g["mel"])[0].T.numpy(), atol=2e-3, rtol=1e-4)
    np.testing.assert_allclose(eng.mag.numpy()[:, :513], g["mag"][0].T, atol=3e-4, rtol=1e-4)


def test_vae_tapes_match_oracle_on_cpu():
    cfg = configs.tiny_family("audioldm2")["vae"]
    sd = weights.random_state_dict(weights.vae_param_shapes(cfg), seed=0)
    mel = torch.randn(1, 1, 32, 16, generator=torch.Generator().manual_seed(1)) * 2 - 4
    enc = VAEEncoder(cfg, sd, "cpu", 1, 32, 16)
    lat = enc(mel)
    ref_lat = ovae.vae_encode(cfg, sd, mel)
    assert rel(lat.permute(0, 3, 1, 2), ref_lat) < 1e-4
    dec = VAEDecoder(cfg, sd, "cpu", 1, enc.h, enc.w)
    rec = dec(lat)
    assert rel(rec.permute(0, 3, 1, 2), ovae.vae_decode(cfg, sd, ref_lat)) < 2e-4


def test_vocoder_tape_reproduces_the_transformers_fixture_on_cpu():
    """Conv1d / phase-decomposed ConvTranspose1d / multi-receptive-field accumulate-and-mean wiring."""
    g = np.load(os.path.join(G, "hifigan_c64.npz"))
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}
    cfg = dict(configs.VOCODER_AUDIOLDM, upsample_initial_channel=64)
    mel = torch.from_numpy(g["mel"])
    eng = VocoderEngine(cfg, sd, "cpu", mel.shape[0], mel.shape[1])
    wav = eng(mel)
    assert tuple(wav.shape) == g["wav"].shape
    np.testing.assert_allclose(wav.numpy(), g["wav"], atol=2e-6, rtol=1e-4)


def test_unet_odd_latent_size_on_cpu():
    """forward_upsample_size (models.py:186-188): a latent height that is not a multiple of 2^(levels-1)."""
    fam = configs.tiny_family("audioldm2")
    cfg = fam["unet"]
    sd = weights.random_state_dict(weights.unet_param_shapes(cfg), seed=2)
    g = torch.Generator().manual_seed(3)
    B, H, W = 1, 18, 16
    x = torch.randn(B, cfg["in_channels"], H, W, generator=g)
    e0 = torch.randn(B, 4, fam["ctx"]["gpt2_dim"], generator=g)
    e1 = torch.randn(B, 3, fam["ctx"]["t5_dim"], generator=g)
    eng = UNetEngine(cfg, sd, "cpu", B, H, W, ctx_len0=4, ctx_len1=3)
    eng.set_conditioning(ehs0=e0, ehs1=e1, bias1=torch.zeros(B, 3))
    eng.x_in.copy_(x.permute(0, 2, 3, 1))
    eng.set_timestep(77)
    eng.forward()
    ref, _, _ = ounet.unet_forward(cfg, sd, x, torch.tensor(77), encoder_hidden_states=e0, encoder_hidden_states_1=e1,
                                   encoder_attention_mask_1=torch.ones(B, 3))
    assert (eng.eps.permute(0, 3, 1, 2) - ref).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())


def test_edit_latents_batch_equals_per_clip_runs_on_cpu(cpu_loops):
    """EditEngine.edit_latents (n clips, one U-Net batch per step; the bench's --clips-per-gpu path): clip i of a
    batch of 2 equals the single-clip run on the same noise, for both inversion schedules."""
    T, tstart = 6, 4
    eng, ow, conds, to_c, x0 = _loop_setup("audioldm2", T)
    g = torch.Generator().manual_seed(9)
    x0b = torch.cat([x0, torch.randn(1, 8, LH, LW, generator=g) * 0.8])
    noise = torch.randn(T, 2, 8, LH, LW, generator=torch.Generator().manual_seed(4))
    args = (to_c(conds["src"]), to_c(conds["unc"]), to_c(conds["tgt"]), to_c(conds["unc"]), [3.0], [12.0], tstart)
    for schedule, group in (("sequential", 1), ("batched", 3)):
        both = eng.edit_latents(x0b, *args, schedule=schedule, group=group, noise=noise).clone()
        for i in range(2):
            one = eng.edit_latents(x0b[i:i + 1], *args, schedule=schedule, group=group, noise=noise[:, i:i + 1])
            assert rel(both[i:i + 1], one) < 1e-4, (schedule, i, rel(both[i:i + 1], one))
    # and the whole thing against the oracle loops for clip 0
    xts0 = torch.cat([x0b[:1][None], eng.sample_xts(x0b[:1], noise=noise[:, :1])[1:]])[:, 0]
    _, zs_o, xts_o = oloops.invert(ow, x0b[:1], conds["src"], conds["unc"], [3.0], T, eta=1.0, xts=xts0.clone())
    w_o = oloops.edit(ow, xts_o, torch.tensor([tstart]), conds["tgt"], conds["unc"], [12.0], zs_o[:tstart], eta=1.0)
    assert rel(both[:1], w_o) < 2e-3


def test_tape_image_round_trip_in_host_memory(tmp_path):
    """image.export_image -> aed_image_load(flags=1: arena in host memory): the file carries both programs of a U-Net engine
    (per-prompt context tape + forward tape), every op survives with its pointers relocated into ONE arena at the same
    relative positions, named buffers hold what the engine held (weights and inputs travel, scratch buffers arrive zeroed),
    copy_in / copy_out address buffers by name, and an image that lives in host memory refuses to run."""
    import ctypes

    from audioeditingcode_amd import _lib as L, configs, weights
    from audioeditingcode_amd.image import Image, export_image
    from audioeditingcode_amd.unet import UNetEngine
    fam = configs.tiny_family("audioldm2")
    sd = weights.random_state_dict(weights.unet_param_shapes(fam["unet"]), seed=0)
    ts = torch.tensor([501, 301], dtype=torch.int64)
    state = torch.zeros(4, dtype=torch.int32)
    eng = UNetEngine(fam["unet"], sd, "cpu", 2, 32, 16, ctx_len0=8, ctx_len1=8, timesteps_dev=ts, state_dev=state)
    g = torch.Generator().manual_seed(1)
    eng.x_in.copy_(torch.randn(2, 32, 16, 8, generator=g))
    eng.ehs0.copy_(torch.randn(eng.ehs0.shape, generator=g))
    names = dict(x_in=eng.x_in, eps=eng.eps, ehs0=eng.ehs0, ehs1=eng.ehs1, bias1=eng.bias1, timesteps=ts, state=state,
                 conv_in_w=eng.wd["conv_in.weight"])
    path = str(tmp_path / "unet_tiny.aedimg")
    info = export_image(path, {"context": eng.ctx_tape, "forward": eng.tape}, names, scratch=[eng.eps, eng.h_space])
    assert info["n_ops"] == len(eng.ctx_tape.ops) + len(eng.tape.ops) and info["snapshot_bytes"] < info["arena_bytes"]
    im = Image(path, host=True)
    base, _ = im.buffer("x_in")
    for prog, tp in (("context", eng.ctx_tape), ("forward", eng.tape)):
        ops, n = im.program(prog)
        assert n == len(tp.ops)
        arr = tp.finalize()
        for k in range(n):
            a, b = ops[k], arr[k]
            assert a.code == b.code and a.flags == b.flags and list(a.i) == list(b.i) and list(a.f) == list(b.f)
            for j in range(10):
                assert bool(a.p[j]) == bool(b.p[j] and j != 7), (prog, k, j)
    ops, n = im.program("forward")
    conv_in = ops[[mt["name"] for mt in eng.tape.meta].index("conv_in")]
    assert conv_in.code == L.OP_CONV_GEMM and conv_in.p[0] == base         # conv_in reads the image's x_in
    w_ptr, w_bytes = im.buffer("conv_in_w")
    assert conv_in.p[1] == w_ptr and w_bytes == eng.wd["conv_in.weight"].numel() * 4
    read = lambda name, like: im.copy_out(name, torch.empty_like(like))    # noqa: E731
    assert torch.equal(read("x_in", eng.x_in), eng.x_in) and torch.equal(read("ehs0", eng.ehs0), eng.ehs0)
    assert torch.equal(read("conv_in_w", eng.wd["conv_in.weight"]), eng.wd["conv_in.weight"])
    assert torch.equal(read("timesteps", ts), ts)
    assert float(read("eps", eng.eps).abs().max()) == 0.0                 # scratch: zero after load
    new_x = torch.randn(2, 32, 16, 8, generator=g)
    im.copy_in("x_in", new_x)
    assert torch.equal(read("x_in", new_x), new_x)
    with pytest.raises(L.AedError, match="host memory"):
        im.run("forward")
    with pytest.raises(L.AedError, match="no program"):
        im.program("nope")
    with pytest.raises(L.AedError, match="do not fit"):
        im.copy_in("state", torch.zeros(64))
    im.close()
    with pytest.raises(L.AedError):
        Image(str(tmp_path / "missing.aedimg"))


def test_arith_mode_marks_only_the_lds_staged_gemms():
    """tape.arith_mode("bf16x6"): AED_OP_CONV_GEMM records built inside the context carry flag bits 2|3 exactly where the
    split-bf16 kernel applies (LDS-staged tiles 1-4, 32-wide channel chunks), get tile 8 where two 256x128 workgroups per CU
    still fill the chip, and nothing else about the record changes; outside the context (and after it) tapes are unflagged;
    an engine built by EditEngine below ARITH_MIN_BATCH never is.  The CPU interpreter ignores the bits (fp32 semantics)."""
    from audioeditingcode_amd import _lib as L, tape as tape_mod
    from audioeditingcode_amd.tape import Tape

    def build():
        tp = Tape("cpu")
        big = tp.alloc(64, 32, 16, 128)                       # M = 32768
        w3 = tp.alloc(128, 9 * 128)
        tp.conv(big, w3, None, tp.alloc(64, 32, 16, 128), B=64, IH=32, IW=16, Cin=128, OH=32, OW=16, N=128, KH=3, KW=3,
                pad_h=1, pad_w=1, name="big3x3")              # fp32 tile 1, enough 256x128 blocks? 128 x 1 = 128 < 512 -> stays 1
        huge = tp.alloc(512, 32, 16, 128)                     # M = 262144 -> 1024 blocks of 256x128
        tp.conv(huge, w3, None, tp.alloc(512, 32, 16, 128), B=512, IH=32, IW=16, Cin=128, OH=32, OW=16, N=128, KH=3, KW=3,
                pad_h=1, pad_w=1, name="huge3x3")
        tp.linear(tp.alloc(64, 128), tp.alloc(128, 128), None, tp.alloc(64, 128), M=64, K=128, N=128, name="small")   # lin tile
        odd = tp.alloc(64, 32, 16, 8)
        tp.conv(odd, tp.alloc(128, 72), None, tp.alloc(64, 32, 16, 128), B=64, IH=32, IW=16, Cin=8, OH=32, OW=16, N=128,
                KH=3, KW=3, pad_h=1, pad_w=1, name="cin8")    # scalar-gather shape
        return tp
    plain = build()
    with tape_mod.arith_mode("bf16x6"):
        x6 = build()
        assert tape_mod.ARITH_FLAGS["bf16x6"] == 12
    after = build()
    by_name = {}
    for a, b, c, mt in zip(plain.ops, x6.ops, after.ops, plain.meta):
        assert a.code == b.code == c.code == L.OP_CONV_GEMM and a.flags == c.flags == 0 and list(a.i) == list(c.i)
        assert list(a.i[:29]) == list(b.i[:29]) and list(a.i[30:]) == list(b.i[30:])
        by_name[mt["name"]] = (a.i[29], b.i[29], b.flags)
    # (round 5: a split-bf16 3x3 convolution on a 512-thread tile also carries flag bit 8 = 32-wide K chunks)
    assert by_name["big3x3"] == (1, 1, 12) and by_name["huge3x3"] == (1, 8, 12 | 256)
    assert by_name["small"][0] >= 10 and by_name["small"][1:] == (by_name["small"][0], 0)
    assert by_name["cin8"][2] == 0 and by_name["cin8"][0] == by_name["cin8"][1]
    assert tape_mod.x6_tile(204800, 256, 1, 1) == 8 and tape_mod.x6_tile(2048, 256, 1, 1) == 1
    assert tape_mod.x6_tile(204800, 64, 2, 1) == 2
    assert tape_mod.x6_tile(204800, 768, 1, 1) == 9 and tape_mod.x6_tile(51200, 640, 1, 1) == 8     # whole 256-wide tiles only
    with pytest.raises(KeyError):
        with tape_mod.arith_mode("fp4"):
            pass
    # the fp8 EXPERIMENT marks the same records with bit 6 on top of the split-bf16 bits (the fallback of shapes its kernel
    # does not take); on a CPU tape nothing is pre-quantised (no bit 7)
    with tape_mod.arith_mode("fp8"):
        f8 = build()
    for b, c in zip(x6.ops, f8.ops):
        assert c.flags == (b.flags | 64 if b.flags & 4 else 0) and list(b.i) == list(c.i)


def test_swept_x6_table_decides_kernel_and_tile_per_shape(monkeypatch):
    """tape.X6_TABLES (filled by tools/tile_table_from_sweep.py --x6): under arith_mode("bf16x6") an entry with tile >= 100 puts
    the shape on the split-bf16 kernel with that tile, an entry with tile < 100 keeps it on the fp32 kernel even though the
    default rule would have flagged it; without the mode the table is never consulted; a forced tile wins over the table."""
    from audioeditingcode_amd import tape as tape_mod
    from audioeditingcode_amd.tape import Tape
    M, N, K = 64 * 512, 128, 9 * 128
    monkeypatch.setitem(tape_mod.X6_TABLES, None, {(M, N, K, 0): (103, 1), (M, 256, K, 0): (2, 1)})

    def build(forced=0):
        tp = Tape("cpu")
        x = tp.alloc(64, 32, 16, 128)
        for n in (128, 256):
            tp.conv(x, tp.alloc(n, K), None, tp.alloc(64, 32, 16, n), B=64, IH=32, IW=16, Cin=128, OH=32, OW=16, N=n, KH=3,
                    KW=3, pad_h=1, pad_w=1, tile=forced)
        return [(o.flags, o.i[29]) for o in tp.ops]
    assert [f for f, _ in build()] == [0, 0]
    with tape_mod.arith_mode("bf16x6"):
        assert build() == [(12, 3), (0, 2)]
        assert build(forced=1) == [(12, 1), (12, 1)]


def test_attention_records_under_bf16x6_flag_and_lane_rule():
    """tape.attention under arith_mode("bf16x6"): the record carries flag bit 2 (the launcher then takes attention_x6.hip in the
    throughput regime); on a CU-masked lane's regime the tape forces that kernel (variant 3) as soon as the call has one
    workgroup per CU of the lane; short sequences, head dims the kernel does not have and fp32 engines keep variant 0."""
    from audioeditingcode_amd import tape as tape_mod
    from audioeditingcode_amd.tape import Tape

    def rec(B, N, D, regime, arith, H=8):
        tp = Tape("cpu")
        C = H * D
        q, o = tp.alloc(B, N, C), tp.alloc(B, N, C)
        with tape_mod.tile_regime(regime), tape_mod.arith_mode(arith):
            tp.attention(q, q, q, o, B=B, H=H, Nq=N, Nk=N, D=D, ldq=C, ldk=C, ldv=C, ldo=C, bsq=N * C, bsk=N * C, bsv=N * C,
                         bso=N * C, scale=D ** -0.5)
        return tp.ops[0].flags & 4, tp.ops[0].i[14]
    assert rec(2, 1024, 32, "cus64", "bf16x6") == (4, 3)          # the edit lanes' level-1 self-attention
    assert rec(2, 1024, 32, "cus128", "bf16x6") == (4, 3)
    assert rec(2, 1024, 32, None, "bf16x6") == (4, 0)             # whole chip at batch 2: the key-split fp32 kernel is faster
    assert rec(2, 256, 48, "cus64", "bf16x6") == (4, 0)           # 32 workgroups: fp32 kernel
    assert rec(200, 1024, 32, "cus128", "bf16x6") == (4, 3)       # the inversion's batch
    assert rec(2, 1024, 32, "cus64", "f32") == (0, 0)
    assert rec(8, 64, 80, "cus64", "bf16x6") == (4, 0)            # <= 64 keys: single-pass kernel; d_head 80 not in the split kernel


def test_cfg_row_sharing_computes_the_context_free_head_once(monkeypatch):
    """unet.UNetEngine(share=S) (round 5): rows S*p .. S*p+S-1 carry the same sample and timestep ([uncond | prompt] rows of one
    clip); conv_in, the context-free down block, the first resnet and the double-self-attention transformer of the first site are
    laid out at batch B / S and expanded -- same values as the full-batch engine on the interpreter, fewer executed flops, the
    ALGORITHMIC flop count unchanged; class-conditioned models (AudioLDM-1) and hook engines never share."""
    from audioeditingcode_amd import configs, weights
    from audioeditingcode_amd.tape import Tape
    from audioeditingcode_amd.unet import UNetEngine
    from oracle import tape_interp, unet as ounet
    monkeypatch.setattr(Tape, "run", tape_interp.run_tape)
    g = torch.Generator().manual_seed(1)
    for kind, L0, L1 in (("audioldm2", 8, 5), ("tango", 6, 0)):
        fam = configs.tiny_family(kind)
        cfg = fam["unet"]
        sd = weights.random_state_dict(weights.unet_param_shapes(cfg), seed=0)
        B, H, W, S = 6, 32, 16, 2
        x = torch.randn(B // S, H, W, 8, generator=g).repeat_interleave(S, 0)        # rows 2p, 2p+1 hold the same sample
        if kind == "audioldm2":
            cond = dict(ehs0=torch.randn(B, L0, fam["ctx"]["gpt2_dim"], generator=g),
                        ehs1=torch.randn(B, L1, fam["ctx"]["t5_dim"], generator=g), bias1=torch.zeros(B, L1))
        else:
            cond = dict(ehs0=torch.randn(B, L0, fam["ctx"]["t5_dim"], generator=g), bias0=torch.zeros(B, L0))
        outs, engs = [], []
        for share in (1, S):
            eng = UNetEngine(cfg, sd, "cpu", B, H, W, ctx_len0=L0, ctx_len1=L1, share=share)
            eng.set_conditioning(**cond)
            eng.x_in.copy_(x)
            eng.set_timestep(501)
            eng.forward()
            outs.append(eng.eps.clone())
            engs.append(eng)
        full, shared = engs
        assert shared.S == S and full.S == 1
        assert torch.allclose(outs[0], outs[1], rtol=0, atol=2e-6 * float(outs[0].abs().max()))
        assert shared.tape.flops == full.tape.flops                       # algorithmic work: the reference's count
        saved = 1 - shared.tape.exec_flops / full.tape.exec_flops
        assert (0.02 < saved < 0.08) if kind == "audioldm2" else (0.0 < saved < 0.05), (kind, saved)
        assert sum(1 for m in shared.tape.meta if m["name"] == "cfg_share.expand") >= 2 * S
        assert all(tuple(a.shape) == tuple(b.shape) for a, b in zip(full.skips, shared.skips))     # hooks see full-batch skips
    fam = configs.tiny_family("audioldm")
    sd = weights.random_state_dict(weights.unet_param_shapes(fam["unet"]), seed=0)
    assert UNetEngine(fam["unet"], sd, "cpu", 4, 32, 16, use_ehs=False, share=2).S == 1        # FiLM embedding differs per row

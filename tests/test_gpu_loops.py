"""GPU parity of the device-resident inversion / edit loops against the CPU oracle loops
(same weights, same CPU-drawn noise).  Tiny U-Net so the oracle finishes in seconds."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from audioeditingcode_amd import configs, weights                        # noqa: E402
from audioeditingcode_amd.editing import Conditioning, EditEngine         # noqa: E402
from audioeditingcode_amd.scheduler import DDIMScheduler                  # noqa: E402
from oracle import loops as oloops                                        # noqa: E402
from oracle import unet as ounet                                          # noqa: E402
from oracle.scheduler import OracleDDIMScheduler                          # noqa: E402
from conftest import oracle_run                                           # noqa: E402

DEV = "cuda:0"
H, W = 32, 16


def rel(a, b):
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def _setup(kind="audioldm2", T=20, seed=0):
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
    sched = DDIMScheduler()
    sched.set_timesteps(T)
    osched = OracleDDIMScheduler()
    osched.set_timesteps(T)

    def unet_fn(x, t, cond):
        kw = {k: (v.expand(x.shape[0], *v.shape[1:]) if torch.is_tensor(v) else v) for k, v in cond.items()}
        return ounet.unet_forward(cfg, sd, x, t, **kw)[0]
    ow = oloops.OracleWrapper(osched, unet_fn)
    eng = EditEngine(cfg, sd, sched, DEV, H, W, kind)
    x0 = torch.randn(1, 8, H, W, generator=g) * 0.8
    return fam, eng, ow, conds, to_c, x0


@pytest.mark.parametrize("kind", ["audioldm2", "audioldm"])
def test_ddpm_inversion_and_edit_match_oracle(kind):
    T, tstart = 20, 10
    fam, eng, ow, conds, to_c, x0 = _setup(kind, T)
    gen = torch.Generator().manual_seed(3)
    xts0 = ow.sample_xts_from_x0(x0, T, generator=gen)                      # CPU draws, reference order

    def oracle():
        _, zs_o, xts_o = oloops.invert(ow, x0, conds["src"], conds["unc"], [3.0], T, eta=1.0, xts=xts0.clone())
        w_o = oloops.edit(ow, xts_o, torch.tensor([tstart]), conds["tgt"], conds["unc"], [12.0], zs_o[:tstart], eta=1.0)
        return zs_o, xts_o, w_o
    zs_o, xts_o, w_o = oracle_run(f"loops_ddpm_T20_{kind}", oracle, xts0)

    xts_in = xts0.unsqueeze(1)                                              # [T+1, n=1, C, H, W]
    zs, xts = eng.invert(x0, to_c(conds["src"]), to_c(conds["unc"]), [3.0], eta=1.0, xts=xts_in)
    w = eng.edit(xts, zs, tstart, to_c(conds["tgt"]), to_c(conds["unc"]), [12.0], eta=1.0)
    torch.cuda.synchronize()
    zs_n = eng.to_nchw(zs)[:, 0].cpu()
    xts_n = eng.to_nchw(xts)[:, 0].cpu()
    w_n = eng.to_nchw(w).cpu()
    assert torch.equal(zs_n[0], torch.zeros_like(zs_n[0]))                  # inversion_utils.py:131-133
    assert rel(xts_n[1:], xts_o[1:]) < 1e-5                                 # numerically-fixed trajectory
    assert rel(zs_n[1:], zs_o[1:]) < 2e-3, rel(zs_n[1:], zs_o[1:])         # z amplifies eps error by 1/sigma_t
    assert rel(w_n, w_o) < 2e-3, rel(w_n, w_o)


def test_sample_xts_bit_exact_and_rng_order():
    fam, eng, ow, conds, to_c, x0 = _setup("audioldm2", 20)
    xts_o = ow.sample_xts_from_x0(x0, 20, generator=torch.Generator().manual_seed(11))
    xts = eng.sample_xts(x0, generator=torch.Generator().manual_seed(11))
    torch.cuda.synchronize()
    assert torch.equal(xts[:, 0].cpu(), xts_o)


def test_batched_timestep_inversion_close_to_sequential():
    T = 20
    fam, eng, ow, conds, to_c, x0 = _setup("audioldm2", T)
    xts0 = eng.sample_xts(x0, generator=torch.Generator().manual_seed(2))
    zs_a, xts_a = eng.invert(x0, to_c(conds["src"]), to_c(conds["unc"]), [3.0], xts=xts0.clone())
    zs_b, xts_b = eng.invert(x0, to_c(conds["src"]), to_c(conds["unc"]), [3.0], xts=xts0.clone(), mode="batched",
                             group=5)
    torch.cuda.synchronize()
    assert rel(xts_b[1:].cpu(), xts_a[1:].cpu()) < 1e-5
    assert rel(zs_b[1:].cpu(), zs_a[1:].cpu()) < 2e-3, rel(zs_b[1:].cpu(), zs_a[1:].cpu())


def test_two_clips_equal_single_clip_runs():
    T = 10
    fam, eng, ow, conds, to_c, x0 = _setup("audioldm2", T)
    g = torch.Generator().manual_seed(9)
    x0b = torch.cat([x0, torch.randn(1, 8, H, W, generator=g) * 0.8])
    xts0 = eng.sample_xts(x0b, generator=torch.Generator().manual_seed(4))      # [T+1, 2, C, H, W]
    src2 = Conditioning(ehs0=conds["src"]["encoder_hidden_states"].repeat(2, 1, 1),
                        ehs1=conds["src"]["encoder_hidden_states_1"].repeat(2, 1, 1),
                        mask1=conds["src"]["encoder_attention_mask_1"].repeat(2, 1))
    zs2, _ = eng.invert(x0b, src2, to_c(conds["unc"]), [3.0], xts=xts0.clone())
    for i in range(2):
        zs1, _ = eng.invert(x0b[i:i + 1], to_c(conds["src"]), to_c(conds["unc"]), [3.0],
                            xts=xts0[:, i:i + 1].clone())
        torch.cuda.synchronize()
        assert rel(zs2[1:, i].cpu(), zs1[1:, 0].cpu()) < 2e-3


def test_empty_source_prompt_skips_cond_pass():
    T = 10
    fam, eng, ow, conds, to_c, x0 = _setup("audioldm2", T)
    gen = torch.Generator().manual_seed(3)
    xts0 = ow.sample_xts_from_x0(x0, T, generator=gen)
    _, zs_o, _ = oloops.invert(ow, x0, None, conds["unc"], [3.0], T, eta=1.0, src_is_empty=True, xts=xts0.clone())
    zs, _ = eng.invert(x0, None, to_c(conds["unc"]), [3.0], xts=xts0.unsqueeze(1))
    torch.cuda.synchronize()
    assert rel(eng.to_nchw(zs)[1:, 0].cpu(), zs_o[1:]) < 2e-3


def test_ddim_baseline_matches_oracle():
    T, skip = 10, 3
    fam, eng, ow, conds, to_c, x0 = _setup("audioldm2", T)
    def oracle():
        wT_o = oloops.ddim_invert(ow, x0, conds["src"], conds["unc"], 3.0, T, skip)
        return wT_o, oloops.ddim_sample(ow, wT_o, conds["tgt"], conds["unc"], 12.0, skip=skip)
    wT_o, we_o = oracle_run("loops_ddim_T10", oracle, x0)
    wT = eng.ddim_invert(x0, to_c(conds["src"]), to_c(conds["unc"]), 3.0, skip=skip)
    we = eng.ddim_sample(wT, to_c(conds["tgt"]), to_c(conds["unc"]), 12.0, skip=skip)
    torch.cuda.synchronize()
    assert rel(eng.to_nchw(wT).cpu(), wT_o) < 1e-4
    assert rel(eng.to_nchw(we).cpu(), we_o) < 1e-3


def test_full_size_audioldm2_loop_both_schedules_vs_oracle():
    """BASELINE config 2 shapes (AudioLDM2 U-Net, latent 8x256x16) against a LIVE run of the CPU oracle at T=2/tstart=1 (the
    full-length comparison -- T=200, both schedules -- is tests/test_gpu_zzz_fullsize_oracle_fixture.py against a committed
    oracle run; this one needs no fixture and was T=8 = 124 s of the GPU suite's budget until round 4): the reference step
    order and the timestep-batched inversion land at the same distance from the oracle (reference order)."""
    T, tstart = 2, 1
    fam = configs.FAMILIES["audioldm2"]
    cfg = fam["unet"]
    sd = weights.random_state_dict(weights.unet_param_shapes(cfg), seed=0)
    g = torch.Generator().manual_seed(11)
    mk = lambda L1: dict(encoder_hidden_states=torch.randn(1, 8, 768, generator=g),          # noqa: E731
                         encoder_hidden_states_1=torch.randn(1, L1, 1024, generator=g),
                         encoder_attention_mask_1=torch.ones(1, L1))
    src, tgt, unc = mk(7), mk(9), mk(1)
    to_c = lambda d: Conditioning(ehs0=d["encoder_hidden_states"], ehs1=d["encoder_hidden_states_1"],  # noqa: E731
                                  mask1=d["encoder_attention_mask_1"])
    sched = DDIMScheduler()
    sched.set_timesteps(T)
    osched = OracleDDIMScheduler()
    osched.set_timesteps(T)
    ow = oloops.OracleWrapper(osched, lambda x, t, c: ounet.unet_forward(
        cfg, sd, x, t, **{k: v.expand(x.shape[0], *v.shape[1:]) for k, v in c.items()})[0])
    x0 = torch.randn(1, 8, 256, 16, generator=g) * 0.8
    xts0 = ow.sample_xts_from_x0(x0, T, generator=torch.Generator().manual_seed(1))
    _, zs_o, xts_o = oloops.invert(ow, x0, src, unc, [3.0], T, eta=1.0, xts=xts0.clone())
    w_o = oloops.edit(ow, xts_o, torch.tensor([tstart]), tgt, unc, [12.0], zs_o[:tstart], eta=1.0)
    eng = EditEngine(cfg, sd, sched, DEV, 256, 16, "audioldm2")
    errs = {}
    for mode in ("sequential", "batched"):
        zs, xts = eng.invert(x0, to_c(src), to_c(unc), [3.0], xts=xts0.unsqueeze(1), mode=mode, group=2)
        w = eng.edit(xts, zs, tstart, to_c(tgt), to_c(unc), [12.0], eta=1.0)
        torch.cuda.synchronize()
        errs[mode] = (rel(eng.to_nchw(zs)[1:, 0].cpu(), zs_o[1:]), rel(eng.to_nchw(w).cpu(), w_o))
    print("full-size loop vs live oracle (zs, edited latent):", errs)
    for mode, (ez, ew) in errs.items():
        assert ez < 2e-4 and ew < 2e-4, (mode, ez, ew)
    assert errs["batched"][1] < 3 * errs["sequential"][1] + 1e-5, errs


def test_replay_invariant_compares_with_the_recorded_trajectory():
    """SURVEY section 4's known-answer invariant on the device loops: replaying zs with the SOURCE prompt and the
    SOURCE cfg from x_T retraces the numerically-fixed trajectory (xts[idx] for idx = T-1 .. 1; x_0 itself is not
    recoverable because zs[0] := 0, inversion_utils.py:131-133)."""
    T = 20
    fam, eng, ow, conds, to_c, x0 = _setup("audioldm2", T)
    zs, xts = eng.invert(x0, to_c(conds["src"]), to_c(conds["unc"]), [3.0], generator=torch.Generator().manual_seed(1))
    zs, xts = zs.clone(), xts.clone()
    for start, steps in ((T, T - 1), (T, 7), (12, 11), (12, 5)):
        w = eng.edit(xts, zs, start, to_c(conds["src"]), to_c(conds["unc"]), [3.0], eta=1.0, n_steps=steps)
        torch.cuda.synchronize()
        assert torch.isfinite(w).all()
        err = rel(w.cpu(), xts[start - steps].cpu())
        assert err < 1e-4, (start, steps, err)


def test_plan_cache_is_bounded():
    """cfg / tstart sweeps must not grow HBM without bound (every loop plan owns trajectory buffers + a hipGraph)."""
    T = 6
    fam, eng, ow, conds, to_c, x0 = _setup("audioldm2", T)
    eng.max_plans = 3
    zs, xts = eng.invert(x0, to_c(conds["src"]), to_c(conds["unc"]), [3.0], generator=torch.Generator().manual_seed(1))
    zs, xts = zs.clone(), xts.clone()
    first = None
    for k, cfg in enumerate([2.0, 3.0, 4.0, 5.0, 6.0, 2.0]):
        w = eng.edit(xts, zs, 4, to_c(conds["tgt"]), to_c(conds["unc"]), [cfg], eta=1.0)
        torch.cuda.synchronize()
        first = w.cpu() if k == 0 else first
        assert len(eng._plans) <= 3
    assert torch.equal(w.cpu(), first)              # an evicted plan is rebuilt to the same result
    eng.clear_plans()
    assert not eng._plans


def test_full_size_headline_length_batched_vs_sequential():
    """BASELINE config 2 at its real LENGTH and the bench's real SHAPE: AudioLDM2 U-Net, latent 8x256x16, T=200, tstart=100,
    G=100 timesteps per U-Net call (U-Net batch 200, ~24 GB of activations: the engine the bench times) against the
    reference step order on the same device path -- the CPU oracle needs ~6 minutes per clip at this size, so the oracle
    comparison stays at T=8 above and this test pins the size-independent properties: finiteness, batched == sequential
    within the stated tolerance, the replay invariant."""
    T, tstart, G = 200, 100, 100
    fam = configs.FAMILIES["audioldm2"]
    cfg = fam["unet"]
    sd = weights.random_state_dict(weights.unet_param_shapes(cfg), seed=0)
    g = torch.Generator().manual_seed(11)
    mk = lambda L1: Conditioning(ehs0=torch.randn(1, 8, 768, generator=g), ehs1=torch.randn(1, L1, 1024, generator=g),  # noqa: E731
                                 mask1=torch.ones(1, L1))
    src, tgt, unc = mk(7), mk(9), mk(1)
    sched = DDIMScheduler()
    sched.set_timesteps(T)
    eng = EditEngine(cfg, sd, sched, DEV, 256, 16, "audioldm2")
    x0 = torch.randn(1, 8, 256, 16, generator=g) * 0.8
    xts0 = eng.sample_xts(x0, generator=torch.Generator().manual_seed(1))
    out = {}
    for mode in ("sequential", "batched"):
        zs, xts = eng.invert(x0, src, unc, [3.0], xts=xts0.clone(), mode=mode, group=G)
        zs, xts = zs.clone(), xts.clone()
        w = eng.edit(xts, zs, tstart, tgt, unc, [12.0], eta=1.0)
        torch.cuda.synchronize()
        assert torch.isfinite(zs).all() and torch.isfinite(xts[1:]).all() and torch.isfinite(w).all(), mode
        out[mode] = (zs.cpu(), xts.cpu(), w.cpu())
        # replay invariant at full length: source prompt + source cfg from x_100 retraces xts[1]
        back = eng.edit(xts, zs, tstart, src, unc, [3.0], eta=1.0, n_steps=tstart - 1)
        torch.cuda.synchronize()
        assert rel(back.cpu(), xts[1].cpu()) < 1e-3, (mode, rel(back.cpu(), xts[1].cpu()))
    (zs_s, xts_s, w_s), (zs_b, xts_b, w_b) = out["sequential"], out["batched"]
    assert rel(xts_b[1:], xts_s[1:]) < 1e-5, rel(xts_b[1:], xts_s[1:])
    assert rel(zs_b[1:], zs_s[1:]) < 5e-3, rel(zs_b[1:], zs_s[1:])
    assert rel(w_b, w_s) < 5e-3, rel(w_b, w_s)


def test_eight_clips_per_engine_full_size_audioldm2():
    """BASELINE config 3's per-rank shape at FULL size: 8 clips of the 346.9 M-parameter AudioLDM2 U-Net (latent 8x256x16)
    edited as one U-Net batch per step (edit: batch 16; batched inversion, 2 timesteps per call: batch 32) against
    single-clip runs with the same per-clip noise, at a short schedule."""
    T, tstart, n = 4, 2, 8
    fam = configs.FAMILIES["audioldm2"]
    cfg = fam["unet"]
    sd = weights.random_state_dict(weights.unet_param_shapes(cfg), seed=0)
    g = torch.Generator().manual_seed(11)
    mk = lambda L1: Conditioning(ehs0=torch.randn(1, 8, 768, generator=g), ehs1=torch.randn(1, L1, 1024, generator=g),  # noqa: E731
                                 mask1=torch.ones(1, L1))
    src, tgt, unc = mk(7), mk(9), mk(1)
    sched = DDIMScheduler()
    sched.set_timesteps(T)
    eng = EditEngine(cfg, sd, sched, DEV, 256, 16, "audioldm2")
    x0s = torch.randn(n, 8, 256, 16, generator=g) * 0.8
    noise = torch.randn(T, n, 8, 256, 16, generator=g)
    w8 = eng.edit_latents(x0s, src, unc, tgt, unc, [3.0], [12.0], tstart, schedule="batched", group=2, noise=noise)
    torch.cuda.synchronize()
    assert w8.shape == (n, 8, 256, 16) and torch.isfinite(w8).all()
    assert {key[0] for key in eng._unets} >= {16, 32}
    for i in (0, 5):
        w1 = eng.edit_latents(x0s[i:i + 1], src, unc, tgt, unc, [3.0], [12.0], tstart, schedule="sequential",
                              noise=noise[:, i:i + 1])
        torch.cuda.synchronize()
        assert rel(w8[i:i + 1].cpu(), w1.cpu()) < 3e-3, (i, rel(w8[i:i + 1].cpu(), w1.cpu()))


def test_eight_clips_per_engine_equal_eight_single_runs():
    """BASELINE config 3's per-rank shape: 8 clips edited as ONE U-Net batch per step (EditEngine.edit_latents, batch
    [uncond x 8 | prompt x 8]) against 8 single-clip runs with the same per-clip noise -- same kernels at another batch
    size, so the tolerance is the loop tolerance (z amplifies eps differences by 1/sigma_t)."""
    T, tstart, n = 10, 6, 8
    fam, eng, ow, conds, to_c, x0 = _setup("audioldm2", T)
    g = torch.Generator().manual_seed(21)
    x0s = torch.randn(n, 8, H, W, generator=g) * 0.8
    noise = torch.randn(T, n, 8, H, W, generator=g)
    w8 = eng.edit_latents(x0s, to_c(conds["src"]), to_c(conds["unc"]), to_c(conds["tgt"]), to_c(conds["unc"]), [3.0],
                          [12.0], tstart, schedule="batched", group=5, noise=noise)
    torch.cuda.synchronize()
    assert w8.shape == (n, 8, H, W) and torch.isfinite(w8).all()
    for i in (0, 3, 7):
        w1 = eng.edit_latents(x0s[i:i + 1], to_c(conds["src"]), to_c(conds["unc"]), to_c(conds["tgt"]),
                              to_c(conds["unc"]), [3.0], [12.0], tstart, schedule="sequential", noise=noise[:, i:i + 1])
        torch.cuda.synchronize()
        assert rel(w8[i:i + 1].cpu(), w1.cpu()) < 3e-3, (i, rel(w8[i:i + 1].cpu(), w1.cpu()))

"""Pin the CPU oracle (oracle/) to vectors produced by the reference's own code
(oracle/make_golden.py, run in the build container).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import audio as oaudio
from oracle import hifigan as ohifi
from oracle import loops as oloops
from oracle import unet as ounet
from oracle import vae as ovae
from oracle.scheduler import OracleDDIMScheduler
from oracle.synth import prompt_vec, synthetic_unet

G = os.path.join(os.path.dirname(__file__), "golden")


def _wrapper(T, pred="epsilon", alpha_one=False):
    s = OracleDDIMScheduler(prediction_type=pred, set_alpha_to_one=bool(alpha_one))
    s.set_timesteps(T)
    return oloops.OracleWrapper(s, synthetic_unet)


def _same_host_arithmetic(g, w, x0):
    """Bit-exact comparison is only meaningful when this CPU reproduces the generating machine's
    table and libm (torch's vectorised linspace/cumprod/tanh differ in the last bit across ISAs)."""
    same_tab = np.array_equal(w.model.scheduler.alphas_cumprod.numpy(), g["alphas_cumprod"])
    probe = synthetic_unet(x0, torch.tensor(501), torch.stack([prompt_vec("probe")])).numpy()
    return same_tab and np.array_equal(probe, g["probe"])


def _check(a, b, exact, what=""):
    a, b = np.asarray(a), np.asarray(b)
    if exact:
        np.testing.assert_array_equal(a, b, err_msg=what)
    else:
        fin = np.isfinite(b)
        assert np.array_equal(fin, np.isfinite(a)), what
        np.testing.assert_allclose(a[fin], b[fin], rtol=2e-3, atol=2e-3, err_msg=what)


def _cond(prompts):
    return torch.stack([prompt_vec(str(p)) for p in prompts])


@pytest.mark.parametrize("name", ["ddpm_T20", "ddpm_T20_emptysrc", "ddpm_T10_vpred", "ddpm_T8_alphaone",
                                  "ddpm_T12_two_prompts"])
def test_ddpm_loops_match_reference(name):
    g = np.load(os.path.join(G, f"loop_{name}.npz"))
    T, tstart = int(g["T"]), int(g["tstart"])
    w = _wrapper(T, str(g["pred"]), bool(g["alpha_one"]))
    x0 = torch.from_numpy(g["x0"])
    src, tgt = list(g["src"]), list(g["tgt"])
    exact = _same_host_arithmetic(g, w, x0)
    gen = torch.Generator().manual_seed(int(g["seed"]))
    xts0 = w.sample_xts_from_x0(x0, T, generator=gen)
    _check(xts0, g["xts_init"], exact, "RNG draw order (models.py:79-81)")
    empty = (len(src) == 1 and src[0] == "")
    _, zs, xts = oloops.invert(w, x0, _cond(src), _cond([""]), list(g["cfg_src"]), T, eta=1.0,
                               src_is_empty=empty, n_prompts=len(src), xts=xts0.clone(),
                               prompt_empty=[p == "" for p in src])
    ref_zs, ref_xts = torch.from_numpy(g["zs"]), torch.from_numpy(g["xts"])
    # idx 0 may hold inf/nan for set_alpha_to_one: compare with equal_nan
    _check(zs, ref_zs, exact, "zs")
    _check(xts, ref_xts, exact, "xts")
    ts = torch.tensor([tstart] * len(tgt), dtype=torch.int)
    w_edit = oloops.edit(w, xts, ts, _cond(tgt), _cond([""]), list(g["cfg_tar"]), zs[:tstart], eta=1.0,
                         n_prompts=len(tgt), fix_alpha=float(g["fix_alpha"]))
    _check(w_edit, g["w_edit"], exact, "w_edit")


def test_trajectory_replay_invariant():
    """SURVEY section 4 known-answer: replaying zs under the SAME prompt/cfg retraces xts bit-exactly."""
    g = np.load(os.path.join(G, "loop_ddpm_T20.npz"))
    T = int(g["T"])
    w = _wrapper(T)
    xts, zs = torch.from_numpy(g["xts"]), torch.from_numpy(g["zs"])
    src = list(g["src"])
    exact = _same_host_arithmetic(g, w, torch.from_numpy(g["x0"]))
    for tstart in (T - 1, 10):
        ts = torch.tensor([tstart], dtype=torch.int)
        xt = xts[tstart].unsqueeze(0)
        cfg, _ = oloops.segment_scales(1, xt.shape[1:], list(g["cfg_src"]), None, xt.dtype)
        s = w.model.scheduler
        tl = s.timesteps[-tstart:]
        for it, t in enumerate(tl):
            idx = tstart - it - 1
            eps = oloops.cfg_combine(w.unet(xt, t, _cond([""])), w.unet(xt, t, _cond(src)), cfg)
            xt = w.reverse_step_with_custom_noise(eps, t, xt, variance_noise=zs[idx].unsqueeze(0), eta=1.0)
            if idx >= 1:
                _check(xt[0], xts[idx], exact, f"replay tstart={tstart} idx={idx}")


def test_ddim_baseline_matches_reference():
    g = np.load(os.path.join(G, "loop_ddim_T10.npz"))
    T, skip = int(g["T"]), int(g["skip"])
    w = _wrapper(T)
    w0 = torch.from_numpy(g["w0"])
    g2 = np.load(os.path.join(G, "loop_ddpm_T20.npz"))
    exact = _same_host_arithmetic(g2, w, torch.from_numpy(g2["x0"]))
    wT = oloops.ddim_invert(w, w0, _cond(g["src"]), _cond([""]), float(g["cfg_src"]), T, skip)
    _check(wT, g["wT"], exact, "wT")
    we = oloops.ddim_sample(w, wT, _cond(g["tgt"]), _cond([""]), float(g["cfg_tar"]), skip=skip)
    _check(we, g["w_edit"], exact, "w_edit")


def test_step_math_vectors():
    g = np.load(os.path.join(G, "step_math_T200.npz"))
    w = _wrapper(200)
    # scalar (0-dim) arithmetic is IEEE-exact everywhere; only the vectorised table build is not
    w.model.scheduler.alphas_cumprod = torch.from_numpy(g["alphas_cumprod"])
    w.model.scheduler.final_alpha_cumprod = w.model.scheduler.alphas_cumprod[0]
    for i in range(int(g["n"])):
        t = torch.tensor(int(g[f"t{i}"]))
        z, xfix = w.get_zs_from_xts(torch.from_numpy(g[f"xt{i}"]), torch.from_numpy(g[f"xtm1{i}"]),
                                    torch.from_numpy(g[f"eps{i}"]), t, eta=1.0)
        np.testing.assert_array_equal(z.numpy(), g[f"z{i}"])
        np.testing.assert_array_equal(xfix.numpy(), g[f"xfix{i}"])
        prev = w.reverse_step_with_custom_noise(torch.from_numpy(g[f"eps{i}"]), t, torch.from_numpy(g[f"xt{i}"]),
                                                variance_noise=torch.from_numpy(g[f"z_in{i}"]), eta=1.0)
        np.testing.assert_array_equal(prev.numpy(), g[f"prev{i}"])


def test_scheduler_tables():
    s = OracleDDIMScheduler()
    s.set_timesteps(200)
    assert s.timesteps[0] == 996 and s.timesteps[-1] == 1 and len(s.timesteps) == 200
    g = np.load(os.path.join(G, "loop_ddpm_T20.npz"))
    s.set_timesteps(20)
    np.testing.assert_array_equal(s.timesteps.numpy(), g["timesteps"])
    np.testing.assert_array_equal(s.alphas_cumprod.numpy(), g["alphas_cumprod"])


def test_stft_mel_matches_reference():
    g = np.load(os.path.join(G, "stft_mel_64f.npz"))
    basis = oaudio.stft_basis()
    np.testing.assert_allclose(basis[g["basis_row_ids"]], g["basis_rows"], atol=2e-7, rtol=0)
    melb = oaudio.mel_basis()
    np.testing.assert_allclose(melb, g["mel_basis"], atol=1e-7, rtol=1e-5)
    wav = torch.from_numpy(g["wav"])[None]
    mag = oaudio.stft_magnitude(wav, basis)
    np.testing.assert_allclose(mag.numpy(), g["mag"], atol=2e-4, rtol=1e-4)
    mel, _, energy = oaudio.mel_spectrogram(wav, basis, melb)
    np.testing.assert_allclose(mel.numpy(), g["mel"], atol=1e-3, rtol=1e-4)
    np.testing.assert_allclose(energy.numpy(), g["energy"], atol=1e-3, rtol=1e-4)


def test_waveform_preparation_rules():
    w = np.sin(np.arange(1000) / 7.0).astype(np.float32) + 0.3
    out = oaudio.prepare_waveform(w, 1600)
    assert out.shape == (1600,) and out.dtype == np.float32
    assert abs(np.abs(out).max() - 0.5) < 1e-6            # second normalisation (tools.py:61-62)
    assert np.all(out[1000:] == 0)
    out2 = oaudio.prepare_waveform(w, 640)
    assert out2.shape == (1000,)                          # NOT cropped: the reference's first-axis slice (tools.py:39-40)
    fb = oaudio.pad_spec(torch.ones(7, 65), 10)
    assert fb.shape == (10, 64) and fb[7:].abs().sum() == 0   # even-bin trim + zero pad (tools.py:18-31)


def test_waveform_preparation_matches_the_reference_tools():
    """A1 pinned: audioldm/audio/tools.py (normalize_wav, pad_wav with its float64 zero-pad and its first-axis slice
    quirk, the second normalisation, get_mel_from_wav, _pad_spec) run by oracle/make_golden.py on in-memory waveforms
    -> the oracle's and the product's host-side restatements reproduce waveform and fbank."""
    from audioeditingcode_amd import utils as putils
    g = np.load(os.path.join(G, "waveform_prep.npz"))
    for name in ("short", "long", "exact"):
        raw, frames = g[f"{name}.raw"], int(g[f"{name}.frames"])
        ref_wave = g[f"{name}.wave"][0].astype(np.float32)
        for prep in (oaudio.prepare_waveform, putils.prepare_waveform):
            got = prep(raw, frames * 160)
            assert got.shape == ref_wave.shape, (name, got.shape, ref_wave.shape)
            np.testing.assert_allclose(got, ref_wave, atol=1e-7, rtol=0)
        fb, w = oaudio.wav_to_fbank(raw, frames)
        assert tuple(fb.shape) == g[f"{name}.fbank"].shape
        np.testing.assert_allclose(fb.numpy(), g[f"{name}.fbank"], atol=2e-4, rtol=0)
        np.testing.assert_allclose(w.numpy(), g[f"{name}.wav_t"], atol=1e-7, rtol=0)
    x = torch.from_numpy(g["pad_spec.in"])
    for fn in (oaudio.pad_spec, putils.pad_spec):
        assert torch.equal(fn(x, 10), torch.from_numpy(g["pad_spec.out10"]))
        assert torch.equal(fn(x, 5), torch.from_numpy(g["pad_spec.out5"]))


def test_hifigan_matches_transformers_class():
    g = np.load(os.path.join(G, "hifigan_c64.npz"))
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}
    cfg = dict(upsample_rates=[5, 4, 2, 2, 2], upsample_kernel_sizes=[16, 16, 8, 4, 4],
               resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3)
    wav = ohifi.hifigan_forward(cfg, sd, torch.from_numpy(g["mel"]))
    assert wav.shape == g["wav"].shape
    np.testing.assert_allclose(wav.numpy(), g["wav"], atol=1e-6, rtol=1e-5)


def test_unet_matches_twin():
    g = np.load(os.path.join(G, "unet_twin_c32.npz"))
    tsd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}
    sd = ounet.twin_to_diffusers(tsd, [1, 2], 2, {1})
    assert len(sd) == len(tsd)
    cfg = dict(block_out_channels=[32, 64], layers_per_block=2,
               down_block_types=["DownBlock2D", "CrossAttnDownBlock2D"],
               up_block_types=["CrossAttnUpBlock2D", "UpBlock2D"], attention_head_dim=[2, 4],
               cross_attention_dim=[32, 64], class_embed_type="simple_projection",
               class_embeddings_concat=True, norm_num_groups=32)
    out, h_space, skips = ounet.unet_forward(cfg, sd, torch.from_numpy(g["x"]), torch.from_numpy(g["t"]),
                                             class_labels=torch.from_numpy(g["y"]))
    np.testing.assert_allclose(out.numpy(), g["out"], atol=2e-5, rtol=1e-5)
    assert h_space.shape == (2, 64, 16, 4) and sorted(skips) == [0, 1]


def test_vae_matches_twin():
    g = np.load(os.path.join(G, "vae_twin_c32.npz"))
    tsd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}
    sd = ovae.twin_to_diffusers(tsd, 3, 2)
    assert len(sd) == len(tsd)
    cfg = dict(block_out_channels=[32, 32, 64], layers_per_block=2, latent_channels=8, scaling_factor=1.0)
    mom = ovae.encode_moments(cfg, sd, torch.from_numpy(g["mel"]))
    np.testing.assert_allclose(mom[:, :8].numpy(), g["mean"], atol=5e-5, rtol=1e-4)
    rec = ovae.decode(cfg, sd, torch.from_numpy(g["mean"]))
    np.testing.assert_allclose(rec.numpy(), g["recon"], atol=1e-4, rtol=1e-4)


def test_product_scheduler_and_coefficients_match_reference_scalars():
    """Host-side product code (no GPU): DDIM tables and the K1 coefficient rows."""
    from audioeditingcode_amd.scheduler import DDIMScheduler, step_coefficients
    g = np.load(os.path.join(G, "step_math_T200.npz"))
    s = DDIMScheduler()
    s.set_timesteps(200)
    o = OracleDDIMScheduler()
    o.set_timesteps(200)
    assert torch.equal(s.timesteps, o.timesteps) and torch.equal(s.alphas_cumprod, o.alphas_cumprod)
    if np.array_equal(s.alphas_cumprod.numpy(), g["alphas_cumprod"]):      # same-CPU tables: rows are bit-exact
        for i in range(int(g["n"])):
            np.testing.assert_array_equal(step_coefficients(s, int(g[f"t{i}"]), 1.0).numpy(), g[f"coef{i}"])
    else:
        np.testing.assert_allclose(s.alphas_cumprod.numpy(), g["alphas_cumprod"], rtol=2e-6)


def test_pc_drift_oracle_matches_reference():
    """forward_directional / get_eigenvectors / apply_drift vs the reference's pc_drift.py."""
    from oracle import pc as opc
    g = np.load(os.path.join(G, "pc_drift.npz"))
    T = int(g["T"])
    w = _wrapper(T)
    w.model.scheduler.alphas_cumprod = torch.from_numpy(g["alphas_cumprod"])
    w.model.scheduler.final_alpha_cumprod = w.model.scheduler.alphas_cumprod[0]
    xt, latent, mask = (torch.from_numpy(g[k]) for k in ("xt", "latent", "mask"))
    t = torch.tensor(int(g["t"]))
    unc, txt = _cond([""]), _cond(["a dog barking"])
    xtm1, x0p = opc.forward_directional(w, xt, t, latent, unc, txt, 3.0, eta=1.0)
    np.testing.assert_allclose(xtm1.numpy(), g["xtm1"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(x0p.numpy(), g["x0_pred"], rtol=1e-5, atol=1e-6)
    for n_ev in (1, 3):
        ev, val, _, nrm = opc.get_eigenvectors(w, xt, txt.repeat(n_ev, 1), unc.repeat(n_ev, 1), latent, mask, t,
                                               torch.from_numpy(g["x0_pred"]) * mask,
                                               torch.from_numpy(g[f"init{n_ev}"]), const=1e-3, cfg_tar=3.0,
                                               iters=int(g[f"iters{n_ev}"]), eta=1.0, n_ev=n_ev)
        np.testing.assert_allclose(torch.as_tensor(val).reshape(-1).numpy(), g[f"eigval{n_ev}"], rtol=2e-3)
        cos = (ev.reshape(n_ev, -1) * torch.from_numpy(g[f"eigvec{n_ev}"]).reshape(n_ev, -1)).sum(1)
        assert (cos.abs() > 0.999).all(), cos
    drift = opc.apply_drift(w, torch.from_numpy(g["xtm1"]), torch.from_numpy(g["x0_pred"]), t,
                            torch.from_numpy(g["eigvec3"]), torch.from_numpy(g["eigval3"]), latent, amount=2.0,
                            eta=1.0, ev_nums=(1, 2))
    np.testing.assert_allclose(drift.numpy(), g["drift"], rtol=1e-5, atol=2e-6)


def test_oracle_reproduces_the_reference_main_run_script():
    """tests/golden/main_run.npz holds outputs of the reference's OWN code/main_run.py run end to end on the synthetic
    model (oracle/make_golden.py pc_cli): single prompt, two target prompts with unequal tstart + cutoff + fix_alpha,
    and --mode ddim.  The oracle loops must retrace the script's glue (tstart/skip handling, zs slicing, cfg lists)."""
    g = np.load(os.path.join(G, "main_run.npz"))
    T = int(g["T"])
    w0 = torch.from_numpy(g["w0"])
    tol = dict(rtol=2e-5, atol=2e-6)
    # a: -s 21, source "a dog barking" cfg 3 -> target "a cat meowing" cfg 12, tstart 7
    w = _wrapper(T)
    torch.manual_seed(21)
    _, zs, xts = oloops.invert(w, w0, _cond(["a dog barking"]), _cond([""]), [3.0], T, eta=1.0)
    np.testing.assert_allclose(zs.numpy(), g["a_zs"], **tol)
    np.testing.assert_allclose(xts.numpy(), g["a_wts"], **tol)
    we = oloops.edit(w, xts, torch.tensor([7], dtype=torch.int), _cond(["a cat meowing"]), _cond([""]), [12.0], zs[:7],
                     eta=1.0, fix_alpha=0.1)
    np.testing.assert_allclose(we.numpy(), g["a_w_edit"], **tol)
    # b: -s 22, two target segments (cutoff 0.5) with tstart 7 and 5, cfg 12 / 8, fix_alpha 0.2
    w = _wrapper(T)
    torch.manual_seed(22)
    _, zs, xts = oloops.invert(w, w0, _cond(["rain"]), _cond([""]), [3.0], T, eta=1.0)
    we = oloops.edit(w, xts, torch.tensor([7, 5], dtype=torch.int), _cond(["jazz", "rock"]), _cond([""]), [12.0, 8.0],
                     zs[:7], eta=1.0, n_prompts=2, cutoff_points=[0.5], fix_alpha=0.2)
    np.testing.assert_allclose(we.numpy(), g["b_w_edit"], **tol)
    # c: --mode ddim, tstart 9 -> skip 3
    w = _wrapper(T)
    wT = oloops.ddim_invert(w, w0, _cond(["a dog barking"]), _cond([""]), 3.0, T, 3)
    np.testing.assert_allclose(wT.numpy(), g["c_wT"], **tol)
    we = oloops.ddim_sample(w, wT, _cond(["a cat meowing"]), _cond([""]), 12.0, skip=3)
    np.testing.assert_allclose(we.numpy(), g["c_w_edit"], **tol)


def test_oracle_unet_matches_the_references_inline_forward_graphs():
    """SURVEY A8, graph level: oracle/unet.py's forward against the reference's OWN `PipelineWrapper.unet_forward`
    (models.py:160-393) and `AudioLDM2Wrapper.unet_forward` (:691-899) executed on stand-in diffusers blocks
    (oracle/make_golden.py unet_graph): plain, replace_h_space, mid_block_additional_residual, replace_skip_conns,
    zero_out_resconns (int and list), and a size that forces forward_upsample_size -- eps, h_space and every extracted
    residual, for the three families."""
    import unet_graph_cases as ugc
    g = ugc.load()
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-12))                # noqa: E731
    n = 0
    for fam in ugc.FAMILIES:
        f, sd = ugc.family(g, fam)
        for fam_, size, hook in ugc.cases():
            if fam_ != fam:
                continue
            x, t, cond = ugc.inputs(g, fam, size)
            with torch.no_grad():
                got = ounet.unet_forward(f["unet"], sd, x, torch.tensor(t), **ugc.oracle_kwargs(fam, cond),
                                         **ugc.hook_kwargs(g, fam, hook))
            ugc.check(fam, size, hook, got, ugc.expected(g, fam, size, hook), 2e-5, rel)   # fp32 rounding: the CPU thread count changes summation orders
            n += 1
    assert n == 24

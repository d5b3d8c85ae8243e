"""The HIP loops at the BENCHED size and length against the CPU oracle: AudioLDM2 U-Net (346.9 M), latent 8x256x16, T=200,
tstart=100, cfg 3 / 12 -- both inversion schedules (reference order; 100 timesteps per U-Net call = batch 200).  The oracle
side (~6 min of CPU) is a committed fixture, tests/golden/fullsize_loop_T200.npz, written by
oracle/make_fullsize_loop_golden.py from seeds; every input is regenerated here from the same seeds.

First run on hardware: the round-3 driver's GPUTEST (passed); tolerances are ~10x the values observed since."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from audioeditingcode_amd import configs, weights                          # noqa: E402
from audioeditingcode_amd.editing import Conditioning, EditEngine          # noqa: E402
from audioeditingcode_amd.scheduler import DDIMScheduler                   # noqa: E402
from oracle import loops as oloops                                         # noqa: E402
from oracle.make_fullsize_loop_golden import T, TSTART, inputs             # noqa: E402
from oracle.scheduler import OracleDDIMScheduler                           # noqa: E402

DEV = "cuda:0"


def test_full_size_headline_length_loops_vs_the_oracle_fixture(golden_dir):
    path = os.path.join(golden_dir, "fullsize_loop_T200.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/fullsize_loop_T200.npz: run oracle/make_fullsize_loop_golden.py")
    fx = np.load(path)
    assert int(fx["T"]) == T and int(fx["tstart"]) == TSTART
    cfg = configs.FAMILIES["audioldm2"]["unet"]
    sd = weights.random_state_dict(weights.unet_param_shapes(cfg), seed=0)
    src, tgt, unc, x0 = inputs()
    to_c = lambda d: Conditioning(ehs0=d["encoder_hidden_states"], ehs1=d["encoder_hidden_states_1"],  # noqa: E731
                                  mask1=d["encoder_attention_mask_1"])
    osched = OracleDDIMScheduler()
    osched.set_timesteps(T)
    xts0 = oloops.OracleWrapper(osched, None).sample_xts_from_x0(x0, T, generator=torch.Generator().manual_seed(1))
    sched = DDIMScheduler()
    sched.set_timesteps(T)
    eng = EditEngine(cfg, sd, sched, DEV, 256, 16, "audioldm2")
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())                    # noqa: E731
    keep = [int(k) for k in fx["keep"]]
    errs = {}
    for mode, group in (("batched", 100), ("sequential", 1)):
        zs, xts = eng.invert(x0, to_c(src), to_c(unc), [3.0], xts=xts0.unsqueeze(1), mode=mode, group=group)
        w = eng.edit(xts, zs, TSTART, to_c(tgt), to_c(unc), [12.0], eta=1.0)
        torch.cuda.synchronize()
        zs_c, xts_c = eng.to_nchw(zs)[:, 0].cpu(), eng.to_nchw(xts)[:, 0].cpu()
        assert list(zs_c.shape) == [int(v) for v in fx["zs_shape"]] and list(xts_c.shape) == [int(v) for v in fx["xts_shape"]]
        errs[mode] = dict(
            w_edit=rel(eng.to_nchw(w).cpu(), torch.from_numpy(fx["w_edit"])),
            xT=rel(xts_c[-1], torch.from_numpy(fx["xT"])),
            zs=max(rel(zs_c[k], torch.from_numpy(fx["zs_keep"][j])) for j, k in enumerate(keep)),
            zs_norms=float((zs_c.flatten(1).norm(dim=1) - torch.from_numpy(fx["zs_norms"])).abs().max()
                           / torch.from_numpy(fx["zs_norms"]).max()),
            xts_norms=float((xts_c.flatten(1).norm(dim=1) - torch.from_numpy(fx["xts_norms"])).abs().max()
                            / torch.from_numpy(fx["xts_norms"]).max()))
    print("HIP vs oracle at T=200, full size:", errs)
    # ~10x the values observed on the MI355X (round 4, both arithmetics: w_edit 4.4e-6 / 4.6e-6, zs 3.8e-6 ... 4.7e-6, xT 7e-8,
    # norms 1.9e-7); the path's stated tolerance is 5e-3 on the edited latent
    for mode, e in errs.items():
        assert e["w_edit"] < 5e-5 and e["xT"] < 2e-6 and e["zs"] < 5e-5 and e["zs_norms"] < 5e-6 and e["xts_norms"] < 5e-6, (mode, e)


def test_stable_audio_dit_at_full_depth_vs_the_oracle_fixture(golden_dir):
    """Stable Audio Open 1.0 DiT, all 24 layers (1.06 B seeded-random parameters, 1025-token sequence, batch 2 = [zero context
    | prompt]): the compiled tape against the CPU oracle's forward stored in tests/golden/dit_full_depth.npz
    (oracle/make_fullsize_dit_golden.py).  The live oracle comparison of test_gpu_stable_audio.py stops at 4 layers."""
    from audioeditingcode_amd.scheduler import CosineDPMSolverMultistepScheduler
    from audioeditingcode_amd.stable_audio import DiTEngine
    from oracle.make_fullsize_dit_golden import S, STEP, case
    path = os.path.join(golden_dir, "dit_full_depth.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/dit_full_depth.npz: run oracle/make_fullsize_dit_golden.py")
    fx = np.load(path)
    cfg = dict(configs.FAMILIES["stable_audio"]["dit"])
    assert int(fx["num_layers"]) == cfg["num_layers"] == 24
    sd = weights.random_state_dict(weights.dit_param_shapes(cfg), seed=3)
    x, ctx, glob = case(cfg)
    s = CosineDPMSolverMultistepScheduler()
    s.set_timesteps(200)
    t = s.timesteps[STEP]
    assert abs(float(t) - float(fx["t"])) <= 1e-6 * abs(float(fx["t"]))
    eng = DiTEngine(cfg, sd, DEV, 2, S)
    eng.set_conditioning(ctx, glob)
    eng.set_timestep(t)
    eng.x_in.copy_(x.transpose(1, 2))
    v = eng.forward().transpose(1, 2).cpu()
    torch.cuda.synchronize()
    ref = torch.from_numpy(fx["v"])
    r = float((v.double() - ref.double()).norm() / ref.double().norm())
    print("DiT at full depth, HIP vs oracle: rel L2", r)
    assert torch.isfinite(v).all() and r < 5e-5, r              # observed 2.3e-6 (round 4)
    assert float((v[0] - v[1]).abs().max()) > 1e-3


def test_eight_clips_per_engine_at_full_size_vs_the_oracle_fixture(golden_dir):
    """BASELINE config 3's per-rank shape (8 clips of the full-size AudioLDM2 U-Net per engine: edit batch 16, batched
    inversion batch 32) against the CPU ORACLE's one-at-a-time edits of clips 0, 5 and 7 from the same per-clip noise maps
    (tests/golden/fullsize_eight_clips_T4.npz; test_gpu_loops.py compares the same batch with single-clip GPU runs)."""
    from oracle.make_fullsize_loop_golden import CLIPS8, N8, T8, TSTART8, inputs_eight
    path = os.path.join(golden_dir, "fullsize_eight_clips_T4.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/fullsize_eight_clips_T4.npz: run oracle/make_fullsize_loop_golden.py eight")
    fx = np.load(path)
    assert [int(c) for c in fx["clips"]] == list(CLIPS8) and int(fx["T"]) == T8
    cfg = configs.FAMILIES["audioldm2"]["unet"]
    sd = weights.random_state_dict(weights.unet_param_shapes(cfg), seed=0)
    src, tgt, unc, x0s, noise = inputs_eight()
    to_c = lambda d: Conditioning(ehs0=d["encoder_hidden_states"], ehs1=d["encoder_hidden_states_1"],  # noqa: E731
                                  mask1=d["encoder_attention_mask_1"])
    sched = DDIMScheduler()
    sched.set_timesteps(T8)
    eng = EditEngine(cfg, sd, sched, DEV, 256, 16, "audioldm2")
    w8 = eng.edit_latents(x0s, to_c(src), to_c(unc), to_c(tgt), to_c(unc), [3.0], [12.0], TSTART8, schedule="batched",
                          group=2, noise=noise)
    torch.cuda.synchronize()
    assert w8.shape == (N8, 8, 256, 16)
    ref = torch.from_numpy(fx["w_edit"])
    errs = [float((w8[c].cpu().double() - ref[j].double()).norm() / ref[j].double().norm()) for j, c in enumerate(CLIPS8)]
    print("8 clips per engine, full size, HIP vs oracle:", errs)
    assert max(errs) < 1e-4, errs                                # observed 7.0e-6 ... 7.3e-6 (round 4, both arithmetics)


def test_baseline_config1_audioldm_s_ddim_clip_at_full_size_vs_the_oracle_fixture(golden_dir):
    """BASELINE configs[0] at its STATED size on the HIP path (VERDICT r4 `configs_untested`): one 10 s clip, AudioLDM-S
    (185 M U-Net), `--mode ddim`, 50 DDIM steps -- ddim_inversion (ddim_inversion.py:44-56, cfg 3) + text2image_ldm_stable
    (:59-84, cfg 12), VAE encode / decode, vocoder -- through main_run.edit_clip, against the CPU oracle's run of the same clip
    (tests/golden/config1_ddim_T50.npz, oracle/make_config1_golden.py).  Latent after the inversion, edited latent, decoded mel
    and waveform."""
    from audioeditingcode_amd import models
    from audioeditingcode_amd.main_run import edit_clip
    path = os.path.join(golden_dir, "config1_ddim_T50.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/config1_ddim_T50.npz: run oracle/make_config1_golden.py")
    fx = np.load(path)
    T = int(fx["T"])
    assert T == 50 and str(fx["model"]) == "cvssp/audioldm-s-full"
    src, tgt = [str(fx["prompts"][0])], [str(fx["prompts"][1])]
    m = models.load_model(str(fx["model"]), DEV, T, allow_synthetic=True)          # seeded-random weights: unet 0, vae 1, vocoder 2
    x0 = torch.from_numpy(fx["x0"]).to(DEV)
    rel = lambda a, b: float((a.double().cpu() - b.double()).norm() / b.double().norm())              # noqa: E731
    with torch.inference_mode():
        w0 = m.vae_encode(x0)
        from audioeditingcode_amd.ddm_inversion.ddim_inversion import ddim_inversion
        wT = ddim_inversion(m, w0, src, 3.0, num_inference_steps=T, skip=0)
    audio, _, w_edit = edit_clip(m, x0, src, tgt, [""], [3.0], [12.0], T, T, mode="ddim")
    with torch.inference_mode():
        mel = m.vae_decode(w_edit)
    torch.cuda.synchronize()
    errs = dict(w0=rel(w0, torch.from_numpy(fx["w0"])), wT=rel(wT.reshape(fx["wT"].shape), torch.from_numpy(fx["wT"])),
                w_edit=rel(w_edit, torch.from_numpy(fx["w_edit"])), mel=rel(mel, torch.from_numpy(fx["mel"])),
                wav=rel(audio, torch.from_numpy(fx["wav"])))
    print("config 1 (AudioLDM-S, 50-step DDIM, full size), HIP vs oracle:", errs)
    assert torch.isfinite(w_edit).all() and torch.isfinite(audio).all()
    # the path's stated tolerance: 5e-3 on the edited latent after a full loop, 2e-2 on the waveform;
    # 100 deterministic DDIM steps at cfg 3 / 12 amplify forward-level differences (1e-6) by far less than the DDPM z maps do
    assert errs["w0"] < 2e-4 and errs["wT"] < 5e-4 and errs["w_edit"] < 5e-3 and errs["mel"] < 5e-3 and errs["wav"] < 2e-2, errs

#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REFERENCE's own Python here.

TEST INFRASTRUCTURE.  Runs only in the build container (needs /root/reference);
nothing from /root/reference is copied -- only input/output tensors are saved.
The reference imports third-party packages that are absent here (diffusers,
torchvision, torchaudio, librosa, wandb, soundfile, progressbar); name-only stub
modules are injected exactly as SURVEY.md 8(c)/Appendix B describes.  What each
fixture pins is listed in tests/golden/README.md.

Usage:  python oracle/make_golden.py [loops|pc|pc_cli|audio|hifigan|unet|unet_graph|vae|stable_audio|text|all]
"""
import importlib.util
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/code"
OUT = os.path.join(ROOT, "tests", "golden")

import transformers  # noqa: E402  (must precede the stubs)
from transformers.audio_utils import mel_filter_bank  # noqa: E402

from oracle import loops as oloops  # noqa: E402
from oracle.scheduler import OracleDDIMScheduler  # noqa: E402
from oracle.synth import synthetic_unet, prompt_vec, chirp_waveform  # noqa: E402


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install_stubs():
    from dataclasses import dataclass

    @dataclass
    class UNet2DConditionOutput:
        sample: torch.Tensor = None

    dummy = type("Dummy", (), {})
    _stub("diffusers", DDIMScheduler=dummy, UNet2DModel=dummy, VQModel=dummy,
          CosineDPMSolverMultistepScheduler=dummy, AudioLDMPipeline=dummy, AudioLDM2Pipeline=dummy,
          StableDiffusionPipeline=dummy, StableAudioPipeline=dummy)
    _stub("diffusers.schedulers")
    _stub("diffusers.schedulers.scheduling_dpmsolver_sde", BrownianTreeNoiseSampler=dummy)
    _stub("diffusers.models")
    _stub("diffusers.models.unets")
    _stub("diffusers.models.unets.unet_2d_condition", UNet2DConditionOutput=UNet2DConditionOutput)
    _stub("diffusers.models.embeddings", get_1d_rotary_pos_embed=None)
    tvf = types.SimpleNamespace(gaussian_blur=lambda x, kernel_size, sigma: oloops.gaussian_blur(x, kernel_size, sigma))
    tvt = _stub("torchvision.transforms", functional=tvf)
    _stub("torchvision", transforms=tvt)
    _stub("wandb")
    _stub("torchaudio")
    _stub("soundfile")
    _stub("progressbar")
    return UNet2DConditionOutput


# --------------------------------------------------------------------------- loops
def gen_loops():
    UOut = install_stubs()
    sys.path.insert(0, REF)
    import models as ref_models  # reference code/models.py
    from ddm_inversion import inversion_utils as ref_inv
    from ddm_inversion import ddim_inversion as ref_ddim

    class FakeRef(ref_models.PipelineWrapper):
        """Reference PipelineWrapper + AudioLDM2Wrapper's variance helpers, synthetic eps-model."""
        get_variance = ref_models.AudioLDM2Wrapper.get_variance
        get_alpha_prod_t_prev = ref_models.AudioLDM2Wrapper.get_alpha_prod_t_prev

        def __init__(self, sched, T):
            super().__init__(model_id="fake", device=torch.device("cpu"))
            sched.set_timesteps(T)
            unet = SimpleNamespace(config=SimpleNamespace(in_channels=8))
            self.model = SimpleNamespace(scheduler=sched, unet=unet)

        def encode_text(self, prompts, **kw):
            return None, torch.stack([prompt_vec(p) for p in prompts]), None

        def unet_forward(self, sample, timestep, encoder_hidden_states=None, class_labels=None,
                         encoder_attention_mask=None, **kw):
            return UOut(sample=synthetic_unet(sample, timestep, class_labels)), None, None

    _probe_cond = torch.stack([prompt_vec("probe")])

    def run_case(name, T, tstart, src, tgt, cfg_src, cfg_tar, pred="epsilon", alpha_one=False,
                 seed=0, shape=(8, 16, 16), cutoff=None, fix_alpha=0.1):
        sched = OracleDDIMScheduler(prediction_type=pred, set_alpha_to_one=alpha_one)
        m = FakeRef(sched, T)
        g = torch.Generator().manual_seed(100 + seed)
        x0 = torch.randn((1, *shape), generator=g) * 0.7
        torch.manual_seed(seed)
        with torch.no_grad():
            _, zs, wts, _ = ref_inv.inversion_forward_process(
                m, x0, etas=1.0, prompts=list(src), cfg_scales=list(cfg_src), num_inference_steps=T,
                numerical_fix=True, cutoff_points=cutoff)
            ts = torch.tensor([tstart] * len(tgt), dtype=torch.int)
            w0, _ = ref_inv.inversion_reverse_process(
                m, xT=wts, tstart=ts, fix_alpha=fix_alpha, etas=1.0, prompts=list(tgt), neg_prompts=[""],
                cfg_scales=list(cfg_tar), zs=zs[:int(tstart)], cutoff_points=cutoff)
        # the independent x_t draws are re-derivable from (x0, seed); stored for convenience
        torch.manual_seed(seed)
        xts_init = m.sample_xts_from_x0(x0, num_inference_steps=T)
        np.savez_compressed(os.path.join(OUT, f"loop_{name}.npz"),
                            x0=x0.numpy(), xts_init=xts_init.numpy(), zs=zs.numpy(), xts=wts.numpy(),
                            w_edit=w0.numpy(), T=T, tstart=tstart, seed=seed,
                            cfg_src=np.array(cfg_src, dtype=np.float64), cfg_tar=np.array(cfg_tar, dtype=np.float64),
                            src=np.array(list(src)), tgt=np.array(list(tgt)), pred=pred, alpha_one=alpha_one,
                            fix_alpha=fix_alpha,
                            alphas_cumprod=sched.alphas_cumprod.numpy(), timesteps=sched.timesteps.numpy(),
                            probe=synthetic_unet(x0, torch.tensor(501), _probe_cond).numpy())
        print("loop", name, "zs", tuple(zs.shape), "finite w_edit:", bool(torch.isfinite(w0).all()))

    run_case("ddpm_T20", 20, 10, ["a dog barking"], ["a cat meowing"], [3.0], [12.0])
    run_case("ddpm_T20_emptysrc", 20, 12, [""], ["a cat meowing"], [3.0], [12.0], seed=1)
    run_case("ddpm_T10_vpred", 10, 6, ["rain"], ["jazz"], [3.5], [9.0], pred="v_prediction", seed=2)
    run_case("ddpm_T8_alphaone", 8, 5, ["rain"], ["jazz"], [3.0], [12.0], alpha_one=True, seed=3)
    run_case("ddpm_T12_two_prompts", 12, 7, ["rain", "wind"], ["jazz", "rock"], [3.0], [12.0, 8.0],
             seed=4, shape=(8, 32, 16))

    # --- DDIM baseline (deterministic) ---
    _stub("utils")  # ddim_inversion imports utils.get_text_embeddings; supply a minimal one
    sched = OracleDDIMScheduler()
    T = 10
    m = FakeRef(sched, T)

    def get_text_embeddings(tp, tn, model):
        a = SimpleNamespace(embedding_hidden_states=None, boolean_prompt_mask=None,
                            embedding_class_lables=model.encode_text(tp)[1])
        b = SimpleNamespace(embedding_hidden_states=None, boolean_prompt_mask=None,
                            embedding_class_lables=model.encode_text(tn)[1])
        return a.embedding_class_lables, a, b
    ref_ddim.get_text_embeddings = get_text_embeddings
    g = torch.Generator().manual_seed(77)
    w0 = torch.randn((1, 8, 16, 16), generator=g) * 0.7
    skip = 3
    wT = ref_ddim.ddim_inversion(m, w0, ["a dog barking"], 3.0, num_inference_steps=T, skip=skip)
    w_rec = ref_ddim.text2image_ldm_stable(m, ["a cat meowing"], T, 12.0, wT, skip=skip)
    np.savez_compressed(os.path.join(OUT, "loop_ddim_T10.npz"), w0=w0.numpy(), wT=wT.numpy(),
                        w_edit=w_rec.numpy(), T=T, skip=skip, cfg_src=3.0, cfg_tar=12.0,
                        src=np.array(["a dog barking"]), tgt=np.array(["a cat meowing"]))
    print("ddim", float(wT.abs().mean()), float(w_rec.abs().mean()))

    # --- bare step-math vectors (bit-exact targets for kernel K1) ---
    sched = OracleDDIMScheduler()
    T = 200
    m = FakeRef(sched, T)
    g = torch.Generator().manual_seed(5)
    rec = {}
    for i, t in enumerate([996, 501, 6, 1]):
        tt = torch.tensor(t)
        xt, xtm1, eps, z = (torch.randn((1, 8, 16, 16), generator=g) for _ in range(4))
        zz, xfix, _ = m.get_zs_from_xts(xt, xtm1, eps, tt, eta=1.0, numerical_fix=True)
        prev = m.reverse_step_with_custom_noise(eps, tt, xt, variance_noise=z, eta=1.0)
        abar = sched.alphas_cumprod
        prev_t = tt - 1000 // T
        a_prev = m.get_alpha_prod_t_prev(prev_t)
        var = m.get_variance(tt, prev_t)
        coef = torch.stack([(1 - abar[tt]) ** 0.5, abar[tt] ** 0.5, a_prev ** 0.5, (1 - a_prev - 1.0 * var) ** 0.5,
                            1.0 * var ** 0.5, torch.tensor(0.), torch.tensor(0.), torch.tensor(0.)])
        rec[f"coef{i}"] = coef.numpy()
        rec.update({f"t{i}": t, f"xt{i}": xt.numpy(), f"xtm1{i}": xtm1.numpy(), f"eps{i}": eps.numpy(),
                    f"z_in{i}": z.numpy(), f"z{i}": zz.numpy(), f"xfix{i}": xfix.numpy(), f"prev{i}": prev.numpy()})
    np.savez_compressed(os.path.join(OUT, "step_math_T200.npz"), n=4, alphas_cumprod=sched.alphas_cumprod.numpy(), **rec)
    print("step math ok")


# --------------------------------------------------------------------------- pc drift
def gen_pc():
    UOut = install_stubs()
    sys.path.insert(0, REF)
    import models as ref_models
    import pc_drift as ref_pc

    class FakeRef(ref_models.PipelineWrapper):
        def __init__(self, sched, T):
            super().__init__(model_id="fake", device=torch.device("cpu"))
            sched.set_timesteps(T)
            self.model = SimpleNamespace(scheduler=sched, unet=SimpleNamespace(config=SimpleNamespace(in_channels=8)))

        def unet_forward(self, sample, timestep, encoder_hidden_states=None, class_labels=None,
                         encoder_attention_mask=None, **kw):
            return UOut(sample=synthetic_unet(sample, timestep, class_labels)), None, None

    T = 50
    sched = OracleDDIMScheduler()
    m = FakeRef(sched, T)
    g = torch.Generator().manual_seed(21)
    xt = torch.randn((1, 8, 16, 16), generator=g) * 0.8
    latent = torch.randn((1, 8, 16, 16), generator=g)
    t = sched.timesteps[30]
    mk = lambda p: ref_pc.PromptEmbeddings(embedding_hidden_states=None, boolean_prompt_mask=None,   # noqa: E731
                                           embedding_class_lables=torch.stack([prompt_vec(p)]))
    unc, txt = mk(""), mk("a dog barking")
    mask = torch.ones_like(xt)
    mask[..., 12:] = 0
    with torch.no_grad():
        xtm1, x0_pred = ref_pc.forward_directional(m, xt, t, latent, unc, txt, 3.0, eta=1.0)
        rec = {}
        for n_ev, iters in ((1, 6), (3, 5)):
            torch.manual_seed(100 + n_ev)
            init = torch.randn((n_ev, 8, 16, 16))          # what randn_like(expanded xt) draws from this seed
            torch.manual_seed(100 + n_ev)
            ev, val, corr, nrm, _, _ = ref_pc.get_eigenvectors(m, xt, txt, unc, latent, mask, t, x0_pred * mask,
                                                               const=1e-3, cfg_tar=3.0, iters=iters, eta=1.0,
                                                               n_ev=n_ev)
            rec.update({f"init{n_ev}": init.numpy(), f"eigvec{n_ev}": ev.numpy(),
                        f"eigval{n_ev}": torch.as_tensor(val).reshape(-1).numpy(), f"iters{n_ev}": iters,
                        f"norm{n_ev}": torch.stack([torch.as_tensor(v).reshape(-1) for v in nrm]).numpy()})
        eigdata = {int(t): dict(eigvec=ev, eigval=torch.as_tensor(val).reshape(-1))}
        drift = ref_pc.apply_drift(m, xtm1, x0_pred, t, sched.timesteps, T, eigdata, latent, torch.device("cpu"),
                                   amount=2.0, eta=1.0, ev_nums=[1, 2])
    np.savez_compressed(os.path.join(OUT, "pc_drift.npz"), xt=xt.numpy(), latent=latent.numpy(), t=int(t), T=T,
                        mask=mask.numpy(), xtm1=xtm1.numpy(), x0_pred=x0_pred.numpy(), drift=drift.numpy(),
                        alphas_cumprod=sched.alphas_cumprod.numpy(), **rec)
    print("pc", float(np.abs(rec["eigval3"]).max()), tuple(drift.shape))



# --------------------------------------------------------------------------- PC extract / apply CLIs
def gen_pc_cli():
    """Run the reference's OWN scripts main_pc_extract_inv.py and main_pc_apply_drift.py end to end with a synthetic
    eps-model (load_model / load_audio / plotting / wandb / torchaudio replaced by stand-ins; the one line that
    names the device, f"cuda:{args.device_num}", is replaced by "cpu" in memory before exec -- nothing is stored)."""
    import glob
    import tempfile
    from unittest import mock
    UOut = install_stubs()
    sys.modules["wandb"] = mock.MagicMock()
    sys.modules["torchaudio"] = mock.MagicMock()
    sys.path.insert(0, REF)
    import models as ref_models
    import utils as ref_utils

    class FakeRef(ref_models.PipelineWrapper):
        get_variance = ref_models.AudioLDM2Wrapper.get_variance
        get_alpha_prod_t_prev = ref_models.AudioLDM2Wrapper.get_alpha_prod_t_prev

        def __init__(self, T):
            super().__init__(model_id="fake", device=torch.device("cpu"))
            sched = OracleDDIMScheduler()
            sched.set_timesteps(T)
            self.model = SimpleNamespace(scheduler=sched, unet=SimpleNamespace(config=SimpleNamespace(in_channels=8)),
                                         vocoder=SimpleNamespace(config=SimpleNamespace(model_in_dim=64)),
                                         vae_scale_factor=4)

        def encode_text(self, prompts, **kw):
            return None, torch.stack([prompt_vec(p) for p in prompts]), None

        def unet_forward(self, sample, timestep, encoder_hidden_states=None, class_labels=None,
                         encoder_attention_mask=None, **kw):
            return UOut(sample=synthetic_unet(sample, timestep, class_labels)), None, None

        def get_fn_STFT(self):
            return None

        def vae_encode(self, x):
            return x

        def vae_decode(self, x):
            return x

        def decode_to_mel(self, x):
            return torch.zeros(len(x), 16)

    T = 12
    g = torch.Generator().manual_seed(33)
    w0 = torch.randn((1, 8, 16, 16), generator=g) * 0.7
    ref_models.load_model = lambda *a, **k: FakeRef(T)
    ref_utils.load_audio = lambda *a, **k: w0.clone()
    ref_utils.plot_corrs = lambda *a, **k: None
    real_load = torch.load

    def run_script(name, argv):
        src = open(os.path.join(REF, name)).read()
        assert 'f"cuda:{args.device_num}"' in src
        src = src.replace('f"cuda:{args.device_num}"', '"cpu"')
        glb = {"__name__": "__main__", "__file__": os.path.join(REF, name)}
        old_argv = sys.argv
        sys.argv = [name] + argv
        try:
            with mock.patch("torch.cuda.set_device"), mock.patch("torch.cuda.empty_cache"), \
                    mock.patch("torch.load", lambda f, *a, **k: real_load(f, map_location="cpu", weights_only=False)), \
                    mock.patch("matplotlib.pyplot.imsave"), mock.patch("torch.use_deterministic_algorithms"):
                exec(compile(src, name, "exec"), glb)
        finally:
            sys.argv = old_argv
        return glb

    rec = {"w0": w0.numpy(), "T": T}
    with tempfile.TemporaryDirectory() as tmp:
        for tag, swap in (("a", "0.8"), ("b", "-2.0")):        # b: corr <= 2 always -> every PC sign is flipped
            out_dir = os.path.join(tmp, tag)
            run_script("main_pc_extract_inv.py",
                       ["--init_aud", "synth.wav", "--num_diffusion_steps", str(T), "--source_prompt", "a dog barking",
                        "--drift_start", "8", "--drift_end", "4", "--n_evs", "2", "--iters", "4", "-s", "5",
                        "--patch", "2", "12", "--corr_to_swap", swap, "--results_path", out_dir, "--wandb_disable"])
            pts = glob.glob(os.path.join(out_dir, "**", "*.pt"), recursive=True)
            assert len(pts) == 1, pts
            ck = real_load(pts[0], map_location="cpu", weights_only=False)
            ts = sorted(ck["eigdata"].keys(), reverse=True)
            rec[f"{tag}_ts"] = np.array(ts)
            rec[f"{tag}_eigvec"] = np.stack([ck["eigdata"][t]["eigvec"].numpy() for t in ts])
            rec[f"{tag}_eigval"] = np.stack([ck["eigdata"][t]["eigval"].numpy() for t in ts])
            rec[f"{tag}_it"] = np.array([ck["eigdata"][t]["it"] for t in ts])
            rec[f"{tag}_norm_factor"] = np.array([float(ck["eigdata"][t]["norm_factor"]) for t in ts])
            rec[f"{tag}_corrs"] = np.stack([c.numpy() for c in ck["corrs"]])
            rec[f"{tag}_xts"] = np.concatenate([x.numpy() for x in ck["xts"]])
            rec[f"{tag}_latents"] = np.concatenate([x.numpy() for x in ck["latents"]])
            if tag == "a":
                keys = sorted(ck.keys())
                rec["ckpt_keys"] = np.array(keys)
                rec["eigdata_keys"] = np.array(sorted(ck["eigdata"][ts[0]].keys()))
                rec["args_fields"] = np.array(sorted(vars(ck["args"]).keys()))
                evals = {t: ck["eigdata"][t]["eigval"].numpy() for t in ts}
                ev_path = os.path.join(tmp, "eigvals.pt")
                torch.save(evals, ev_path)
                # the reference's apply script reads extraction_args.target_prompt, which its extractor never writes
                ck["args"].target_prompt = ck["args"].source_prompt
                torch.save(ck, pts[0])
                base = ["--extraction_path", pts[0], "--drift_start", "8", "--drift_end", "4", "--amount", "2.0",
                        "--evals_pt", ev_path, "-s", "9", "--wandb_disable"]
                glb = run_script("main_pc_apply_drift.py", base + ["--evs", "1", "2"])
                rec["apply_sep"] = glb["xt"].detach().numpy()
                glb = run_script("main_pc_apply_drift.py", base + ["--evs", "1"])
                rec["apply_single"] = glb["xt"].detach().numpy()
                glb = run_script("main_pc_apply_drift.py", base + ["--evs", "1", "2", "--combine_evs", "--fix_alpha", "0.3",
                                                                  "--fade_length", "2.0"])
                rec["apply_comb_fix"] = glb["xt"].detach().numpy()
        # SDEdit baseline script on the same synthetic model
        glb = run_script("main_run_sdedit.py",
                         ["--init_aud", "synth.wav", "--num_diffusion_steps", str(T), "--target_prompt", "a cat meowing",
                          "--tstart", "7", "--cfg_tar", "5", "-s", "11", "--results_path", os.path.join(tmp, "sd"),
                          "--wandb_disable"])
        rec["sdedit_xt"] = glb["xt"].detach().numpy()
        rec["sdedit_tstart"] = 7
    np.savez_compressed(os.path.join(OUT, "pc_cli.npz"), **rec)
    print("pc_cli", rec["a_ts"], rec["a_eigval"][0], rec["apply_sep"].shape, rec["apply_comb_fix"].shape)

    # ---- the headline script itself, main_run.py (SURVEY 8a A1-A17 glue: tstart / skip handling, zs slicing, cfg lists,
    # multi-prompt segments with unequal tstart, the DDIM mode), same harness
    FakeRef.get_sr = lambda self: 16000
    ref_utils.load_audio = lambda *a, **k: (w0.clone(), 16000, 10.0)
    mr = {"w0": w0.numpy(), "T": T}
    with tempfile.TemporaryDirectory() as tmp:
        base = ["--init_aud", "synth.wav", "--model_id", "cvssp/audioldm2", "--num_diffusion_steps", str(T),
                "--results_path", tmp]
        glb = run_script("main_run.py", base + ["--source_prompt", "a dog barking", "--target_prompt", "a cat meowing",
                                                "--tstart", "7", "--cfg_src", "3", "--cfg_tar", "12", "-s", "21"])
        mr["a_w_edit"], mr["a_zs"], mr["a_wts"] = glb["w0"].numpy(), glb["zs"].numpy(), glb["wts"].numpy()
        glb = run_script("main_run.py", base + ["--source_prompt", "rain", "--target_prompt", "jazz", "rock",
                                                "--tstart", "7", "5", "--cfg_src", "3", "--cfg_tar", "12", "8",
                                                "--cutoff_points", "0.5", "--fix_alpha", "0.2", "-s", "22"])
        mr["b_w_edit"] = glb["w0"].numpy()
        glb = run_script("main_run.py", base + ["--source_prompt", "a dog barking", "--target_prompt", "a cat meowing",
                                                "--tstart", "9", "--cfg_src", "3", "--cfg_tar", "12", "--mode", "ddim",
                                                "-s", "23"])
        mr["c_w_edit"], mr["c_wT"] = glb["w0"].numpy(), glb["wT"].numpy()
    np.savez_compressed(os.path.join(OUT, "main_run.npz"), **mr)
    print("main_run", [tuple(mr[k].shape) for k in ("a_w_edit", "b_w_edit", "c_w_edit")])


# --------------------------------------------------------------------------- audio
def _load_by_path(modname, path):
    spec = importlib.util.spec_from_file_location(modname, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def gen_audio():
    install_stubs()

    def slaney_mel(sr, n_fft, n_mels, fmin, fmax):
        # librosa 0.9.2 filters.mel defaults: htk=False (Slaney scale), norm="slaney".
        fb = mel_filter_bank(num_frequency_bins=n_fft // 2 + 1, num_mel_filters=n_mels, min_frequency=fmin,
                             max_frequency=fmax, sampling_rate=sr, norm="slaney", mel_scale="slaney")
        return fb.T.astype(np.float32)

    _stub("librosa")
    _stub("librosa.util", pad_center=lambda w, size=None, **kw: w, tiny=lambda x: np.finfo(np.float32).tiny,
          normalize=lambda x, norm=None: x)
    _stub("librosa.filters", mel=slaney_mel)
    pkg = types.ModuleType("audioldm")
    pkg.__path__ = [os.path.join(REF, "audioldm")]
    sys.modules["audioldm"] = pkg
    sub = types.ModuleType("audioldm.audio")
    sub.__path__ = [os.path.join(REF, "audioldm", "audio")]
    sys.modules["audioldm.audio"] = sub
    _load_by_path("audioldm.audio.audio_processing", os.path.join(REF, "audioldm/audio/audio_processing.py"))
    stft = _load_by_path("audioldm.audio.stft", os.path.join(REF, "audioldm/audio/stft.py"))
    fn = stft.TacotronSTFT(1024, 160, 1024, 64, 16000, 0, 8000)
    n = 160 * 64  # 64 frames' worth: small fixture, same code path
    wav = chirp_waveform(n=n, seed=1234)
    wav = wav - wav.mean()
    wav = wav / (wav.abs().max() + 1e-8) * 0.5
    mel, logmag, energy = fn.mel_spectrogram(wav[None])
    mag, _ = fn.stft_fn.transform(wav[None])
    np.savez_compressed(os.path.join(OUT, "stft_mel_64f.npz"), wav=wav.numpy(), mel=mel.numpy(), mag=mag.numpy(),
                        energy=energy.numpy(), mel_basis=fn.mel_basis.numpy(),
                        basis_rows=fn.stft_fn.forward_basis[[0, 1, 7, 512, 513, 514, 900, 1025], 0, :].numpy(),
                        basis_row_ids=np.array([0, 1, 7, 512, 513, 514, 900, 1025]))
    print("audio", tuple(mel.shape), float(mel.mean()))
    # ---- A1: the reference's own waveform preparation (audioldm/audio/tools.py:18-85) run here.  torchaudio is absent;
    # its two calls (load, functional.resample) are replaced by an in-memory 16 kHz waveform -- everything after them
    # (normalize_wav, pad_wav incl. its float64 zero-pad, the second normalisation, get_mel_from_wav, _pad_spec) is the
    # reference's code.
    ta = sys.modules["torchaudio"]
    tools = _load_by_path("audioldm.audio.tools", os.path.join(REF, "audioldm/audio/tools.py"))
    rec = {}
    for name, n_in, frames in (("short", 160 * 40 + 37, 64), ("long", 160 * 80 + 11, 64), ("exact", 160 * 64, 64)):
        raw = (chirp_waveform(n=n_in, seed=77).numpy() * 1.7 + 0.2).astype(np.float32)      # offset + gain: exercises both norms
        ta.load = lambda fn, _w=raw: (torch.from_numpy(_w)[None], 16000)
        ta.functional = types.SimpleNamespace(resample=lambda w, orig_freq, new_freq: w)
        wave = tools.read_wav_file("in-memory", frames * 160)
        fbank, logmag, wav_t = tools.wav_to_fbank("in-memory", target_length=frames, fn_STFT=fn)
        rec.update({f"{name}.raw": raw, f"{name}.wave": np.asarray(wave), f"{name}.fbank": fbank.numpy(),
                    f"{name}.wav_t": wav_t.numpy(), f"{name}.frames": frames})
    rec["pad_spec.in"] = torch.arange(7 * 65, dtype=torch.float32).reshape(7, 65).numpy()
    rec["pad_spec.out10"] = tools._pad_spec(torch.from_numpy(rec["pad_spec.in"]), 10).numpy()
    rec["pad_spec.out5"] = tools._pad_spec(torch.from_numpy(rec["pad_spec.in"]), 5).numpy()
    np.savez_compressed(os.path.join(OUT, "waveform_prep.npz"), **rec)
    print("waveform prep", {k: v.shape for k, v in rec.items() if hasattr(v, "shape")})


# --------------------------------------------------------------------------- hifigan
def gen_hifigan():
    from transformers import SpeechT5HifiGan, SpeechT5HifiGanConfig
    cfg = SpeechT5HifiGanConfig(model_in_dim=64, sampling_rate=16000, upsample_initial_channel=64,
                                upsample_rates=[5, 4, 2, 2, 2], upsample_kernel_sizes=[16, 16, 8, 4, 4],
                                resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3,
                                normalize_before=False)
    torch.manual_seed(0)
    voc = SpeechT5HifiGan(cfg).eval()
    sd = {}
    g = torch.Generator().manual_seed(11)
    for k, v in voc.state_dict().items():
        if v.dtype.is_floating_point and "mean" not in k and "scale" not in k:
            v.copy_(torch.randn(v.shape, generator=g) * (0.08 if "weight" in k else 0.02))
        sd[k] = v.numpy().copy()
    mel = torch.randn((2, 24, 64), generator=g) * 2.0 - 4.0
    with torch.no_grad():
        wav = voc(mel)
    np.savez_compressed(os.path.join(OUT, "hifigan_c64.npz"), mel=mel.numpy(), wav=wav.numpy(),
                        **{"sd." + k: v for k, v in sd.items()})
    print("hifigan", tuple(wav.shape), float(wav.abs().mean()))
    # cross-check against the in-tree twin with the same weights (two independent implementations)
    install_stubs()
    pkg = types.ModuleType("audioldm_h")
    sys.modules["audioldm_h"] = pkg
    twin = _load_by_path("audioldm_h.models", os.path.join(REF, "audioldm/hifigan/models.py"))
    h = SimpleNamespace(resblock_kernel_sizes=[3, 7, 11], upsample_rates=[5, 4, 2, 2, 2],
                        upsample_kernel_sizes=[16, 16, 8, 4, 4], upsample_initial_channel=64,
                        resblock_dilation_sizes=[[1, 3, 5]] * 3, num_mels=64)
    gen = twin.Generator(h).eval()
    gen.remove_weight_norm()
    tsd = {}
    for k, v in sd.items():
        k2 = k.replace("upsampler.", "ups.")
        if k2 in gen.state_dict():
            tsd[k2] = torch.from_numpy(v)
    gen.load_state_dict(tsd, strict=True)
    with torch.no_grad():
        wav2 = gen(mel.transpose(1, 2)).squeeze(1)
    print("hifigan twin max|diff| =", float((wav2 - wav).abs().max()))


# --------------------------------------------------------------------------- unet / vae twins
def _load_twin_pkg():
    install_stubs()
    pkg = types.ModuleType("audioldm")
    pkg.__path__ = [os.path.join(REF, "audioldm")]
    sys.modules["audioldm"] = pkg
    _load_by_path("audioldm.utils", os.path.join(REF, "audioldm/utils.py"))
    ld = types.ModuleType("audioldm.latent_diffusion")
    ld.__path__ = [os.path.join(REF, "audioldm/latent_diffusion")]
    sys.modules["audioldm.latent_diffusion"] = ld
    _load_by_path("audioldm.latent_diffusion.util", os.path.join(REF, "audioldm/latent_diffusion/util.py"))
    _load_by_path("audioldm.latent_diffusion.attention", os.path.join(REF, "audioldm/latent_diffusion/attention.py"))
    return _load_by_path("audioldm.latent_diffusion.openaimodel",
                         os.path.join(REF, "audioldm/latent_diffusion/openaimodel.py"))


def gen_unet():
    oai = _load_twin_pkg()
    torch.manual_seed(0)
    cfg = dict(image_size=64, extra_film_condition_dim=24, extra_film_use_concat=True, in_channels=8,
               out_channels=8, model_channels=32, attention_resolutions=[2], num_res_blocks=2,
               channel_mult=[1, 2], num_head_channels=16, use_spatial_transformer=True)
    net = oai.UNetModel(**cfg).eval()
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for n_, p in net.named_parameters():
            if p.dim() > 1:
                p.copy_(torch.randn(p.shape, generator=g) * (1.5 / (p[0].numel() ** 0.5)))
            elif "norm" in n_ or n_.endswith("0.weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
    x = torch.randn((2, 8, 32, 8), generator=g)
    y = torch.randn((2, 24), generator=g)
    t = torch.tensor([501, 37])
    with torch.no_grad():
        out = net(x, t, y=y)
    sd = {k: v.numpy() for k, v in net.state_dict().items()}
    np.savez_compressed(os.path.join(OUT, "unet_twin_c32.npz"), x=x.numpy(), y=y.numpy(), t=t.numpy(),
                        out=out.numpy(), **{"sd." + k: v for k, v in sd.items()})
    print("unet twin", tuple(out.shape), float(out.abs().mean()), "params", sum(p.numel() for p in net.parameters()))


# --------------------------------------------------------------------------- the reference's INLINE U-Net forward graphs
def gen_unet_graph():
    """Pins SURVEY row A8's graph level: the reference's own `PipelineWrapper.unet_forward` (models.py:160-393) and
    `AudioLDM2Wrapper.unet_forward` (models.py:691-899) are EXECUTED here, unbound, on a stand-in `self.model.unet` that
    exposes the diffusers attribute surface those two functions touch (time_proj / time_embedding / class_embedding /
    conv_in / down_blocks[i](...) -> (sample, res_samples) / mid_block / up_blocks[i].resnets + __call__ / conv_norm_out /
    conv_act / conv_out / num_upsamplers / config).  The stand-in BLOCKS compute with oracle/unet.py's block functions (pinned
    by the in-tree twin, unet_twin_c32.npz); everything BETWEEN the blocks -- the time / class-embedding prelude, mask -> bias
    conversion, which residuals an up block receives (`down_block_res_samples[-len(resnets):]`), replace_h_space /
    mid_block_additional_residual / replace_skip_conns / zero_out_resconns (int and list) semantics, the forward_upsample_size
    rule and the returned (sample, h_space, extracted_res_conns) -- is the reference's code running.  Weights are
    weights.random_state_dict(unet_param_shapes(tiny family), seed) so consumers regenerate them; the fixture holds inputs
    and outputs only."""
    UOut = install_stubs()
    sys.path.insert(0, REF)
    import models as ref_models
    import torch.nn.functional as F
    from audioeditingcode_amd import configs, weights
    from oracle import unet as ou

    class Block:
        """One diffusers down / mid / up block as the reference calls it (keyword surface of models.py:303-372, :806-885)."""

        def __init__(self, cfg, sd, kind, i):
            self.cfg, self.sd, self.kind, self.i = cfg, sd, kind, i
            boc = cfg["block_out_channels"]
            nb = len(boc)
            self.lpb = cfg.get("layers_per_block", 2)
            self.groups, self.eps = cfg.get("norm_num_groups", 32), cfg.get("norm_eps", 1e-5)
            heads = ou._per_block(cfg.get("num_attention_heads") or cfg.get("attention_head_dim", 8), nb)
            ctx_pb, self.multi = ou._ctx_list(cfg)
            lvl = {"down": i, "mid": nb - 1, "up": nb - 1 - i}[kind]
            self.heads, self.dims = heads[lvl], ctx_pb[lvl]
            types = {"down": cfg["down_block_types"], "up": cfg["up_block_types"]}.get(kind)
            self.has_cross_attention = True if kind == "mid" else "CrossAttn" in types[i]
            self.prefix = "mid_block" if kind == "mid" else f"{kind}_blocks.{i}"
            self.last = i == nb - 1
            self.resnets = [None] * (self.lpb + 1 if kind == "up" else self.lpb)      # the reference reads len(...) only
            self.linear, self.depth = cfg.get("use_linear_projection", False), cfg.get("transformer_layers_per_block", 1)

        def _site(self, k0, x, ehs, bias, ehs1, bias1):
            for j, cdim in enumerate(self.dims):
                if cdim is None:
                    ctx, b, dbl = None, None, True
                elif self.multi and j > 1:
                    ctx, b, dbl = ehs1, bias1, False
                else:
                    ctx, b, dbl = ehs, bias, False
                x = ou.transformer2d(self.sd, f"{self.prefix}.attentions.{k0 + j}", x, ctx, self.heads, self.groups, b, dbl,
                                     self.linear, self.depth)
            return x

        def __call__(self, hidden_states=None, temb=None, res_hidden_states_tuple=None, upsample_size=None,
                     encoder_hidden_states=None, attention_mask=None, cross_attention_kwargs=None,
                     encoder_attention_mask=None, encoder_hidden_states_1=None, encoder_attention_mask_1=None):
            assert attention_mask is None and cross_attention_kwargs is None
            h, sd = hidden_states, self.sd
            site = lambda k0, x: self._site(k0, x, encoder_hidden_states, encoder_attention_mask,      # noqa: E731
                                            encoder_hidden_states_1, encoder_attention_mask_1)
            if self.kind == "mid":
                h = ou.resnet(sd, "mid_block.resnets.0", h, temb, self.groups, self.eps)
                h = site(0, h)
                return ou.resnet(sd, "mid_block.resnets.1", h, temb, self.groups, self.eps)
            if self.kind == "down":
                outs = ()
                for j in range(self.lpb):
                    h = ou.resnet(sd, f"{self.prefix}.resnets.{j}", h, temb, self.groups, self.eps)
                    if self.has_cross_attention:
                        h = site(j * len(self.dims), h)
                    outs += (h,)
                if not self.last:
                    h = ou._conv(sd, f"{self.prefix}.downsamplers.0.conv", h, stride=2, padding=1)
                    outs += (h,)
                return h, outs
            res = tuple(res_hidden_states_tuple)
            for j in range(self.lpb + 1):
                h = torch.cat([h, res[-1]], dim=1)
                res = res[:-1]
                h = ou.resnet(sd, f"{self.prefix}.resnets.{j}", h, temb, self.groups, self.eps)
                if self.has_cross_attention:
                    h = site(j * len(self.dims), h)
            if not self.last:                  # diffusers Upsample2D: nearest, scale 2 unless an output size is forwarded
                h = F.interpolate(h, scale_factor=2.0, mode="nearest") if upsample_size is None else \
                    F.interpolate(h, size=tuple(upsample_size), mode="nearest")
                h = ou._conv(sd, f"{self.prefix}.upsamplers.0.conv", h)
            return h

    def stand_in_unet(cfg, sd):
        boc = cfg["block_out_channels"]
        nb = len(boc)
        lin = lambda pfx: (lambda x: ou._lin(sd, pfx, x))                                         # noqa: E731
        return SimpleNamespace(
            config=SimpleNamespace(center_input_sample=False, class_embed_type=cfg.get("class_embed_type"),
                                   class_embeddings_concat=bool(cfg.get("class_embeddings_concat")), addition_embed_type=None,
                                   encoder_hid_dim_type=None, in_channels=cfg["in_channels"]),
            num_upsamplers=nb - 1,
            time_proj=lambda t: ou.timestep_embedding(t, boc[0], cfg.get("flip_sin_to_cos", True), cfg.get("freq_shift", 0)),
            time_embedding=lambda t_emb, cond=None: ou._lin(sd, "time_embedding.linear_2",
                                                            F.silu(ou._lin(sd, "time_embedding.linear_1", t_emb))),
            class_embedding=lin("class_embedding") if cfg.get("class_embed_type") is not None else None,
            time_embed_act=None, encoder_hid_proj=None, add_embedding=None,
            conv_in=lambda x: ou._conv(sd, "conv_in", x),
            down_blocks=[Block(cfg, sd, "down", i) for i in range(nb)], mid_block=Block(cfg, sd, "mid", 0),
            up_blocks=[Block(cfg, sd, "up", i) for i in range(nb)],
            conv_norm_out=lambda x: ou._gn(sd, "conv_norm_out", x, cfg.get("norm_num_groups", 32), cfg.get("norm_eps", 1e-5)),
            conv_act=F.silu, conv_out=lambda x: ou._conv(sd, "conv_out", x))

    rec = {}
    g = torch.Generator().manual_seed(11)
    rn = lambda *s: torch.randn(*s, generator=g)                                                  # noqa: E731
    for fam_name, fn, seed in (("audioldm", ref_models.PipelineWrapper.unet_forward, 21),
                               ("tango", ref_models.PipelineWrapper.unet_forward, 22),
                               ("audioldm2", ref_models.AudioLDM2Wrapper.unet_forward, 23)):
        fam = configs.tiny_family(fam_name)
        cfg = fam["unet"]
        sd = weights.random_state_dict(weights.unet_param_shapes(cfg), seed=seed)
        me = SimpleNamespace(model=SimpleNamespace(unet=stand_in_unet(cfg, sd)))
        boc = cfg["block_out_channels"]
        B = 2
        for size_name, (H, W) in (("even", (16, 8)), ("odd", (20, 6))):       # 20 x 6 is not a multiple of 8: forward_upsample_size
            x = rn(B, 8, H, W)
            t = 481
            if fam_name == "audioldm":
                cond = dict(encoder_hidden_states=None, class_labels=F.normalize(rn(B, fam["ctx"]["clap_dim"]), dim=-1))
            elif fam_name == "tango":
                mask = torch.ones(B, 5, dtype=torch.long)
                mask[1, 3:] = 0
                cond = dict(encoder_hidden_states=rn(B, 5, fam["ctx"]["t5_dim"]), encoder_attention_mask=mask)
            else:
                mask = torch.ones(B, 6, dtype=torch.long)
                mask[0, 4:] = 0
                cond = dict(encoder_hidden_states=rn(B, 8, fam["ctx"]["gpt2_dim"]), class_labels=rn(B, 6, fam["ctx"]["t5_dim"]),
                            encoder_attention_mask=mask)
            with torch.no_grad():
                out, h_space, skips = fn(me, x, t, **cond)
            key = f"{fam_name}.{size_name}"
            rec[f"{key}.x"], rec[f"{key}.t"] = x.numpy(), np.int64(t)
            for k, v in cond.items():
                if v is not None:
                    rec[f"{key}.cond.{k}"] = v.numpy()
            rec[f"{key}.plain.eps"], rec[f"{key}.plain.h_space"] = out.sample.numpy(), h_space.numpy()
            for i, lst in skips.items():
                for j, s_ in enumerate(lst):
                    rec[f"{key}.plain.skip.{i}.{j}"] = s_.numpy()
            if size_name == "odd":
                continue
            # hooks (same inputs): values a caller could pass (pc_drift / the h-space scripts of the reference)
            hs_new = rn(*h_space.shape) * 0.5
            add = rn(*h_space.shape) * 0.1
            rep = {1: [rn(*s_.shape) * 0.3 for s_ in skips[1]]}
            hooks = dict(replace_h_space=dict(replace_h_space=hs_new), mid_add=dict(mid_block_additional_residual=add),
                         replace_skips=dict(replace_skip_conns=rep), zero_int=dict(zero_out_resconns=3),
                         zero_list=dict(zero_out_resconns=[0, 2]),
                         combined=dict(replace_h_space=hs_new, mid_block_additional_residual=add, zero_out_resconns=[1]))
            rec[f"{key}.hook.h_space_new"], rec[f"{key}.hook.mid_add"] = hs_new.numpy(), add.numpy()
            for j, r_ in enumerate(rep[1]):
                rec[f"{key}.hook.replace.1.{j}"] = r_.numpy()
            for hname, kw in hooks.items():
                with torch.no_grad():
                    o2, h2, s2 = fn(me, x, t, **cond, **kw)
                rec[f"{key}.{hname}.eps"], rec[f"{key}.{hname}.h_space"] = o2.sample.numpy(), h2.numpy()
                rec[f"{key}.{hname}.skip_abs_sums"] = np.array([[float(s_.abs().sum()) for s_ in s2[i]] for i in sorted(s2)])
            print("unet graph", key, tuple(out.sample.shape), "h_space", tuple(h_space.shape),
                  "skips", {i: len(v) for i, v in skips.items()}, float(out.sample.abs().mean()))
        rec[f"{fam_name}.seed"] = np.int64(seed)
    np.savez_compressed(os.path.join(OUT, "unet_forward_graph.npz"), **rec)
    print("unet_forward_graph.npz keys", len(rec), "bytes", os.path.getsize(os.path.join(OUT, "unet_forward_graph.npz")))


def gen_vae():
    _load_twin_pkg()
    va = types.ModuleType("audioldm.variational_autoencoder")
    va.__path__ = [os.path.join(REF, "audioldm/variational_autoencoder")]
    sys.modules["audioldm.variational_autoencoder"] = va
    _load_by_path("audioldm.variational_autoencoder.distributions",
                  os.path.join(REF, "audioldm/variational_autoencoder/distributions.py"))
    mods = _load_by_path("audioldm.variational_autoencoder.modules",
                         os.path.join(REF, "audioldm/variational_autoencoder/modules.py"))
    dd = dict(double_z=True, z_channels=8, resolution=256, in_channels=1, out_ch=1, ch=32, ch_mult=[1, 1, 2],
              num_res_blocks=2, attn_resolutions=[], dropout=0.0)
    torch.manual_seed(0)
    enc = mods.Encoder(**dd).eval()
    dec = mods.Decoder(**dd).eval()
    quant = torch.nn.Conv2d(16, 16, 1)
    post = torch.nn.Conv2d(8, 8, 1)
    g = torch.Generator().manual_seed(9)
    with torch.no_grad():
        for mod in (enc, dec, quant, post):
            for n_, p in mod.named_parameters():
                if p.dim() > 1:
                    p.copy_(torch.randn(p.shape, generator=g) * (1.2 / (p[0].numel() ** 0.5)))
                elif "norm" in n_ and n_.endswith("weight"):
                    p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
                else:
                    p.copy_(0.1 * torch.randn(p.shape, generator=g))
        mel = torch.randn((1, 1, 64, 32), generator=g)
        moments = quant(enc(mel))
        mean = moments[:, :8]
        recon = dec(post(mean))
    rec = {}
    for pre, mod in (("encoder.", enc), ("decoder.", dec), ("quant_conv.", quant), ("post_quant_conv.", post)):
        for k, v in mod.state_dict().items():
            rec["sd." + pre + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "vae_twin_c32.npz"), mel=mel.numpy(), mean=mean.numpy(),
                        recon=recon.numpy(), **rec)
    print("vae twin", tuple(mean.shape), tuple(recon.shape), float(recon.abs().mean()))


# --------------------------------------------------------------------------- Stable Audio wrapper math + loops
def gen_stable_audio():
    """Runs the reference's StableAudWrapper methods (models.py:1069-1354) and its loops on a 3-D latent.  diffusers is
    absent: the scheduler is oracle.stable_audio.OracleCosineDPMSolverScheduler (restated), the tokenizer / text
    encoder / projection model / transformer are small deterministic stand-ins -- what is pinned is the reference's own
    code: encode_text (negative / empty-prompt / double-mask rules), unet_forward (context assembly, zeroing),
    sample_xts_from_x0, setup_extra_inputs (history re-seeding), get_zs_from_xts, reverse_step_with_custom_noise."""
    install_stubs()
    sys.path.insert(0, REF)
    import models as ref_models
    from ddm_inversion import inversion_utils as ref_inv
    from oracle import stable_audio as osa

    ref_models.get_1d_rotary_pos_embed = lambda dim, n, use_real=True, repeat_interleave_real=False: osa.rotary_table(dim, n)
    SI = osa.StandIns
    C, Lz, D, S = SI.C, SI.Lz, SI.D, SI.S
    Tok, TextEnc, projection_model, encode_duration, transformer = (SI.Tok, SI.TextEnc, SI.projection_model,
                                                                    SI.encode_duration, SI.transformer)

    class FakeSA(ref_models.StableAudWrapper):
        def __init__(self, sched, T):
            ref_models.PipelineWrapper.__init__(self, model_id="fake/stable-audio", device=torch.device("cpu"))
            sched.set_timesteps(T)
            self.model = SimpleNamespace(
                scheduler=sched, tokenizer=Tok(), text_encoder=TextEnc(), projection_model=projection_model,
                encode_duration=encode_duration, rotary_embed_dim=4, transformer=transformer,
                vae=SimpleNamespace(hop_length=16, config=SimpleNamespace(sampling_rate=100, audio_channels=2)))
            self.model.transformer.config = SimpleNamespace(in_channels=C, sample_size=Lz)

    rec = {}

    def run_case(name, T, tstart, src, tgt, cfg_src, cfg_tar, seed, first_order=False):
        sched = osa.OracleCosineDPMSolverScheduler()
        m = FakeSA(sched, T)
        g = torch.Generator().manual_seed(300 + seed)
        x0 = torch.randn((1, C, Lz), generator=g) * 0.8
        torch.manual_seed(seed)
        with torch.no_grad():
            _, zs, wts, extra = ref_inv.inversion_forward_process(
                m, x0, etas=1.0, prompts=[src], cfg_scales=[cfg_src], num_inference_steps=T, numerical_fix=True,
                duration=3.0, first_order=first_order)
            w0, _ = ref_inv.inversion_reverse_process(
                m, xT=wts, tstart=torch.tensor([tstart], dtype=torch.int), etas=1.0, prompts=[tgt], neg_prompts=[""],
                cfg_scales=[cfg_tar], zs=zs[:tstart], duration=3.0, extra_info=extra, first_order=first_order)
        torch.manual_seed(seed)
        sched2 = osa.OracleCosineDPMSolverScheduler()
        xts_init = FakeSA(sched2, T).sample_xts_from_x0(x0, num_inference_steps=T)
        ex = torch.stack([e if e is not None else torch.full_like(x0, float("nan")) for e in extra])
        rec.update({f"{name}.x0": x0.numpy(), f"{name}.xts_init": xts_init.numpy(), f"{name}.zs": zs.numpy(),
                    f"{name}.xts": wts.numpy(), f"{name}.extra": ex.numpy(), f"{name}.w_edit": w0.numpy(),
                    f"{name}.meta": np.array([T, tstart, cfg_src, cfg_tar, seed, int(first_order)], dtype=np.float64),
                    f"{name}.src": np.array(src), f"{name}.tgt": np.array(tgt),
                    f"{name}.sigmas": sched.sigmas.numpy(), f"{name}.timesteps": sched.timesteps.numpy()})
        print("sa loop", name, tuple(zs.shape), "recon err", float((w0 - x0).abs().max()),
              "finite:", bool(torch.isfinite(w0).all()))

    run_case("T20", 20, 12, "a dog barking", "a cat meowing", 1.0, 7.0, seed=0)
    run_case("T20_emptysrc", 20, 20, "", "rain on a roof", 1.0, 4.0, seed=1)
    run_case("T12_first", 12, 8, "rain", "jazz", 3.0, 6.0, seed=2, first_order=True)      # also T < 15: lower_order_final

    # encode_text / unet_forward semantics on their own
    m = FakeSA(osa.OracleCosineDPMSolverScheduler(), 8)
    for tag, prompts, neg in (("pos", ["a dog barking"], False), ("neg", ["low quality"], True), ("empty", [""], True)):
        e, _, mask = m.encode_text(prompts, negative=neg)
        rec[f"enc.{tag}.embeds"] = e.numpy()
        rec[f"enc.{tag}.mask"] = (mask.numpy() if mask is not None else np.zeros(0))
    raw_ids = m.model.tokenizer(["a dog barking"], max_length=S)
    rec["enc.raw"] = m.model.text_encoder(raw_ids.input_ids)[0].numpy()
    x = torch.randn((1, C, Lz), generator=torch.Generator().manual_seed(5))
    m.setup_extra_inputs(x, init_timestep=m.model.scheduler.timesteps[0], audio_end_in_s=2.5)
    e, _, mask = m.encode_text(["a dog barking"])
    rec["fwd.x"] = x.numpy()
    rec["fwd.cond"] = m.unet_forward(x, m.model.scheduler.timesteps[3], e, encoder_attention_mask=mask)[0].sample.numpy()
    rec["fwd.uncond"] = m.unet_forward(x, m.model.scheduler.timesteps[3], e, encoder_attention_mask=None)[0].sample.numpy()
    rec["fwd.wave_window"] = np.array([m.waveform_start, m.waveform_end])
    # --- step math at chosen steps of the T=200 schedule (kernel targets; the coefficient rows travel with the fixture,
    #     as in step_math_T200.npz, because host table arithmetic differs in the last bit between CPUs)
    from audioeditingcode_amd.scheduler import CosineDPMSolverMultistepScheduler, sa_step_coefficients
    T = 200
    gg = torch.Generator().manual_seed(77)
    cases = [(0, 1), (1, 2), (57, 2), (150, 2), (198, 2), (199, 1)]
    rec["step.index_order"] = np.array(cases)
    for k, (i, order) in enumerate(cases):
        m = FakeSA(osa.OracleCosineDPMSolverScheduler(), T)
        sch = m.model.scheduler
        xt, xtm1, v, m1, zin = (torch.randn((1, C, Lz), generator=gg) * sc for sc in (1.0 + float(sch.sigmas[i]),
                                                                                    1.0 + float(sch.sigmas[i + 1]), 1.0, 1.0, 1.0))
        def arm():
            sch._step_index = i
            sch.model_outputs = [None, m1.clone() if order == 2 else None]
            sch.lower_order_nums = 0 if i == 0 else 2
        arm()
        z, xfix, ex = m.get_zs_from_xts(xt, xtm1.clone(), v, sch.timesteps[i], numerical_fix=True)
        d = sch.model_outputs[-1].clone()
        arm()
        prev = m.reverse_step_with_custom_noise(v, sch.timesteps[i], xt, variance_noise=zin)
        ps = CosineDPMSolverMultistepScheduler()
        ps.set_timesteps(T)
        assert torch.equal(ps.sigmas, sch.sigmas)
        coef = sa_step_coefficients(ps, i, order, zero_z=(i == T - 1))
        for name, val in (("xt", xt), ("xtm1", xtm1), ("v", v), ("m1", m1), ("z_in", zin), ("z", z), ("xfix", xfix),
                          ("d", d), ("prev", prev), ("coef", coef)):
            rec[f"step.{name}{k}"] = val.numpy()
    # --- the raw-waveform branch of the reference's own utils.load_audio (utils.py:77-95): torchaudio's two calls are the
    #     only stand-ins (load -> an in-memory [channels, n] tensor; resample -> scipy's polyphase resampler, the product's
    #     fallback, so the resampled case pins the call and everything after it, not the resampler)
    _stub("matplotlib")
    _stub("matplotlib.pyplot")
    ta = sys.modules["torchaudio"]
    store = {}
    ta.load = lambda path: store[path]

    def _resample(w, orig_freq, new_freq):
        from scipy.signal import resample_poly
        import math as _m
        g_ = _m.gcd(int(orig_freq), int(new_freq))
        return torch.from_numpy(np.stack([resample_poly(c, int(new_freq) // g_, int(orig_freq) // g_).astype(np.float32)
                                          for c in w.numpy()]))
    ta.functional = SimpleNamespace(resample=_resample)
    ref_utils = _load_by_path("ref_utils_for_sa", os.path.join(REF, "utils.py"))
    gw = torch.Generator().manual_seed(21)
    for tag, ch, n_in, sr_in, sr_model in (("stereo_same_sr", 2, 5000, 44100, 44100), ("mono_resampled", 1, 4000, 16000, 44100)):
        w_in = torch.randn(ch, n_in, generator=gw) * 0.2 + 0.05
        store[tag] = (w_in.clone(), sr_in)
        w_out, sr_out, dur = ref_utils.load_audio(tag, None, stft=False, model_sr=sr_model)
        rec[f"load.{tag}.in"] = w_in.numpy()
        rec[f"load.{tag}.out"] = w_out.numpy()
        rec[f"load.{tag}.meta"] = np.array([sr_in, sr_model, sr_out, dur], dtype=np.float64)
        print("load_audio", tag, tuple(w_out.shape), sr_out, dur)
    np.savez_compressed(os.path.join(OUT, "sa_wrapper.npz"), **rec)
    print("sa_wrapper.npz keys", len(rec))


# --------------------------------------------------------------------------- A15: the wrappers' encode_text
def gen_text():
    """Runs the reference's OWN `AudioLDMWrapper.encode_text` (models.py:511-537), `AudioLDM2Wrapper.encode_text`
    (:599-677) and `TangoWrapper.encode_text` (:455-472) with the stand-in pipeline members of oracle/text_standins.py
    (real transformers CLAP / T5 / GPT-2 classes, random-init at reduced width; diffusers' projection model /
    generate_language_model and tango's encode_text restated -- those packages are absent)."""
    from oracle import text_standins as ts
    # transformers' lazy imports must resolve BEFORE the name-only stubs go in (accelerate probes `wandb.__spec__`)
    clap_t, clap, t5, lm, t5_tango = (ts.clap_text_with_projection(), ts.clap_model_v4_api(), ts.t5_encoder(), ts.gpt2(),
                                      ts.t5_encoder(seed=16))
    pw = ts.projection_weights()
    install_stubs()
    sys.path.insert(0, REF)
    import models as ref_models

    def bare(cls, model):
        w = cls.__new__(cls)                        # the constructors download checkpoints: bypass them
        ref_models.PipelineWrapper.__init__(w, model_id="fake", device=torch.device("cpu"))
        w.model = model
        return w

    rec = {}
    # ---- AudioLDM-1
    m1 = bare(ref_models.AudioLDMWrapper, SimpleNamespace(tokenizer=ts.ClapWordTokenizer(), text_encoder=clap_t))
    # ---- AudioLDM2
    pipe2 = SimpleNamespace(
        tokenizer=ts.ClapWordTokenizer(), tokenizer_2=ts.T5WordTokenizer(), text_encoder=clap, text_encoder_2=t5,
        language_model=lm, projection_model=lambda **kw: ts.audioldm2_projection_forward(pw, **kw),
        generate_language_model=lambda embeds, attention_mask=None, max_new_tokens=None:
            ts.generate_language_model(lm, embeds, attention_mask, max_new_tokens))
    m2 = bare(ref_models.AudioLDM2Wrapper, pipe2)
    # ---- TANGO
    tok_tango = ts.T5WordTokenizer()
    m3 = bare(ref_models.TangoWrapper, SimpleNamespace(encode_text=lambda p: ts.tango_encode_text(tok_tango, t5_tango, p)))
    rec["checksum"] = np.array([ts.param_checksum(x) for x in (clap_t, clap, t5, lm, t5_tango)])
    for k, prompts in enumerate(ts.PROMPT_SETS):
        with torch.no_grad():
            _, cl, _ = m1.encode_text(list(prompts))
            gen, t5s, mask = m2.encode_text(list(prompts), negative=(prompts == [""]))
            th, none, tm = m3.encode_text(list(prompts))
        assert none is None
        rec[f"audioldm.{k}.class_labels"] = cl.numpy()
        rec[f"audioldm2.{k}.generated"] = gen.numpy()
        rec[f"audioldm2.{k}.t5"] = t5s.numpy()
        rec[f"audioldm2.{k}.mask"] = mask.numpy()
        rec[f"tango.{k}.states"] = th.numpy()
        rec[f"tango.{k}.mask"] = tm.numpy()
        print("text", k, prompts, "clap", tuple(cl.shape), "gen", tuple(gen.shape), "t5", tuple(t5s.shape),
              "mask", mask.tolist(), "tango", tuple(th.shape), tm.dtype)
    np.savez_compressed(os.path.join(OUT, "text_encode.npz"), **rec)
    print("text_encode.npz keys", len(rec))


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    # each generator runs in a fresh interpreter when "all" (the stubs of one break another)
    if what == "all":
        import subprocess
        for w in ("loops", "pc", "pc_cli", "audio", "hifigan", "unet", "unet_graph", "vae", "stable_audio", "text"):
            subprocess.check_call([sys.executable, os.path.abspath(__file__), w])
    else:
        {"loops": gen_loops, "pc": gen_pc, "pc_cli": gen_pc_cli, "audio": gen_audio, "hifigan": gen_hifigan, "unet": gen_unet, "unet_graph": gen_unet_graph, "vae": gen_vae,
         "stable_audio": gen_stable_audio, "text": gen_text}[what]()

"""Wrapper API of the reference's code/models.py, MI355X-native underneath (THE drop-in boundary,
SURVEY 8b).  Class names, method names, argument meaning and error behaviour mirror
/root/reference/code/models.py (PipelineWrapper :14-393, TangoWrapper :396-472, AudioLDMWrapper :475-549,
AudioLDM2Wrapper :552-899, load_model :1357-1374) so that main_run.py / main_pc_*.py style callers run
unchanged; every tensor-valued method executes in libaed.so (HIP, gfx950) -- there is no torch/diffusers
model underneath and no CPU fallback.

Tensors cross this boundary exactly as in the reference: NCHW fp32 torch tensors on `device`.
"""
import ctypes
import os
import sys
import weakref
import zlib
from types import SimpleNamespace
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib as L
from . import configs, weights
from . import tape as tape_mod
from .codec import STFTEngine, VAEDecoder, VAEEncoder, VocoderEngine
from .editing import Conditioning, EditEngine
from .scheduler import (CosineDPMSolverMultistepScheduler, DDIMScheduler, sa_step_coefficients, sa_step_orders,
                        step_coefficients)
from .unet import PackedUNetWeights, UNetEngine


class UNet2DConditionOutput(SimpleNamespace):
    """Stand-in for diffusers' output dataclass: callers only read `.sample`."""


class _LazySkips(dict):
    """`{up_block_index: [skip tensors, NCHW]}` of unet_forward's third return value (models.py:854-866).  The engine keeps
    skips channels-last; the NCHW copies are made the first time a block's list is read, so calls that never look at the
    skip connections (every step of the plain loops) do not pay for 12 transposes."""

    def __init__(self, groups):
        super().__init__({i: None for i in groups})
        self._groups = groups

    def __getitem__(self, i):
        v = super().__getitem__(i)
        if v is None:
            v = [b.permute(0, 3, 1, 2).contiguous() for b in self._groups[i]]
            super().__setitem__(i, v)
        return v

    def get(self, i, default=None):
        return self[i] if i in self._groups else default

    def values(self):
        return [self[i] for i in self._groups]

    def materialize(self):
        """Copy every block's skips now (called before the engine's buffers are overwritten by its next forward)."""
        for i in self._groups:
            self[i]

    def items(self):
        return [(i, self[i]) for i in self._groups]


class _FnSTFT:
    """`TacotronSTFT` duck type returned by get_fn_STFT (audioldm/audio/stft.py:130-180)."""

    def __init__(self, owner):
        self.owner = owner
        self.n_mel_channels = owner.family["stft"]["n_mel_channels"]
        self.sampling_rate = owner.family["stft"]["sampling_rate"]

    def mel_spectrogram(self, y, normalize_fun=torch.log):
        assert torch.min(y) >= -1, torch.min(y)
        assert torch.max(y) <= 1, torch.max(y)
        eng = self.owner._stft(y.shape[0], y.shape[1])
        mel = eng(y)                                                   # [B, F, n_mels]
        mag = eng.mag[:, : eng.n_fft // 2 + 1].view(y.shape[0], eng.frames, -1)
        log_mag = torch.log(torch.clamp(mag, min=1e-5)).transpose(1, 2)
        energy = torch.norm(mag, dim=2)
        return mel.transpose(1, 2).contiguous(), log_mag, energy


class PipelineWrapper(torch.nn.Module):
    family_name = None
    # arithmetic of the U-Net engines' LDS-staged GEMMs (editing.EditEngine.arith): "bf16x6" = exact three-way bf16 split of the
    # fp32 operands, six bf16-MFMA piece products, fp32 accumulate (results as close to fp64 as the fp32-MFMA chain's);
    # "f32" = fp32-input MFMAs everywhere.  Set on the class or on an instance BEFORE the first editor() call of a shape.
    arith = "bf16x6"
    # arithmetic of the codec engines' LDS-staged GEMMs (STFT-as-DFT, VAE, vocoder; `_cached`), independent of `arith`.
    # bf16x6 since round 4: VAE decode + two vocoder passes 75 -> 42 ms on the inversion partition's queue, every codec / e2e
    # parity tolerance unchanged (profiles/r04_pipeline_variants.md)
    codec_arith = "bf16x6"

    def __init__(self, model_id: str, device: torch.device, double_precision: bool = False,
                 token: Optional[str] = None, seed: int = 0, state_dicts: Optional[Dict] = None,
                 allow_synthetic: Optional[bool] = None, *args, **kwargs) -> None:
        super().__init__()
        self._init_common(model_id, device, double_precision, token, allow_synthetic)
        self.family = configs.get_family(model_id)
        self.kind = self.family["ctx"]["kind"]
        ckpt = weights.find_checkpoint(model_id) if state_dicts is None else None
        if state_dicts is not None:          # e.g. received through dist.broadcast_state_dicts
            unet_sd, vae_sd, voc_sd = state_dicts["unet"], state_dicts["vae"], state_dicts["vocoder"]
            self.weights_source = "caller-provided state dicts"
        elif ckpt is not None:
            comp = weights.load_checkpoint(ckpt)
            self.family["unet"], unet_sd = comp["unet"]
            self.family["vae"], vae_sd = comp["vae"]
            self.family["vocoder"], voc_sd = comp["vocoder"]
            if "scheduler" in comp:
                self.family["scheduler"] = comp["scheduler"]
            self.weights_source = ckpt
            try:
                from .text_encoders import TextEncoders
                self.text_encoders = TextEncoders.from_pretrained(ckpt, self.kind, device=self.device)
                self.conditioning_source = f"transformers text encoders from {ckpt}"
            except (FileNotFoundError, OSError, ImportError) as e:
                if not self.synthetic_ok:
                    raise L.AedError(f"{ckpt}: U-Net/VAE/vocoder weights were found but the text-conditioning "
                                     f"components could not be loaded ({e}); prompts would have no effect. Pass "
                                     f"allow_synthetic=True (or AED_ALLOW_SYNTHETIC=1) to run with stand-in "
                                     f"embeddings anyway.") from e
        else:
            if not self.synthetic_ok:
                raise L.AedError(f"no checkpoint for '{model_id}' on disk (weights.find_checkpoint) and none can be "
                                 f"downloaded here. Seeded-random weights are for benchmarks and parity tests only: "
                                 f"pass allow_synthetic=True to load_model, set AED_ALLOW_SYNTHETIC=1, or use "
                                 f"--allow_synthetic on the CLIs.")
            unet_sd = weights.random_state_dict(weights.unet_param_shapes(self.family["unet"]), seed=seed)
            vae_sd = weights.random_state_dict(weights.vae_param_shapes(self.family["vae"]), seed=seed + 1)
            voc_sd = weights.random_state_dict(weights.vocoder_param_shapes(self.family["vocoder"]), seed=seed + 2)
            self.weights_source = f"seeded-random(seed={seed})"
        self.state_dicts = dict(unet=unet_sd, vae=vae_sd, vocoder=voc_sd)
        self.unet_weights = PackedUNetWeights(unet_sd, self.device)
        ucfg, vcfg, ocfg = self.family["unet"], self.family["vae"], self.family["vocoder"]
        n_up = sum(1 for _ in ucfg["up_block_types"]) - 1
        self.model = SimpleNamespace(
            scheduler=None,
            unet=SimpleNamespace(config=SimpleNamespace(**{k: v for k, v in ucfg.items()}), num_upsamplers=n_up),
            vae=SimpleNamespace(config=SimpleNamespace(scaling_factor=vcfg["scaling_factor"])),
            vocoder=SimpleNamespace(config=SimpleNamespace(**ocfg)),
            vae_scale_factor=2 ** (len(vcfg["block_out_channels"]) - 1))
        self._engines = {}
        self._editors = {}

    def _init_common(self, model_id, device, double_precision, token, allow_synthetic) -> None:
        """Preamble shared by every wrapper: the synthetic opt-in, the fp32-only rule, the HIP-only rule, libaed.so."""
        # Seeded-random weights and stand-in text embeddings exist for benchmarks and parity tests (no checkpoint can be
        # downloaded here).  They must never be used silently: a run that edits audio with them exits 0 and writes
        # noise.  They need an explicit opt-in: allow_synthetic=True, AED_ALLOW_SYNTHETIC=1, or a "tiny/" test model.
        if allow_synthetic is None:
            allow_synthetic = os.environ.get("AED_ALLOW_SYNTHETIC") == "1" or model_id.startswith("tiny/")
        self.synthetic_ok = bool(allow_synthetic)
        self.text_encoders = None
        self.conditioning_source = "unset"
        if double_precision:
            raise NotImplementedError("double_precision=True: the native path is fp32 (the reference's default)")
        self.model_id = model_id
        self.device = torch.device(device)
        self.double_precision = double_precision
        self.token = token
        self._require_device()
        L.lib()                                                       # fail loudly if libaed.so is missing

    def _require_device(self) -> None:
        """The product runs on HIP only.  (The CPU test suite subclasses the wrapper and overrides this hook to execute
        the wrapper's HOST logic on tapes interpreted by oracle/tape_interp.py; nothing in the product does.)"""
        if self.device.type != "cuda":
            raise L.AedError(f"device={self.device}: the product path needs an MI355X (HIP); there is no CPU fallback. "
                             f"The CPU restatement lives in oracle/ and is test infrastructure only.")

    # ------------------------------------------------------------------ engine caches
    def _cached(self, key, make, arith=None):
        # codec engines (STFT-as-DFT, VAE, vocoder) are built under tape.arith_mode(self.codec_arith); the single-call U-Net /
        # DiT engines of `unet_forward` (hook path, PC power iteration: finite differences of the U-Net) stay on fp32 MFMAs,
        # the arithmetic their tolerances were observed with
        arith = self.codec_arith if arith is None else arith
        key = key if arith == "f32" else key + (arith,)
        if key not in self._engines:
            with tape_mod.arith_mode(arith):
                self._engines[key] = make()
        return self._engines[key]

    def _stft(self, B, N):
        return self._cached(("stft", B, N), lambda: STFTEngine(self.family["stft"], self.device, B, N))

    def _vae_enc(self, B, T, F):
        return self._cached(("enc", B, T, F), lambda: VAEEncoder(self.family["vae"], self.state_dicts["vae"],
                                                                   self.device, B, T, F))

    def _vae_dec(self, B, h, w):
        return self._cached(("dec", B, h, w), lambda: VAEDecoder(self.family["vae"], self.state_dicts["vae"],
                                                                   self.device, B, h, w))

    def _vocoder(self, B, T):
        return self._cached(("voc", B, T), lambda: VocoderEngine(self.family["vocoder"], self.state_dicts["vocoder"],
                                                                  self.device, B, T))

    def editor(self, H, W) -> EditEngine:
        """Device-resident loop engine for latents of size HxW (used by ddm_inversion.*)."""
        if (H, W) not in self._editors:
            self._editors[(H, W)] = EditEngine(self.family["unet"], self.unet_weights, self.model.scheduler,
                                               self.device, H, W, self.kind)
        ed = self._editors[(H, W)]
        ed.sched = self.model.scheduler
        # inside a clip pipeline (pipeline.ClipPipeline) the loops replay on the caller's CU-partition lane
        ed.lane_stream = self.__dict__.get("_lane_stream")
        ed.eager_steps = bool(self.__dict__.get("_lane_eager"))       # lanes issue their steps launch by launch
        ed.lane_chooser = self.__dict__.get("_lane_chooser")          # per-chunk lane choice of a pipeline's edit loop
        ed.capture_lock = self.__dict__.get("_capture_lock")          # graph captures vs another thread's launches on the lane
        ed.arith = self.arith
        if hasattr(self, "arith_min_batch"):           # smallest U-Net batch whose engine takes `arith` (default: every engine)
            ed.ARITH_MIN_BATCH = int(self.arith_min_batch)
        return ed

    def lane_view(self):
        """A second handle on this model for a concurrent lane (pipeline.ClipPipeline): shares everything frozen --
        packed U-Net weights, codec state dicts, scheduler tables, text encoders, configuration -- and owns everything
        mutable: its own engine caches (activations, loop plans, hipGraphs) and noise-prefetch state."""
        import copy
        v = copy.copy(self)
        v._engines, v._editors = {}, {}
        v._noise_box = None
        v.next_noise_seed = None
        return v

    # ------------------------------------------------------------------ reference API
    def get_sigma(self, timestep: int) -> float:
        return torch.sqrt(1.0 / self.model.scheduler.alphas_cumprod - 1)[timestep]

    def load_scheduler(self) -> None:
        self.model.scheduler = DDIMScheduler.from_config(self.family["scheduler"])

    def get_fn_STFT(self):
        return _FnSTFT(self)

    def get_sr(self) -> int:
        return 16000

    def vae_encode(self, x: torch.Tensor) -> torch.Tensor:
        if x.shape[2] % 4:                                              # front pad (models.py:583-584)
            x = torch.nn.functional.pad(x, (0, 0, 4 - (x.shape[2] % 4), 0))
        B, _, T, F = x.shape
        enc = self._vae_enc(B, T, F)
        lat = enc(x)                                                    # [B, h, w, C] channels-last
        return lat.permute(0, 3, 1, 2).contiguous().float()

    def vae_decode(self, x: torch.Tensor) -> torch.Tensor:
        B, C, h, w = x.shape
        dec = self._vae_dec(B, h, w)
        mel = dec(x.to(self.device, torch.float32).permute(0, 2, 3, 1).contiguous())   # [B, T, F, 1]
        return mel.permute(0, 3, 1, 2).contiguous()

    def decode_to_mel(self, x: torch.Tensor) -> torch.Tensor:
        mel = x[:, 0].detach().float()                                  # [B, T, n_mels]
        voc = self._vocoder(mel.shape[0], mel.shape[1])
        wav = voc(mel).detach().to("cpu", torch.float32, copy=True)     # a CPU tensor (never a view of the engine buffer)
        if len(wav.shape) == 1:
            wav = wav.unsqueeze(0)
        return wav

    def setup_extra_inputs(self, *args, **kwargs) -> None:
        pass

    def encode_text(self, prompts: List[str], **kwargs):
        raise NotImplementedError

    def get_variance(self, timestep, prev_timestep):
        s = self.model.scheduler
        a_t = s.alphas_cumprod[int(timestep)]
        a_p = self.get_alpha_prod_t_prev(prev_timestep)
        return ((1 - a_p) / (1 - a_t)) * (1 - a_t / a_p)

    def get_alpha_prod_t_prev(self, prev_timestep):
        s = self.model.scheduler
        return s.alphas_cumprod[int(prev_timestep)] if prev_timestep >= 0 else s.final_alpha_cumprod

    def get_noise_shape(self, x0: torch.Tensor, num_steps: int) -> Tuple[int, ...]:
        return (num_steps, self.model.unet.config.in_channels, x0.shape[-2], x0.shape[-1])

    def sample_xts_from_x0(self, x0: torch.Tensor, num_inference_steps: int = 50) -> torch.Tensor:
        """Samples from P(x_1:T|x_0) (models.py:67-83); noise from the torch CPU generator in the reference's
        draw order, arithmetic on the device."""
        ed = self.editor(x0.shape[-2], x0.shape[-1])
        x = x0.reshape(1, *x0.shape[-3:])
        noise = self._take_prefetched_noise(tuple(x.shape), num_inference_steps)
        nxt = getattr(self, "next_noise_seed", None)
        if nxt is not None:            # a serving loop announced the next clip's seed: draw its noise under this clip
            self.next_noise_seed = None
            self.prefetch_noise(nxt, x.shape, num_inference_steps)
        return ed.sample_xts(x, noise=noise)[:, 0]

    # ------------------------------------------------------------------ x_t noise prefetch (serving loops)
    def prefetch_noise(self, seed: int, shape, num_inference_steps: int) -> None:
        """Draw the T x_t noise maps of the NEXT clip on a host thread while the current clip runs on the GPU.

        Same generator algorithm, same seed, same draw order as the in-line path (`torch.manual_seed(seed)` followed by
        one `randn` per timestep, models.py:76-81): a fresh CPU `torch.Generator` seeded with `seed` yields the identical
        stream, and the worker makes the same T sequential `randn(shape)` calls (one stacked `randn` of T shapes equals them
        only when the per-step element count is a multiple of 16 -- not relied on).  The ~40 ms of
        single-threaded CPU RNG per 10 s clip then overlap the previous clip's edit loop.  The buffer is consumed by
        the next `sample_xts_from_x0` call with the same shape / T and ONLY if the global CPU generator is in the state
        `torch.manual_seed(seed)` leaves it in (so a caller that seeded differently gets the in-line draws)."""
        import threading
        shape = (1, *tuple(shape)[-3:])
        box = {"seed": int(seed), "shape": shape, "T": int(num_inference_steps), "noise": None}

        def work():
            g = torch.Generator().manual_seed(int(seed))
            n = torch.empty((int(num_inference_steps), *shape), dtype=torch.float32)
            for k in range(int(num_inference_steps)):
                n[k] = torch.randn(shape, generator=g, dtype=torch.float32)
            box["state_after"] = g.get_state()
            box["noise"] = n.pin_memory() if torch.cuda.is_available() else n
        box["thread"] = threading.Thread(target=work, daemon=True)
        box["thread"].start()
        self._noise_box = box

    def _take_prefetched_noise(self, shape, num_inference_steps):
        box = getattr(self, "_noise_box", None)
        if box is None:
            return None
        self._noise_box = None
        box["thread"].join()
        if box["shape"] != tuple(shape) or box["T"] != int(num_inference_steps):
            return None
        # only valid if the caller's global generator sits exactly where manual_seed(seed) puts it
        ref = torch.Generator().manual_seed(box["seed"])
        if not torch.equal(torch.get_rng_state(), ref.get_state()):
            return None
        torch.set_rng_state(box["state_after"])        # the global generator ends where the in-line draws would leave it
        return box["noise"]

    def get_zs_from_xts(self, xt, xtm1, noise_pred, t, eta: float = 0, numerical_fix: bool = True, **kwargs):
        c = step_coefficients(self.model.scheduler, int(t), eta=eta)
        cf = (ctypes.c_float * L.COEF_STRIDE)(*c.tolist())
        xt, noise_pred = xt.contiguous(), noise_pred.contiguous()
        xtm1 = xtm1.contiguous().clone()
        z = torch.empty_like(xt)
        vp = int(self.model.scheduler.config.prediction_type == "v_prediction")
        L.check(L.lib().aed_get_zs_from_xts(xt.data_ptr(), xtm1.data_ptr(), noise_pred.data_ptr(), None, None, 0.0, 0,
                                            cf, vp, int(bool(numerical_fix)), z.data_ptr(), None, xt.numel(),
                                            L.current_stream_ptr()), "aed_get_zs_from_xts")
        return z, xtm1, None

    def reverse_step_with_custom_noise(self, model_output, timestep, sample, variance_noise=None, eta: float = 0,
                                       **kwargs):
        c = step_coefficients(self.model.scheduler, int(timestep), eta=eta)
        cf = (ctypes.c_float * L.COEF_STRIDE)(*c.tolist())
        model_output, sample = model_output.contiguous(), sample.contiguous()
        if eta > 0 and variance_noise is None:
            variance_noise = torch.randn(model_output.shape).to(self.device)
        z = variance_noise.contiguous() if eta > 0 else None
        prev = torch.empty_like(sample)
        vp = int(self.model.scheduler.config.prediction_type == "v_prediction")
        L.check(L.lib().aed_reverse_step_with_custom_noise(sample.data_ptr(), model_output.data_ptr(), None, None, 0.0,
                                                           0, cf, vp, None if z is None else z.data_ptr(),
                                                           prev.data_ptr(), sample.numel(), L.current_stream_ptr()),
                "aed_reverse_step_with_custom_noise")
        return prev

    # ------------------------------------------------------------------ unet_forward (models.py:160-393, :691-899)
    def _cond_from_args(self, B, encoder_hidden_states, class_labels, encoder_attention_mask):
        """Map the reference's positional conditioning triple to engine inputs."""
        if self.kind == "audioldm2":      # (generated GPT-2 states, T5 states as class_labels, T5 mask)
            return dict(ehs0=encoder_hidden_states, ehs1=class_labels,
                        bias1=None if encoder_attention_mask is None else
                        (1 - encoder_attention_mask.float()) * -10000.0), encoder_hidden_states.shape[1], \
                class_labels.shape[1]
        if self.kind == "audioldm":
            return dict(class_labels=class_labels), 0, 0
        return dict(ehs0=encoder_hidden_states,
                    bias0=None if encoder_attention_mask is None else
                    (1 - encoder_attention_mask.float()) * -10000.0), encoder_hidden_states.shape[1], 0

    def unet_forward(self, sample, timestep, encoder_hidden_states=None, class_labels=None, timestep_cond=None,
                     attention_mask=None, cross_attention_kwargs=None, added_cond_kwargs=None,
                     down_block_additional_residuals=None, mid_block_additional_residual=None,
                     encoder_attention_mask=None, replace_h_space=None, replace_skip_conns=None,
                     return_dict: bool = True, zero_out_resconns=None):
        B, C, H, W = sample.shape
        cond, L0, L1 = self._cond_from_args(B, encoder_hidden_states, class_labels, encoder_attention_mask)
        eng = self._cached(("unet", B, H, W, L0, L1),
                           lambda: UNetEngine(self.family["unet"], self.unet_weights, self.device, B, H, W,
                                              ctx_len0=L0, ctx_len1=L1, use_ehs=self.kind != "audioldm"),
                           arith="f32")
        pending = getattr(eng, "_pending_skips", None)
        pending = pending() if pending is not None else None
        if pending is not None:            # a previous call's result is still held by the caller: its views go stale now
            pending.materialize()
        eng.set_conditioning(**{k: v for k, v in cond.items() if v is not None})
        eng.x_in.copy_(sample.to(self.device, torch.float32).permute(0, 2, 3, 1))
        eng.set_timestep(int(timestep))
        eng.forward(first_half_only=True)
        if replace_h_space is None:
            h_space = eng.h_space.permute(0, 3, 1, 2).contiguous()
        else:
            h_space = replace_h_space
            eng.h_space.copy_(replace_h_space.to(self.device).permute(0, 2, 3, 1).expand_as(eng.h_space))
        if mid_block_additional_residual is not None:
            eng.h_space.add_(mid_block_additional_residual.to(self.device).permute(0, 2, 3, 1))
        nres = self.family["unet"].get("layers_per_block", 2) + 1
        groups, sk = {}, list(eng.skips)
        for i in range(len(self.family["unet"]["up_block_types"])):
            grp, sk = sk[-nres:], sk[:-nres]
            if replace_skip_conns is not None and replace_skip_conns.get(i):
                for buf, rep in zip(grp, replace_skip_conns.get(i)):
                    buf.copy_(rep.to(self.device).permute(0, 2, 3, 1))
            if zero_out_resconns is not None:
                if (type(zero_out_resconns) is int and i >= (zero_out_resconns - 1)) or \
                        (type(zero_out_resconns) is list and i in zero_out_resconns):
                    for buf in grp:
                        buf.zero_()
            groups[i] = grp
        extracted = _LazySkips(groups)
        eng._pending_skips = weakref.ref(extracted)
        eng.forward(second_half_only=True)
        out = eng.eps.permute(0, 3, 1, 2).contiguous()
        if not return_dict:
            return (out,)
        return UNet2DConditionOutput(sample=out), h_space, extracted


    def unet_forward_pair(self, sample_u, sample_c, timestep, cond_u: Conditioning, cond_c: Conditioning):
        """CFG pair in ONE batched U-Net call: rows [sample_u | sample_c] with their own conditioning
        (ragged contexts are padded with zero-weight keys).  Returns (eps_u, eps_c) NCHW."""
        n, C, H, W = sample_u.shape
        ed = self.editor(H, W)
        groups = [cond_u.repeat(n), cond_c.repeat(n)]
        L0, L1 = ed._ctx_lens(groups)
        eng = ed.unet(2 * n, L0, L1)
        ed._set_cond(eng, groups)
        with torch.inference_mode():
            eng.x_in[:n].copy_(sample_u.to(self.device, torch.float32).permute(0, 2, 3, 1))
            eng.x_in[n:].copy_(sample_c.to(self.device, torch.float32).permute(0, 2, 3, 1))
        op = eng.tape.ops[eng.time_op]
        arr = eng.tape.finalize()
        for o in (op, arr[eng.time_op]):        # immediate timestep (no device table) for this stand-alone call
            o.p[1] = None
            o.p[2] = None
            o.p[4] = None
            o.i[4] = int(timestep)
        eng.forward()
        eps = eng.eps.permute(0, 3, 1, 2).contiguous()
        return eps[:n], eps[n:]


# --------------------------------------------------------------------------------------------------
def _prompt_generator(prompt, salt):
    return torch.Generator().manual_seed(zlib.crc32((salt + "|" + prompt).encode("utf-8")))


class _SyntheticText:
    """Deterministic stand-in for the text encoders (conditioning tensors are INPUTS to the
    measured path; CLAP / T5 / GPT-2 weights do not exist in this container).  Shapes follow the real
    encoders: T5 length = #whitespace tokens + 1 (EOS), so '' -> length 1 like `padding=True` tokenisation."""

    @staticmethod
    def t5(prompts, dim):
        lens = [len(p.split()) + 1 for p in prompts]
        Lm = max(lens)
        e = torch.zeros(len(prompts), Lm, dim)
        m = torch.zeros(len(prompts), Lm, dtype=torch.long)
        for i, (p, l) in enumerate(zip(prompts, lens)):
            e[i, :l] = torch.randn(l, dim, generator=_prompt_generator(p, "t5"))
            m[i, :l] = 1
        return e, m

    @staticmethod
    def vec(prompts, dim, salt):
        return torch.stack([torch.randn(dim, generator=_prompt_generator(p, salt)) for p in prompts])


def _require_synthetic_text(w):
    """Stand-in text embeddings need the explicit opt-in (see PipelineWrapper.__init__)."""
    if not w.synthetic_ok:
        raise L.AedError("encode_text: no text encoders are loaded (wrapper.text_encoders is None), so prompts would be "
                         "replaced by prompt-seeded random tensors. Load a checkpoint with its tokenizer/text_encoder "
                         "directories, set wrapper.text_encoders = TextEncoders(...), or opt in with allow_synthetic.")
    if w.conditioning_source != "synthetic":
        w.conditioning_source = "synthetic"
        print(f"[audioeditingcode_amd] WARNING: {w.model_id}: text conditioning is SYNTHETIC (prompt-seeded random "
              f"tensors of the real shapes); weights: {w.weights_source}", file=sys.stderr, flush=True)


class AudioLDMWrapper(PipelineWrapper):
    family_name = "audioldm"

    def encode_text(self, prompts: List[str], **kwargs):
        """(None, L2-normalised CLAP text embedding [P,512], None) -- models.py:511-537."""
        enc = getattr(self, "text_encoders", None)
        if enc is not None:
            return enc.encode_audioldm(prompts, self.device, **kwargs)
        _require_synthetic_text(self)
        v = torch.nn.functional.normalize(_SyntheticText.vec(prompts, self.family["ctx"]["clap_dim"], "clap"), dim=-1)
        return None, v.to(self.device), None


class AudioLDM2Wrapper(PipelineWrapper):
    family_name = "audioldm2"

    def encode_text(self, prompts: List[str], **kwargs):
        """(GPT-2 generated states [P,8,768], T5 states [P,L,1024], T5 mask [P,L]) -- models.py:599-677."""
        enc = getattr(self, "text_encoders", None)
        if enc is not None:
            return enc.encode_audioldm2(prompts, self.device, **kwargs)
        _require_synthetic_text(self)
        c = self.family["ctx"]
        gen = torch.stack([torch.randn(c["gpt2_len"], c["gpt2_dim"], generator=_prompt_generator(p, "gpt2"))
                           for p in prompts])
        t5, mask = _SyntheticText.t5(prompts, c["t5_dim"])
        return gen.to(self.device), t5.to(self.device), mask.to(self.device)


class TangoWrapper(PipelineWrapper):
    family_name = "tango"

    def vae_encode(self, x: torch.Tensor) -> torch.Tensor:
        if x.shape[2] % 4:
            x = torch.nn.functional.pad(x, (0, 0, 4 - (x.shape[2] % 4), 0))
        if x.shape[2] > 1700:
            raise RuntimeWarning("This model dies at this point")        # models.py:444-445
        return super().vae_encode(x)

    def encode_text(self, prompts: List[str], **kwargs):
        """(T5 states [P,L,1024], None, mask [P,L]) -- models.py:462-467."""
        enc = getattr(self, "text_encoders", None)
        if enc is not None:
            return enc.encode_tango(prompts, self.device, **kwargs)
        _require_synthetic_text(self)
        t5, mask = _SyntheticText.t5(prompts, self.family["ctx"]["t5_dim"])
        return t5.to(self.device), None, mask.to(self.device)

class StableAudWrapper(PipelineWrapper):
    """StableAudWrapper of the reference (models.py:1051-1354): Stable Audio Open 1.0 -- DiT backbone on a 3-D latent
    [1, 64, 1024], CosineDPMSolver++ (SDE, order 2) inversion / edit, Oobleck VAE on raw 44.1 kHz stereo.  Same method
    names and argument meaning; every tensor-valued method runs in libaed.so.  `self.model` has no `unet` attribute,
    which is how the reference's loops pick the 3-D expands (inversion_utils.py:88-89)."""
    # DiT engines: split-bf16 GEMMs like the U-Net families since its full-length run against the oracle fixture (round 4: edited
    # latent 3.7e-6 from the CPU oracle at T=200 in both arithmetics, 10.2 s per clip against 13.6 s)

    family_name = "stable_audio"
    codec_arith = "f32"         # Oobleck encoder / decoder: fp32 MFMAs (not measured under the split-bf16 kernel yet)

    def __init__(self, model_id: str, device: torch.device, double_precision: bool = False,
                 token: Optional[str] = None, seed: int = 0, state_dicts: Optional[Dict] = None,
                 allow_synthetic: Optional[bool] = None, *args, **kwargs) -> None:
        torch.nn.Module.__init__(self)
        self._init_common(model_id, device, double_precision, token, allow_synthetic)
        self.family = configs.get_family(model_id)
        self.kind = "stable_audio"
        ckpt = weights.find_checkpoint(model_id) if state_dicts is None else None
        if state_dicts is not None:
            dit_sd, vae_sd, proj_sd = state_dicts["transformer"], state_dicts["vae"], state_dicts["projection_model"]
            self.weights_source = "caller-provided state dicts"
        elif ckpt is not None:
            comp = weights.load_stable_audio_checkpoint(ckpt)
            self.family["dit"], dit_sd = comp["transformer"]
            self.family["oobleck"], vae_sd = comp["vae"]
            self.family["projection"], proj_sd = comp["projection_model"]
            self.family["ctx"]["t5_dim"] = self.family["projection"].get("text_encoder_dim", self.family["ctx"]["t5_dim"])
            if "scheduler" in comp:
                self.family["scheduler"] = comp["scheduler"]
            self.weights_source = ckpt
            try:
                from .text_encoders import TextEncoders
                self.text_encoders = TextEncoders.from_pretrained(ckpt, "stable_audio", device=self.device)
                self.conditioning_source = f"transformers text encoder from {ckpt}"
            except (FileNotFoundError, OSError, ImportError) as e:
                if not self.synthetic_ok:
                    raise L.AedError(f"{ckpt}: DiT/VAE weights were found but the text-conditioning components could "
                                     f"not be loaded ({e}); pass allow_synthetic=True to run with stand-ins.") from e
        else:
            if not self.synthetic_ok:
                raise L.AedError(f"no checkpoint for '{model_id}' on disk (weights.find_checkpoint) and none can be "
                                 f"downloaded here. Seeded-random weights are for benchmarks and parity tests only: "
                                 f"pass allow_synthetic=True to load_model, set AED_ALLOW_SYNTHETIC=1, or use "
                                 f"--allow_synthetic on the CLIs.")
            dit_sd = weights.random_state_dict(weights.dit_param_shapes(self.family["dit"]), seed=seed)
            vae_sd = weights.random_state_dict(weights.oobleck_param_shapes(self.family["oobleck"]), seed=seed + 1)
            proj_sd = weights.random_state_dict(weights.projection_param_shapes(self.family["projection"]), seed=seed + 2)
            self.weights_source = f"seeded-random(seed={seed})"
        self.state_dicts = dict(transformer=dit_sd, vae=vae_sd, projection_model=proj_sd)
        from .stable_audio import PackedDiTWeights
        self.dit_weights = PackedDiTWeights(dit_sd, self.family["dit"], self.device)
        if self.text_encoders is None:
            from .text_encoders import StableAudioProjection
            self._projection = StableAudioProjection(**self.family["projection"])
            self._projection.load_diffusers_state_dict(proj_sd)
        dcfg, ocfg = self.family["dit"], self.family["oobleck"]
        hop = 1
        for r in ocfg["downsampling_ratios"]:
            hop *= r
        self.model = SimpleNamespace(
            scheduler=None,
            transformer=SimpleNamespace(config=SimpleNamespace(**dcfg)),
            vae=SimpleNamespace(hop_length=hop, config=SimpleNamespace(sampling_rate=ocfg["sampling_rate"],
                                                                       audio_channels=ocfg["audio_channels"])),
            rotary_embed_dim=dcfg["attention_head_dim"] // 2)
        self._engines = {}
        self._editor = None
        self.waveform_start, self.waveform_end = 0, dcfg["sample_size"] * hop
        self.seconds_start_hidden_states = self.seconds_end_hidden_states = self.audio_duration_embeds = None
        self._hist = None

    # ------------------------------------------------------------------ engines
    def editor(self, *unused):
        from .stable_audio import StableAudioEditEngine
        if self._editor is None:
            self._editor = StableAudioEditEngine(self.family["dit"], self.dit_weights, self.model.scheduler, self.device)
        self._editor.sched = self.model.scheduler
        self._editor.arith = self.arith
        return self._editor

    def load_scheduler(self) -> None:
        self.model.scheduler = CosineDPMSolverMultistepScheduler.from_config(self.family["scheduler"])

    def get_sr(self) -> int:
        return self.model.vae.config.sampling_rate

    def get_noise_shape(self, x0: torch.Tensor, num_steps: int) -> Tuple[int, int, int]:
        c = self.model.transformer.config
        return (num_steps, c.in_channels, int(c.sample_size))

    # ------------------------------------------------------------------ conditioning (models.py:1069-1103, :1142-1165)
    def encode_text(self, prompts: List[str], negative: bool = False):
        enc = getattr(self, "text_encoders", None)
        if enc is not None:
            return enc.encode_stable_audio(prompts, self.device, negative=negative)
        _require_synthetic_text(self)
        c = self.family["ctx"]
        S, dim = c["max_length"], c["t5_dim"]
        e = torch.zeros(len(prompts), S, dim)
        mask = torch.zeros(len(prompts), S, dtype=torch.long)
        for i, p in enumerate(prompts):
            n = min(len(p.split()) + 1, S)
            e[i] = torch.randn(S, dim, generator=_prompt_generator(p, "t5-sa"))      # non-zero at padded positions too
            mask[i, :n] = 1
        if negative:
            e = torch.where(mask.to(torch.bool).unsqueeze(2), e, 0.0)
        e = self._projection(text_hidden_states=e).text_hidden_states
        if prompts == [""]:
            return torch.zeros_like(e).to(self.device), None, None
        m = mask.unsqueeze(-1).to(e.dtype)
        return (e * m * m).to(self.device), None, mask.to(self.device)

    def _encode_duration(self, start, end):
        enc = getattr(self, "text_encoders", None)
        if enc is not None:
            return enc.encode_duration(start, end, self.device)
        with torch.no_grad():
            out = self._projection(start_seconds=[float(start)], end_seconds=[float(end)])
        return out.seconds_start_hidden_states.to(self.device), out.seconds_end_hidden_states.to(self.device)

    def assemble_context(self, encoder_hidden_states, encoder_attention_mask):
        """The DiT's cross-attention input of unet_forward (models.py:1340-1343)."""
        ctx = torch.cat([encoder_hidden_states.to(self.device), self.seconds_start_hidden_states,
                         self.seconds_end_hidden_states], dim=1)
        return torch.zeros_like(ctx) if encoder_attention_mask is None else ctx

    def setup_extra_inputs(self, x: torch.Tensor, init_timestep: torch.Tensor, extra_info=None,
                           audio_start_in_s: float = 0, audio_end_in_s: Optional[float] = None) -> None:
        c, v = self.model.transformer.config, self.model.vae
        max_len = c.sample_size * v.hop_length / v.config.sampling_rate
        if audio_end_in_s is None:
            audio_end_in_s = max_len
        if audio_end_in_s - audio_start_in_s > max_len:
            raise ValueError(f"The total audio length requested ({audio_end_in_s - audio_start_in_s}s) is longer than "
                             f"the model maximum possible length ({max_len}). Make sure that "
                             f"'audio_end_in_s-audio_start_in_s<={max_len}'.")
        self.waveform_start = int(audio_start_in_s * v.config.sampling_rate)
        self.waveform_end = int(audio_end_in_s * v.config.sampling_rate)
        self.seconds_start_hidden_states, self.seconds_end_hidden_states = self._encode_duration(audio_start_in_s,
                                                                                                  audio_end_in_s)
        self.audio_duration_embeds = torch.cat([self.seconds_start_hidden_states, self.seconds_end_hidden_states], dim=2)
        s = self.model.scheduler
        s._init_step_index(init_timestep)
        t_to_idx = {float(t): k for k, t in enumerate(s.timesteps)}
        idx = len(s.timesteps) - t_to_idx[float(init_timestep)] - 1
        s.model_outputs = [None, extra_info[idx] if extra_info is not None else None]
        s.lower_order_nums = min(s.step_index, s.config.solver_order)

    # ------------------------------------------------------------------ VAE (models.py:1117-1140)
    def get_fn_STFT(self):
        return _FnSTFT(self)

    def vae_encode(self, x: torch.Tensor, generator=None) -> torch.Tensor:
        from .stable_audio import OobleckEncoder
        x = x.unsqueeze(0)
        c, v = self.model.transformer.config, self.model.vae
        n = int(c.sample_size * v.hop_length)
        if x.shape[1] == 1 and v.config.audio_channels == 2:
            x = x.repeat(1, 2, 1)
        audio = x.new_zeros((1, v.config.audio_channels, n))
        audio[:, :, : min(x.shape[-1], n)] = x[:, :, :n]
        enc = self._cached(("oob_enc", n), lambda: OobleckEncoder(self.family["oobleck"], self.state_dicts["vae"],
                                                                   self.device, 1, n))
        lat = self.family["oobleck"]["decoder_input_channels"]
        noise = torch.randn((1, lat, c.sample_size), generator=generator, dtype=torch.float32)   # posterior .sample()
        z = enc(audio.transpose(1, 2), noise.transpose(1, 2))                                     # [1, Lz, lat]
        return z.transpose(1, 2).contiguous()

    def vae_decode(self, x: torch.Tensor) -> torch.Tensor:
        from .stable_audio import OobleckDecoder
        dec = self._cached(("oob_dec", x.shape[-1]), lambda: OobleckDecoder(self.family["oobleck"], self.state_dicts["vae"],
                                                                            self.device, 1, x.shape[-1]))
        aud = dec(x.to(self.device, torch.float32).transpose(1, 2)).transpose(1, 2).contiguous()   # [1, Ca, L]
        return aud[:, :, self.waveform_start:self.waveform_end]

    # ------------------------------------------------------------------ loops' primitives
    def sample_xts_from_x0(self, x0: torch.Tensor, num_inference_steps: int = 50) -> torch.Tensor:
        """x_t = x0 + n * sigma_t, independent noise per step drawn on the CPU generator in the reference's order
        (models.py:1186-1207); returns [T+1, C, L] on the device."""
        return self.editor().sample_xts(x0.reshape(1, *x0.shape[-2:]))[:, 0]

    def _coef(self, first_order):
        s = self.model.scheduler
        i = s.step_index
        order = sa_step_orders(s, i, 1, s.lower_order_nums, first_order)[0]
        return i, order

    def _history(self, like, order):
        s = self.model.scheduler
        if self._hist is None or self._hist.shape != like.shape:
            self._hist = torch.zeros_like(like)
        if order == 2:
            self._hist.copy_(s.model_outputs[-1])
        return self._hist

    def get_zs_from_xts(self, xt, xtm1, data_pred, t, numerical_fix: bool = True, first_order: bool = False, **kwargs):
        """models.py:1209-1271; `data_pred` is the (guided) model output.  The scheduler bookkeeping the reference does
        by hand (history shift, lower_order_nums, step index) is kept on `self.model.scheduler` so host-driven callers
        see the same state."""
        s = self.model.scheduler
        if s.step_index is None:
            s._init_step_index(t)
        i, order = self._coef(first_order)
        zero_z = (i == len(s.timesteps) - 1) and s.config.final_sigmas_type == "zero"
        c = sa_step_coefficients(s, i, order, zero_z=zero_z)
        cf = (ctypes.c_float * L.SA_COEF_STRIDE)(*c.tolist())
        xt, v = xt.contiguous(), data_pred.contiguous()
        xtm1 = xtm1.contiguous().clone()
        z = torch.empty_like(xt)
        hist = self._history(xt, order)
        prev = s.model_outputs[-1]
        L.check(L.lib().aed_sa_get_zs_from_xts(xt.data_ptr(), xtm1.data_ptr(), v.data_ptr(), None, 0.0, cf,
                                               hist.data_ptr(), int(bool(numerical_fix)), z.data_ptr(), None, xt.numel(),
                                               L.current_stream_ptr()), "aed_sa_get_zs_from_xts")
        s.model_outputs = [prev, hist.clone()]
        if s.lower_order_nums < s.config.solver_order:
            s.lower_order_nums += 1
        s._step_index += 1
        return z, xtm1, s.model_outputs[-2]

    def reverse_step_with_custom_noise(self, model_output, timestep, sample, variance_noise=None,
                                       first_order: bool = False, **kwargs):
        """models.py:1282-1329.  The editing loops always pass the inverted noise maps; with variance_noise=None the step
        draws from a Brownian path over sigma (models.py:1305-1312; scheduler.BrownianTreeNoiseSampler)."""
        s = self.model.scheduler
        if s.step_index is None:
            s._init_step_index(timestep)
        if variance_noise is None:
            if s.noise_sampler is None:
                from .scheduler import BrownianTreeNoiseSampler
                s.noise_sampler = BrownianTreeNoiseSampler(model_output, sigma_min=s.config.sigma_min,
                                                           sigma_max=s.config.sigma_max, seed=None)
            variance_noise = s.noise_sampler(s.sigmas[s.step_index], s.sigmas[s.step_index + 1]).to(model_output.device)
        i, order = self._coef(first_order)
        c = sa_step_coefficients(s, i, order)
        cf = (ctypes.c_float * L.SA_COEF_STRIDE)(*c.tolist())
        v, x, zn = model_output.contiguous(), sample.contiguous(), variance_noise.contiguous()
        hist = self._history(x, order)
        prev_m = s.model_outputs[-1]
        out = torch.empty_like(x)
        L.check(L.lib().aed_sa_reverse_step_with_custom_noise(x.data_ptr(), v.data_ptr(), None, 0.0, cf, hist.data_ptr(),
                                                              zn.data_ptr(), out.data_ptr(), x.numel(),
                                                              L.current_stream_ptr()),
                "aed_sa_reverse_step_with_custom_noise")
        s.model_outputs = [prev_m, hist.clone()]
        if s.lower_order_nums < s.config.solver_order:
            s.lower_order_nums += 1
        s._step_index += 1
        return out

    def unet_forward(self, sample, timestep, encoder_hidden_states, encoder_attention_mask=None,
                     return_dict: bool = True, **kwargs):
        """models.py:1331-1354: one DiT call; `sample` is already input-scaled by the caller (scale_model_input)."""
        from .stable_audio import DiTEngine
        B = sample.shape[0]
        ctx = self.assemble_context(encoder_hidden_states, encoder_attention_mask)
        eng = self._cached(("dit", B, ctx.shape[1]), lambda: DiTEngine(self.family["dit"], self.dit_weights, self.device,
                                                                        B, ctx.shape[1]), arith="f32")
        eng.set_conditioning(ctx.expand(B, -1, -1), self.audio_duration_embeds.reshape(1, -1).expand(B, -1))
        eng.set_timestep(timestep)
        eng.x_in.copy_(sample.to(self.device, torch.float32).transpose(1, 2))
        out = eng.forward().transpose(1, 2).contiguous()
        if not return_dict:
            return (out,)
        return UNet2DConditionOutput(sample=out), None, None


def load_model(model_id: str, device: torch.device, num_diffusion_steps: int, double_precision: bool = False,
               token: Optional[str] = None, seed: int = 0, state_dicts: Optional[Dict] = None,
               allow_synthetic: Optional[bool] = None) -> PipelineWrapper:
    """Substring dispatch + scheduler setup of models.py:1357-1374.  `allow_synthetic`: opt-in to seeded-random weights
    and stand-in text embeddings when no checkpoint is on disk (benchmarks / tests; see PipelineWrapper.__init__)."""
    fam = configs.family_of(model_id)
    cls = {"tango": TangoWrapper, "audioldm2": AudioLDM2Wrapper, "audioldm": AudioLDMWrapper,
           "stable_audio": StableAudWrapper}[fam]
    ldm_stable = cls(model_id=model_id, device=device, double_precision=double_precision, token=token, seed=seed,
                     state_dicts=state_dicts, allow_synthetic=allow_synthetic)
    ldm_stable.load_scheduler()
    ldm_stable.model.scheduler.set_timesteps(num_diffusion_steps, device=None)
    torch.cuda.empty_cache()
    return ldm_stable

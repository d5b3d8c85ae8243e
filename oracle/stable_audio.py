"""Oracle restatement of the Stable Audio Open path (SURVEY 8(f) row 4 / BASELINE config 5).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Two kinds of content:

(1) The reference's OWN wrapper math, all paths relative to /root/reference/code:
      models.py:1142-1184  setup_extra_inputs  (scheduler bookkeeping part)  -> OracleStableAudio.setup_extra_inputs
      models.py:1186-1207  sample_xts_from_x0   (x_t = x0 + n * sigma_t)      -> OracleStableAudio.sample_xts_from_x0
      models.py:1209-1271  get_zs_from_xts      (SDE-DPM-Solver++ solved for the noise, 1st / 2nd order, numerical fix)
      models.py:1282-1329  reverse_step_with_custom_noise
      models.py:1331-1354  unet_forward         (text | seconds_start | seconds_end context, zeroed when unconditional)
      ddm_inversion/inversion_utils.py:52-144, :200-316  the loops on a 3-D latent (scale_model_input before the model)
    PINNED: tests/golden/sa_*.npz are produced by importing the reference's StableAudWrapper methods and loops
    (oracle/make_golden.py gen_stable_audio) with the scheduler below standing in for diffusers'.

(2) Third-party pieces the reference only calls: `diffusers` (un-vendored and un-pinned, requirements.txt:1; README.md:39
    asks for >= 0.30) -- CosineDPMSolverMultistepScheduler, StableAudioDiTModel, StableAudioProjectionModel,
    AutoencoderOobleck, get_1d_rotary_pos_embed.  Absent from this image, restated from their published definitions:
    PARITY UNPINNED for those (the scheduler's tables and update formulas, the DiT graph, the Oobleck graph); the
    reference's own formulas in get_zs_from_xts (models.py:1238-1255) restate the same solver update and agree with
    the scheduler below to rounding, which the golden test checks (numerical fix ~ identity).
"""
import math
from types import SimpleNamespace

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- scheduler (diffusers, restated)
class OracleCosineDPMSolverScheduler:
    """CosineDPMSolverMultistepScheduler as Stable Audio Open configures it: exponential sigma schedule in
    [sigma_min, sigma_max], EDM-style preconditioning with sigma_data, v-prediction, sde-dpmsolver++ midpoint, order 2,
    final sigma zero.  alpha_t = 1, sigma_t = sigma, lambda = -log(sigma)."""

    def __init__(self, sigma_min=0.3, sigma_max=500.0, sigma_data=1.0, sigma_schedule="exponential",
                 num_train_timesteps=1000, solver_order=2, prediction_type="v_prediction", rho=7.0,
                 solver_type="midpoint", lower_order_final=True, euler_at_final=False, final_sigmas_type="zero"):
        self.config = SimpleNamespace(sigma_min=sigma_min, sigma_max=sigma_max, sigma_data=sigma_data,
                                      sigma_schedule=sigma_schedule, num_train_timesteps=num_train_timesteps,
                                      solver_order=solver_order, prediction_type=prediction_type, rho=rho,
                                      solver_type=solver_type, lower_order_final=lower_order_final,
                                      euler_at_final=euler_at_final, final_sigmas_type=final_sigmas_type)
        self.num_inference_steps = None
        self.model_outputs = [None] * solver_order
        self.lower_order_nums = 0
        self._step_index = None
        self._begin_index = None
        self.noise_sampler = None
        self.set_timesteps(num_train_timesteps)

    @property
    def step_index(self):
        return self._step_index

    @property
    def init_noise_sigma(self):
        return (self.config.sigma_max ** 2 + 1) ** 0.5

    def precondition_noise(self, sigma):
        if not isinstance(sigma, torch.Tensor):
            sigma = torch.tensor([sigma])
        return sigma.atan() / math.pi * 2

    def precondition_inputs(self, sample, sigma):
        return sample * (1 / ((sigma ** 2 + self.config.sigma_data ** 2) ** 0.5))

    def precondition_outputs(self, sample, model_output, sigma):
        sd = self.config.sigma_data
        c_skip = sd ** 2 / (sigma ** 2 + sd ** 2)
        if self.config.prediction_type == "epsilon":
            c_out = sigma * sd / (sigma ** 2 + sd ** 2) ** 0.5
        elif self.config.prediction_type == "v_prediction":
            c_out = -sigma * sd / (sigma ** 2 + sd ** 2) ** 0.5
        else:
            raise ValueError(self.config.prediction_type)
        return c_skip * sample + c_out * model_output

    def set_timesteps(self, num_inference_steps, device=None):
        c = self.config
        self.num_inference_steps = num_inference_steps
        if c.sigma_schedule == "exponential":
            sigmas = torch.linspace(math.log(c.sigma_min), math.log(c.sigma_max), num_inference_steps).exp().flip(0)
        elif c.sigma_schedule == "karras":
            ramp = torch.linspace(0, 1, num_inference_steps)
            lo, hi = c.sigma_min ** (1 / c.rho), c.sigma_max ** (1 / c.rho)
            sigmas = (hi + ramp * (lo - hi)) ** c.rho
        else:
            raise ValueError(c.sigma_schedule)
        sigmas = sigmas.to(torch.float32)
        self.timesteps = self.precondition_noise(sigmas)
        last = c.sigma_min if c.final_sigmas_type == "sigma_min" else 0.0
        self.sigmas = torch.cat([sigmas, torch.tensor([last], dtype=torch.float32)])
        self.model_outputs = [None] * c.solver_order
        self.lower_order_nums = 0
        self._step_index = None
        self._begin_index = None
        self.noise_sampler = None

    def index_for_timestep(self, timestep, schedule_timesteps=None):
        ts = self.timesteps if schedule_timesteps is None else schedule_timesteps
        idx = (ts == timestep).nonzero()
        return idx[1 if len(idx) > 1 else 0].item()

    def _init_step_index(self, timestep):
        if self._begin_index is None:
            self._step_index = self.index_for_timestep(timestep)
        else:
            self._step_index = self._begin_index

    def scale_model_input(self, sample, timestep):
        if self.step_index is None:
            self._init_step_index(timestep)
        return self.precondition_inputs(sample, self.sigmas[self.step_index])

    def convert_model_output(self, model_output, sample=None):
        return self.precondition_outputs(sample, model_output, self.sigmas[self.step_index])

    @staticmethod
    def _lambda(sigma):
        return torch.log(torch.tensor(1.0)) - torch.log(sigma)          # alpha_t = 1

    def dpm_solver_first_order_update(self, model_output, sample=None, noise=None):
        sigma_t, sigma_s = self.sigmas[self.step_index + 1], self.sigmas[self.step_index]
        h = self._lambda(sigma_t) - self._lambda(sigma_s)
        return ((sigma_t / sigma_s * torch.exp(-h)) * sample + (1 - torch.exp(-2.0 * h)) * model_output
                + sigma_t * torch.sqrt(1.0 - torch.exp(-2.0 * h)) * noise)

    def multistep_dpm_solver_second_order_update(self, model_output_list, sample=None, noise=None):
        sigma_t, sigma_s0, sigma_s1 = (self.sigmas[self.step_index + 1], self.sigmas[self.step_index],
                                       self.sigmas[self.step_index - 1])
        lt, l0, l1 = self._lambda(sigma_t), self._lambda(sigma_s0), self._lambda(sigma_s1)
        m0, m1 = model_output_list[-1], model_output_list[-2]
        h, h_0 = lt - l0, l0 - l1
        r0 = h_0 / h
        D0, D1 = m0, (1.0 / r0) * (m0 - m1)
        return ((sigma_t / sigma_s0 * torch.exp(-h)) * sample + (1 - torch.exp(-2.0 * h)) * D0
                + 0.5 * (1 - torch.exp(-2.0 * h)) * D1 + sigma_t * torch.sqrt(1.0 - torch.exp(-2.0 * h)) * noise)


# ----------------------------------------------------------------------------- wrapper math (the reference's own)
class OracleStableAudio:
    """Scheduler-facing methods of StableAudWrapper.  `dit` is any callable
    (x[B,C,L] already input-scaled, t float tensor, cond) -> v[B,C,L]."""

    def __init__(self, scheduler, dit, in_channels=64, sample_size=1024):
        self.model = SimpleNamespace(scheduler=scheduler)
        self.dit = dit
        self.in_channels = in_channels
        self.sample_size = sample_size

    def get_noise_shape(self, x0, n):                       # models.py:1276-1280
        return (n, self.in_channels, self.sample_size)

    def sample_xts_from_x0(self, x0, num_inference_steps, generator=None):      # models.py:1186-1207
        s = self.model.scheduler
        xts = torch.zeros(self.get_noise_shape(x0, num_inference_steps + 1))
        xts[0] = x0
        t_to_idx = {float(v): k for k, v in enumerate(s.timesteps)}
        for t in reversed(s.timesteps):
            idx = num_inference_steps - t_to_idx[float(t)]
            n = torch.randn(x0.shape, generator=generator, dtype=x0.dtype)
            xts[idx] = x0 + n * s.sigmas[t_to_idx[float(t)]]
        return xts

    def setup_extra_inputs(self, init_timestep, extra_info=None):               # models.py:1174-1184
        s = self.model.scheduler
        s._init_step_index(init_timestep)
        t_to_idx = {float(v): k for k, v in enumerate(s.timesteps)}
        idx = len(s.timesteps) - t_to_idx[float(init_timestep)] - 1
        s.model_outputs = [None, extra_info[idx] if extra_info is not None else None]
        s.lower_order_nums = min(s.step_index, s.config.solver_order)

    def _order_flags(self):
        s = self.model.scheduler
        n = len(s.timesteps)
        final = (s.step_index == n - 1) and (s.config.euler_at_final or (s.config.lower_order_final and n < 15)
                                             or s.config.final_sigmas_type == "zero")
        second = (s.step_index == n - 2) and s.config.lower_order_final and n < 15
        return final, second

    def get_zs_from_xts(self, xt, xtm1, v, t, numerical_fix=True, first_order=False):     # models.py:1209-1271
        s = self.model.scheduler
        sig, order = s.sigmas, s.config.solver_order
        if s.step_index is None:
            s._init_step_index(t)
        i = s.step_index
        final, second = self._order_flags()
        d = s.convert_model_output(v, sample=xt)
        for k in range(order - 1):
            s.model_outputs[k] = s.model_outputs[k + 1]
        s.model_outputs[-1] = d
        use_first = first_order or order == 1 or s.lower_order_nums < 1 or final
        if i == len(s.timesteps) - 1 and s.config.final_sigmas_type == "zero":
            z = torch.zeros_like(xt)
        elif use_first:
            st, ss = sig[i + 1], sig[i]
            h = torch.log(ss) - torch.log(st)
            z = (xtm1 - (st / ss * torch.exp(-h)) * xt - (1 - torch.exp(-2.0 * h)) * d) \
                / (st * torch.sqrt(1.0 - torch.exp(-2 * h)))
        else:
            st, s0, s1 = sig[i + 1], sig[i], sig[i - 1]
            m0, m1 = s.model_outputs[-1], s.model_outputs[-2]
            h, h_0 = torch.log(s0) - torch.log(st), torch.log(s1) - torch.log(s0)
            r0 = h_0 / h
            D1 = (1.0 / r0) * (m0 - m1)
            z = (xtm1 - (st / s0 * torch.exp(-h)) * xt - (1 - torch.exp(-2.0 * h)) * m0
                 - 0.5 * (1 - torch.exp(-2.0 * h)) * D1) / (st * torch.sqrt(1.0 - torch.exp(-2 * h)))
        if numerical_fix:
            if use_first:
                xtm1 = s.dpm_solver_first_order_update(d, sample=xt, noise=z)
            else:
                xtm1 = s.multistep_dpm_solver_second_order_update(s.model_outputs, sample=xt, noise=z)
        if s.lower_order_nums < order:
            s.lower_order_nums += 1
        s._step_index += 1
        return z, xtm1, s.model_outputs[-2]

    def reverse_step_with_custom_noise(self, v, t, sample, variance_noise, first_order=False):   # models.py:1282-1329
        s = self.model.scheduler
        if s.step_index is None:
            s._init_step_index(t)
        final, second = self._order_flags()
        d = s.convert_model_output(v, sample=sample)
        for k in range(s.config.solver_order - 1):
            s.model_outputs[k] = s.model_outputs[k + 1]
        s.model_outputs[-1] = d
        if first_order or s.config.solver_order == 1 or s.lower_order_nums < 1 or final:
            prev = s.dpm_solver_first_order_update(d, sample=sample, noise=variance_noise)
        else:
            prev = s.multistep_dpm_solver_second_order_update(s.model_outputs, sample=sample, noise=variance_noise)
        if s.lower_order_nums < s.config.solver_order:
            s.lower_order_nums += 1
        s._step_index += 1
        return prev


def invert(w, x0, cond_src, cond_uncond, cfg_scale, num_inference_steps, numerical_fix=True, src_is_empty=False,
           first_order=False, generator=None, xts=None):
    """inversion_forward_process on a 3-D latent, one prompt (inversion_utils.py:52-144).
    Returns (xt, zs, xts, extra_info)."""
    s = w.model.scheduler
    T = num_inference_steps
    ts = s.timesteps
    if xts is None:
        xts = w.sample_xts_from_x0(x0, T, generator=generator)
    zs = torch.zeros(w.get_noise_shape(x0, T))
    extra_info = [None] * T
    t_to_idx = {float(v): k for k, v in enumerate(ts)}
    xt = x0
    w.setup_extra_inputs(ts[0])
    for t in ts:
        idx = T - t_to_idx[float(t)] - 1
        xt = xts[idx + 1][None]
        xt_inp = s.scale_model_input(xt, t)
        v = w.dit(xt_inp, t, cond_uncond)
        if not src_is_empty:
            v_c = w.dit(xt_inp, t, cond_src)
            v = v + (cfg_scale * (v_c - v)).sum(axis=0).unsqueeze(0)
        z, xtm1, extra = w.get_zs_from_xts(xt, xts[idx][None], v, t, numerical_fix=numerical_fix, first_order=first_order)
        zs[idx] = z
        xts[idx] = xtm1
        extra_info[idx] = extra
    zs[0] = torch.zeros_like(zs[0])
    return xt, zs, xts, extra_info


def edit(w, xT, tstart, cond_tgt, cond_neg, cfg_scale, zs, extra_info=None, first_order=False):
    """inversion_reverse_process on a 3-D latent, one prompt (inversion_utils.py:200-316)."""
    s = w.model.scheduler
    T = len(s.timesteps)
    xt = xT[int(tstart)].unsqueeze(0)
    Z = zs.shape[0]
    ts = s.timesteps[-Z:]
    t_to_idx = {float(v): k for k, v in enumerate(ts)}
    w.setup_extra_inputs(s.timesteps[-Z], extra_info=extra_info)
    for t in ts:
        idx = T - t_to_idx[float(t)] - (T - Z + 1)
        xt_inp = s.scale_model_input(xt, t)
        v_u = w.dit(xt_inp, t, cond_neg)
        v_c = w.dit(xt_inp, t, cond_tgt)
        v = v_u + (cfg_scale * (v_c - v_u)).sum(axis=0).unsqueeze(0)
        xt = w.reverse_step_with_custom_noise(v, t, xt, variance_noise=zs[idx].unsqueeze(0), first_order=first_order)
    return xt


def synthetic_dit(x, t, cond_vec):
    """Smooth seed-free stand-in for the DiT on a 3-D latent (loop fixtures): x[B,C,L], t float, cond_vec[B,D]."""
    c = x.shape[1]
    k = torch.zeros(c, c, 3)
    oi = torch.arange(c, dtype=torch.float32)
    for y in range(3):
        k[:, :, y] = torch.sin(1.0 + 0.7 * oi[:, None] + 1.3 * oi[None, :] + 2.1 * y) / (2.0 * c)
    tt = float(t)
    h = F.conv1d(x, k.to(x.dtype), padding=1)
    shift = cond_vec.to(x.dtype).mean(dim=1).view(-1, 1, 1)
    return torch.tanh(h * (0.5 + tt) + 0.25 * shift) + 0.1 * x * (1.0 - tt)


# ----------------------------------------------------------------------------- conditioning rules (the reference's own)
def encode_text_rule(tokenizer, text_encoder, projection, prompts, negative=False):
    """StableAudWrapper.encode_text (models.py:1069-1103): max-length padding, T5 states, masked tokens zeroed BEFORE the
    projection only for negative prompts, then the projection, then the mask applied twice (idempotent for a 0/1 mask);
    the empty prompt returns zeros and NO mask (which unet_forward reads as "zero the whole context")."""
    ti = tokenizer(prompts, padding="max_length", max_length=tokenizer.model_max_length, truncation=True,
                   return_tensors="pt")
    ids, mask = ti.input_ids, ti.attention_mask
    e = text_encoder(ids, attention_mask=mask)[0]
    if negative:
        e = torch.where(mask.to(torch.bool).unsqueeze(2), e, 0.0)
    e = projection(text_hidden_states=e).text_hidden_states
    if prompts == [""]:
        return torch.zeros_like(e), None, None
    m = mask.unsqueeze(-1).to(e.dtype)
    return e * m * m, None, mask


def assemble_context(text_states, mask, seconds_start, seconds_end):
    """StableAudWrapper.unet_forward's context (models.py:1340-1343)."""
    ctx = torch.cat([text_states, seconds_start, seconds_end], dim=1)
    return torch.zeros_like(ctx) if mask is None else ctx


class StandIns:
    """Small deterministic stand-ins for the pipeline members the reference's wrapper dereferences (tokenizer, T5,
    projection model, encode_duration, transformer); shared by make_golden.py (driving the reference) and the tests
    (driving the oracle / product), so what the fixtures pin is the reference's own code around them."""
    C, Lz, D, S = 6, 24, 8, 5

    class Tok:
        model_max_length = 5

        def __call__(self, prompts, padding=None, max_length=None, truncation=None, return_tensors=None):
            ids = torch.zeros((len(prompts), max_length), dtype=torch.long)
            mask = torch.zeros((len(prompts), max_length), dtype=torch.long)
            for i, p in enumerate(prompts):
                w = [1 + (sum(map(ord, t)) % 50) for t in p.split()][: max_length - 1] + [99]   # words + EOS
                ids[i, : len(w)] = torch.tensor(w)
                mask[i, : len(w)] = 1
            return SimpleNamespace(input_ids=ids, attention_mask=mask)

    class TextEnc:
        def eval(self):
            return self

        def __call__(self, ids, attention_mask=None):
            pos = torch.arange(ids.shape[1], dtype=torch.float32)[None, :, None]
            f = torch.arange(1, StandIns.D + 1, dtype=torch.float32)[None, None, :]
            return (torch.sin(ids[..., None].float() * 0.37 * f + 0.11 * pos) + 0.05,)      # non-zero at padded tokens

    @staticmethod
    def projection_model(text_hidden_states=None):
        return SimpleNamespace(text_hidden_states=text_hidden_states * 1.5)

    @staticmethod
    def encode_duration(start, end, device=None, cfg=False, n=1):
        return (torch.full((1, 1, StandIns.D), float(start) * 0.01 + 0.2),
                torch.full((1, 1, StandIns.D), float(end) * 0.01 - 0.1))

    @staticmethod
    def transformer(sample, timestep, encoder_hidden_states=None, global_hidden_states=None, rotary_embedding=None):
        cond = encoder_hidden_states.mean(dim=1) + 0.1 * global_hidden_states.mean(dim=(1, 2))[:, None] \
            + 0.01 * rotary_embedding[0].sum()
        return SimpleNamespace(sample=synthetic_dit(sample, timestep[0], cond))


# ----------------------------------------------------------------------------- DiT (diffusers StableAudioDiTModel, restated)
def rotary_table(dim, n, theta=10000.0):
    """get_1d_rotary_pos_embed(dim, n, use_real=True, repeat_interleave_real=False): (cos[n,dim], sin[n,dim]), halves
    duplicated (cat, not interleave)."""
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))
    ang = torch.outer(torch.arange(n).float(), freqs)
    return torch.cat([ang.cos(), ang.cos()], dim=-1), torch.cat([ang.sin(), ang.sin()], dim=-1)


def apply_rotary(x, cos, sin):
    """x[B,H,N,D]; rotates the first cos.shape[-1] features of every head (use_real_unbind_dim=-2: the feature vector is
    split in two halves (re, im); rotated = cat(-im, re))."""
    r = cos.shape[-1]
    xr, xp = x[..., :r], x[..., r:]
    re, im = xr.reshape(*xr.shape[:-1], 2, -1).unbind(-2)
    rot = torch.cat([-im, re], dim=-1)
    out = xr.float() * cos[None, None] + rot.float() * sin[None, None]
    return torch.cat([out.to(x.dtype), xp], dim=-1)


def _attn(sd, pre, x, ctx, heads, kv_heads, head_dim, rotary=None):
    B, N, _ = x.shape
    q = F.linear(x, sd[pre + "to_q.weight"])
    k = F.linear(ctx, sd[pre + "to_k.weight"])
    v = F.linear(ctx, sd[pre + "to_v.weight"])
    q = q.view(B, N, heads, head_dim).transpose(1, 2)
    k = k.view(B, -1, kv_heads, head_dim).transpose(1, 2)
    v = v.view(B, -1, kv_heads, head_dim).transpose(1, 2)
    if kv_heads != heads:
        k = k.repeat_interleave(heads // kv_heads, dim=1)
        v = v.repeat_interleave(heads // kv_heads, dim=1)
    if rotary is not None:
        q = apply_rotary(q, *rotary)
        k = apply_rotary(k, *rotary)
    o = F.scaled_dot_product_attention(q, k, v)
    o = o.transpose(1, 2).reshape(B, N, heads * head_dim)
    return F.linear(o, sd[pre + "to_out.0.weight"])


def dit_forward(sd, cfg, sample, timestep, encoder_hidden_states, global_hidden_states, rotary):
    """StableAudioDiTModel.forward.  sample[B,C,L], timestep[1|B] float, encoder_hidden_states[B,S,Dc],
    global_hidden_states[B,1,Dg], rotary = (cos, sin) over L+1 positions.  Returns v[B,C,L]."""
    H, KV, D = cfg["num_attention_heads"], cfg["num_key_value_attention_heads"], cfg["attention_head_dim"]
    ca = F.linear(F.silu(F.linear(encoder_hidden_states, sd["cross_attention_proj.0.weight"])),
                  sd["cross_attention_proj.2.weight"])
    g = F.linear(F.silu(F.linear(global_hidden_states, sd["global_proj.0.weight"])), sd["global_proj.2.weight"])
    tp = 2 * math.pi * timestep.float()[:, None] * sd["time_proj.weight"][None, :]
    tf = torch.cat([tp.cos(), tp.sin()], dim=-1)                      # flip_sin_to_cos=True, log=False
    te = F.linear(F.silu(F.linear(tf, sd["timestep_proj.0.weight"], sd["timestep_proj.0.bias"])),
                  sd["timestep_proj.2.weight"], sd["timestep_proj.2.bias"])
    g = g + te.unsqueeze(1)
    h = F.conv1d(sample, sd["preprocess_conv.weight"]) + sample
    h = F.linear(h.transpose(1, 2), sd["proj_in.weight"])
    h = torch.cat([g.expand(h.shape[0], -1, -1), h], dim=-2)
    for i in range(cfg["num_layers"]):
        p = f"transformer_blocks.{i}."
        n = F.layer_norm(h, h.shape[-1:], sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5)
        h = h + _attn(sd, p + "attn1.", n, n, H, H, D, rotary)
        n = F.layer_norm(h, h.shape[-1:], sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5)
        h = h + _attn(sd, p + "attn2.", n, ca, H, KV, D)
        n = F.layer_norm(h, h.shape[-1:], sd[p + "norm3.weight"], sd[p + "norm3.bias"], 1e-5)
        u = F.linear(n, sd[p + "ff.net.0.proj.weight"], sd[p + "ff.net.0.proj.bias"])
        val, gate = u.chunk(2, dim=-1)
        h = h + F.linear(val * F.silu(gate), sd[p + "ff.net.2.weight"], sd[p + "ff.net.2.bias"])
    h = F.linear(h, sd["proj_out.weight"]).transpose(1, 2)[:, :, 1:]
    return F.conv1d(h, sd["postprocess_conv.weight"]) + h


# ----------------------------------------------------------------------------- projection model (duration conditioning)
def number_conditioner(sd, pre, seconds, lo, hi):
    """StableAudioNumberConditioner: clamp, normalise to [0,1], learned Fourier features (+ the raw value), Linear."""
    x = torch.as_tensor(seconds, dtype=torch.float32).reshape(-1).clamp(lo, hi)
    x = (x - lo) / (hi - lo)
    t = x[..., None]
    fr = t * sd[pre + "time_positional_embedding.0.weights"][None] * 2 * math.pi
    feat = torch.cat([t, fr.sin(), fr.cos()], dim=-1)
    e = F.linear(feat, sd[pre + "time_positional_embedding.1.weight"], sd[pre + "time_positional_embedding.1.bias"])
    return e.view(-1, 1, e.shape[-1])


# ----------------------------------------------------------------------------- Oobleck VAE (diffusers AutoencoderOobleck, restated)
def snake(x, alpha, beta):
    """Snake1d with logscale parameters: x + sin^2(e^alpha x) / (e^beta + 1e-9); alpha, beta [1,C,1]."""
    a, b = torch.exp(alpha), torch.exp(beta)
    return x + (b + 1e-9).reciprocal() * torch.sin(a * x).pow(2)


def _res_unit(sd, p, x, dilation):
    pad = ((7 - 1) * dilation) // 2
    y = F.conv1d(snake(x, sd[p + "snake1.alpha"], sd[p + "snake1.beta"]), sd[p + "conv1.weight"], sd[p + "conv1.bias"],
                 dilation=dilation, padding=pad)
    y = F.conv1d(snake(y, sd[p + "snake2.alpha"], sd[p + "snake2.beta"]), sd[p + "conv2.weight"], sd[p + "conv2.bias"])
    crop = (x.shape[-1] - y.shape[-1]) // 2
    if crop > 0:
        x = x[..., crop:-crop]
    return x + y


def oobleck_encode(sd, cfg, audio):
    """audio[B,Ca,L] -> (mean, std) of the diagonal posterior, each [B, latent, L/hop]."""
    h = F.conv1d(audio, sd["encoder.conv1.weight"], sd["encoder.conv1.bias"], padding=3)
    for i, stride in enumerate(cfg["downsampling_ratios"]):
        p = f"encoder.block.{i}."
        for j, dil in enumerate((1, 3, 9)):
            h = _res_unit(sd, p + f"res_unit{j + 1}.", h, dil)
        h = snake(h, sd[p + "snake1.alpha"], sd[p + "snake1.beta"])
        h = F.conv1d(h, sd[p + "conv1.weight"], sd[p + "conv1.bias"], stride=stride, padding=math.ceil(stride / 2))
    h = snake(h, sd["encoder.snake1.alpha"], sd["encoder.snake1.beta"])
    h = F.conv1d(h, sd["encoder.conv2.weight"], sd["encoder.conv2.bias"], padding=1)
    mean, scale = h.chunk(2, dim=1)
    return mean, F.softplus(scale) + 1e-4


def oobleck_decode(sd, cfg, z):
    """z[B,latent,Lz] -> audio[B,Ca,Lz*hop]."""
    h = F.conv1d(z, sd["decoder.conv1.weight"], sd["decoder.conv1.bias"], padding=3)
    for i, stride in enumerate(cfg["downsampling_ratios"][::-1]):
        p = f"decoder.block.{i}."
        h = snake(h, sd[p + "snake1.alpha"], sd[p + "snake1.beta"])
        h = F.conv_transpose1d(h, sd[p + "conv_t1.weight"], sd[p + "conv_t1.bias"], stride=stride,
                               padding=math.ceil(stride / 2))
        for j, dil in enumerate((1, 3, 9)):
            h = _res_unit(sd, p + f"res_unit{j + 1}.", h, dil)
    h = snake(h, sd["decoder.snake1.alpha"], sd["decoder.snake1.beta"])
    return F.conv1d(h, sd["decoder.conv2.weight"], None, padding=3)

"""DDIM scheduler state the reference reads through `model.model.scheduler`.

The reference takes this object from diffusers (`DDIMScheduler.from_pretrained`,
/root/reference/code/models.py:481,:567) and only ever touches: `.timesteps`, `.alphas_cumprod`,
`.final_alpha_cumprod`, `.num_inference_steps`, `.config.{num_train_timesteps,prediction_type}`,
`.scale_model_input`, `.init_noise_sigma`, `.set_timesteps`, `.step`, `._get_variance`,
`.add_noise`.  This is that duck type, host-side, fp32 tables.

`step_coefficients` evaluates the per-step scalars of get_zs_from_xts /
reverse_step_with_custom_noise (models.py:91-113, :124-150) with the reference's own fp32
0-dim-tensor expression order, so the device kernel (csrc/elementwise.hip, K1) is bit-exact.
"""
from types import SimpleNamespace

import numpy as np
import torch

from ._lib import COEF_STRIDE


class DDIMScheduler:
    def __init__(self, num_train_timesteps=1000, beta_start=0.0015, beta_end=0.0195,
                 beta_schedule="scaled_linear", set_alpha_to_one=False, steps_offset=1,
                 prediction_type="epsilon", timestep_spacing="leading", clip_sample=False, **_ignored):
        if beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        else:
            raise NotImplementedError(f"beta_schedule={beta_schedule}")
        if clip_sample:
            raise NotImplementedError("clip_sample=True is not used by the AudioLDM/TANGO schedulers")
        self.betas = betas
        self.alphas = 1.0 - betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, prediction_type=prediction_type,
                                      steps_offset=steps_offset, timestep_spacing=timestep_spacing,
                                      set_alpha_to_one=set_alpha_to_one, beta_start=beta_start, beta_end=beta_end,
                                      beta_schedule=beta_schedule, clip_sample=clip_sample)
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    @classmethod
    def from_config(cls, cfg):
        return cls(**{k: v for k, v in cfg.items() if not k.startswith("_")})

    def set_timesteps(self, num_inference_steps, device=None):
        n = self.config.num_train_timesteps
        if num_inference_steps > n:
            raise ValueError(f"num_inference_steps={num_inference_steps} > num_train_timesteps={n}")
        self.num_inference_steps = num_inference_steps
        sp = self.config.timestep_spacing
        if sp == "leading":
            ratio = n // num_inference_steps
            ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
            ts += self.config.steps_offset
        elif sp == "trailing":
            ts = np.round(np.arange(n, 0, -n / num_inference_steps)).astype(np.int64) - 1
        elif sp == "linspace":
            ts = np.linspace(0, n - 1, num_inference_steps).round()[::-1].copy().astype(np.int64)
        else:
            raise ValueError(f"timestep_spacing={sp}")
        self.timesteps = torch.from_numpy(ts)
        if device is not None:
            self.timesteps = self.timesteps.to(device)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def _alpha_prev(self, prev_t):
        return self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod

    def _get_variance(self, timestep, prev_timestep):
        a_t = self.alphas_cumprod[timestep]
        a_p = self._alpha_prev(prev_timestep)
        return ((1 - a_p) / (1 - a_t)) * (1 - a_t / a_p)

    def prev_timestep(self, t):
        return t - self.config.num_train_timesteps // self.num_inference_steps

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None,
             variance_noise=None, return_dict=True):
        """diffusers-semantics DDIM step on whatever device the tensors live on (elementwise torch;
        used by callers such as pc_drift.py, not by the native loops)."""
        t = int(timestep)
        pt = self.prev_timestep(t)
        a_t = self.alphas_cumprod[t]
        a_p = self._alpha_prev(pt)
        b_t = 1 - a_t
        if self.config.prediction_type == "epsilon":
            x0 = (sample - float(b_t ** 0.5) * model_output) / float(a_t ** 0.5)
            pe = model_output
        else:
            x0 = float(a_t ** 0.5) * sample - float(b_t ** 0.5) * model_output
            pe = float(a_t ** 0.5) * model_output + float(b_t ** 0.5) * sample
        var = self._get_variance(t, pt)
        std = eta * var ** 0.5
        prev = float(a_p ** 0.5) * x0 + float((1 - a_p - std ** 2) ** 0.5) * pe
        if eta > 0:
            if variance_noise is None:
                variance_noise = torch.randn(model_output.shape, generator=generator).to(model_output.device)
            prev = prev + float(std) * variance_noise
        return SimpleNamespace(prev_sample=prev, pred_original_sample=x0)

    def add_noise(self, original_samples, noise, timesteps):
        ts = torch.as_tensor(timesteps).cpu()
        a = (self.alphas_cumprod[ts] ** 0.5).to(original_samples.device)
        s = ((1 - self.alphas_cumprod[ts]) ** 0.5).to(original_samples.device)
        while a.dim() < original_samples.dim():
            a, s = a.unsqueeze(-1), s.unsqueeze(-1)
        return a * original_samples + s * noise


def step_coefficients(sched, t, eta=1.0):
    """[c0..c4] fp32 for one timestep, reference expression order (fp32 0-dim tensor arithmetic):
    c0=(1-abar_t)**.5  c1=abar_t**.5  c2=abar_prev**.5  c3=(1-abar_prev-eta*var)**.5  c4=eta*var**.5"""
    t = int(t)
    abar = sched.alphas_cumprod
    pt = sched.prev_timestep(t)
    a_t = abar[t]
    a_p = sched._alpha_prev(pt)
    var = ((1 - a_p) / (1 - a_t)) * (1 - a_t / a_p)
    c = torch.zeros(COEF_STRIDE, dtype=torch.float32)
    c[0] = (1 - a_t) ** 0.5
    c[1] = a_t ** 0.5
    c[2] = a_p ** 0.5
    c[3] = (1 - a_p - eta * var) ** 0.5
    c[4] = eta * var ** 0.5
    return c


def ddim_next_coefficients(sched, t):
    """next_step of the DDIM inversion baseline (ddim_inversion.py:10-20), same kernel form:
    x_next = c2 * ((x - c0*eps)/c1) + c3*eps."""
    t = int(t)
    t_cur = min(t - sched.config.num_train_timesteps // sched.num_inference_steps, 999)
    a_t = sched.alphas_cumprod[t_cur] if t_cur >= 0 else sched.final_alpha_cumprod
    a_n = sched.alphas_cumprod[t]
    c = torch.zeros(COEF_STRIDE, dtype=torch.float32)
    c[0] = (1 - a_t) ** 0.5
    c[1] = a_t ** 0.5
    c[2] = a_n ** 0.5
    c[3] = (1 - a_n) ** 0.5
    return c


def ddim_prev_coefficients(sched, t):
    """scheduler.step(eta=0) of the DDIM sampling baseline (ddim_inversion.py:82)."""
    t = int(t)
    pt = sched.prev_timestep(t)
    a_t = sched.alphas_cumprod[t]
    a_p = sched._alpha_prev(pt)
    var = ((1 - a_p) / (1 - a_t)) * (1 - a_t / a_p)
    std = 0.0 * var ** 0.5
    c = torch.zeros(COEF_STRIDE, dtype=torch.float32)
    c[0] = (1 - a_t) ** 0.5
    c[1] = a_t ** 0.5
    c[2] = a_p ** 0.5
    c[3] = (1 - a_p - std ** 2) ** 0.5
    return c


def coefficient_table(sched, timesteps, eta=1.0, kind="ddpm"):
    """One coefficient row per loop step.  `eta`: a scalar, or one value PER ROW (the caller has already put the
    reference's `etas[idx]` list into loop order, see editing.EditEngine._etas_in_loop_order)."""
    n = len(timesteps)
    is_seq = isinstance(eta, (list, tuple)) or (torch.is_tensor(eta) and eta.dim() > 0)
    etas = [float(e) for e in eta] if is_seq else [float(eta)] * n
    if len(etas) != n:
        raise ValueError(f"{len(etas)} eta values for {n} steps")
    if kind == "ddpm":
        return torch.stack([step_coefficients(sched, int(t), e) for t, e in zip(timesteps, etas)]).contiguous()
    fn = {"ddim_next": lambda t: ddim_next_coefficients(sched, t), "ddim_prev": lambda t: ddim_prev_coefficients(sched, t)}[kind]
    return torch.stack([fn(int(t)) for t in timesteps]).contiguous()


# =============================================================================================== Stable Audio Open
class BrownianTreeNoiseSampler:
    """The noise source of the reference's `reverse_step_with_custom_noise(variance_noise=None)` branch
    (models.py:1305-1312: diffusers' BrownianTreeNoiseSampler over torchsde's Brownian tree, both un-vendored, created
    with `seed=None`, i.e. fresh entropy per sampler -- there is no stream to reproduce, the contract is the
    distribution).  One Brownian path W per latent element over sigma; a call returns
        sign * (W(hi) - W(lo)) / sqrt(hi - lo),   lo/hi = the ordered pair, sign = -1 when sigma > sigma_next,
    i.e. N(0, 1) per element, identical for a repeated interval, independent over disjoint intervals and additive over
    adjacent ones.  The path is built lazily at the queried sigmas: a new point between two known ones is drawn from the
    Brownian bridge between them (Levy construction), one outside the known range from an independent increment."""

    def __init__(self, x, sigma_min, sigma_max, seed=None):
        self.shape, self.device = tuple(x.shape), x.device
        self.gen = torch.Generator()
        if seed is None:
            self.gen.seed()
        else:
            self.gen.manual_seed(int(seed))
        t0, t1 = sorted((float(sigma_min), float(sigma_max)))
        self.times = [t0, t1]
        self.values = [torch.zeros(self.shape), self._randn() * (t1 - t0) ** 0.5]

    def _randn(self):
        return torch.randn(self.shape, generator=self.gen, dtype=torch.float32)

    def _at(self, t):
        import bisect
        t = float(t)
        k = bisect.bisect_left(self.times, t)
        if k < len(self.times) and self.times[k] == t:
            return self.values[k]
        if k == 0:                                        # left of every known point
            w = self.values[0] - self._randn() * (self.times[0] - t) ** 0.5
        elif k == len(self.times):                        # right of every known point
            w = self.values[-1] + self._randn() * (t - self.times[-1]) ** 0.5
        else:                                             # Brownian bridge between the neighbours
            a, b = self.times[k - 1], self.times[k]
            wa, wb = self.values[k - 1], self.values[k]
            w = wa + (t - a) / (b - a) * (wb - wa) + self._randn() * ((t - a) * (b - t) / (b - a)) ** 0.5
        self.times.insert(k, t)
        self.values.insert(k, w)
        return w

    def __call__(self, sigma, sigma_next):
        a, b = float(sigma), float(sigma_next)
        if a == b:
            raise ValueError("BrownianTreeNoiseSampler: empty interval")
        lo, hi, sign = (a, b, 1.0) if a < b else (b, a, -1.0)
        return (sign * (self._at(hi) - self._at(lo)) / (hi - lo) ** 0.5).to(self.device)


class CosineDPMSolverMultistepScheduler:
    """The scheduler duck type StableAudWrapper reads (/root/reference/code/models.py:1066-1068, :1142-1329): `.sigmas`,
    `.timesteps` (float, atan(sigma)*2/pi), `.config.{solver_order,final_sigmas_type,lower_order_final,euler_at_final,
    sigma_min,sigma_max,sigma_data}`, `.step_index`/`._step_index`, `.lower_order_nums`, `.model_outputs`,
    `.scale_model_input`, `.convert_model_output`, `._init_step_index`, the two DPM-Solver++ update functions.
    diffusers is un-vendored: restated from the published scheduler (exponential sigma schedule, EDM preconditioning,
    sde-dpmsolver++ midpoint); host-side fp32 tables.  The native loops do not call the update methods -- they read
    `sa_coefficient_table` -- the methods serve host-driven callers and keep the object drop-in."""

    def __init__(self, sigma_min=0.3, sigma_max=500.0, sigma_data=1.0, sigma_schedule="exponential",
                 num_train_timesteps=1000, solver_order=2, prediction_type="v_prediction", rho=7.0,
                 solver_type="midpoint", lower_order_final=True, euler_at_final=False, final_sigmas_type="zero",
                 **_ignored):
        if solver_order not in (1, 2):
            raise NotImplementedError(f"solver_order={solver_order}")
        if prediction_type not in ("v_prediction", "epsilon"):
            raise NotImplementedError(f"prediction_type={prediction_type}")
        self.config = SimpleNamespace(sigma_min=sigma_min, sigma_max=sigma_max, sigma_data=sigma_data,
                                      sigma_schedule=sigma_schedule, num_train_timesteps=num_train_timesteps,
                                      solver_order=solver_order, prediction_type=prediction_type, rho=rho,
                                      solver_type=solver_type, lower_order_final=lower_order_final,
                                      euler_at_final=euler_at_final, final_sigmas_type=final_sigmas_type)
        self.num_inference_steps = None
        self.set_timesteps(num_train_timesteps)

    @classmethod
    def from_config(cls, cfg):
        return cls(**{k: v for k, v in cfg.items() if not k.startswith("_")})

    @property
    def step_index(self):
        return self._step_index

    @property
    def init_noise_sigma(self):
        return (self.config.sigma_max ** 2 + 1) ** 0.5

    def precondition_noise(self, sigma):
        if not isinstance(sigma, torch.Tensor):
            sigma = torch.tensor([sigma])
        return sigma.atan() / np.pi * 2

    def set_timesteps(self, num_inference_steps, device=None):
        c = self.config
        self.num_inference_steps = num_inference_steps
        if c.sigma_schedule == "exponential":
            sigmas = torch.linspace(np.log(c.sigma_min), np.log(c.sigma_max), num_inference_steps).exp().flip(0)
        elif c.sigma_schedule == "karras":
            ramp = torch.linspace(0, 1, num_inference_steps)
            lo, hi = c.sigma_min ** (1 / c.rho), c.sigma_max ** (1 / c.rho)
            sigmas = (hi + ramp * (lo - hi)) ** c.rho
        else:
            raise ValueError(f"sigma_schedule={c.sigma_schedule}")
        sigmas = sigmas.to(torch.float32)
        self.timesteps = self.precondition_noise(sigmas)
        if device is not None:
            self.timesteps = self.timesteps.to(device)
        last = c.sigma_min if c.final_sigmas_type == "sigma_min" else 0.0
        self.sigmas = torch.cat([sigmas, torch.tensor([last], dtype=torch.float32)])       # host table
        self.model_outputs = [None] * c.solver_order
        self.lower_order_nums = 0
        self._step_index = None
        self._begin_index = None
        self.noise_sampler = None

    def index_for_timestep(self, timestep, schedule_timesteps=None):
        ts = self.timesteps if schedule_timesteps is None else schedule_timesteps
        idx = (ts.cpu() == torch.as_tensor(timestep).cpu()).nonzero()
        if len(idx) == 0:                # diffusers' fallback: a timestep outside the current schedule maps to the last index
            return len(ts) - 1
        return idx[1 if len(idx) > 1 else 0].item()

    def _init_step_index(self, timestep):
        self._step_index = self.index_for_timestep(timestep) if self._begin_index is None else self._begin_index

    def scale_model_input(self, sample, timestep):
        if self.step_index is None:
            self._init_step_index(timestep)
        sigma = self.sigmas[self.step_index]
        return sample * float(1 / ((sigma ** 2 + self.config.sigma_data ** 2) ** 0.5))

    def convert_model_output(self, model_output, sample=None):
        c = sa_step_coefficients(self, self.step_index, 1)
        return float(c[1]) * sample + float(c[2]) * model_output

    def dpm_solver_first_order_update(self, model_output, sample=None, noise=None):
        c = sa_step_coefficients(self, self.step_index, 1)
        return (float(c[3]) * sample + float(c[4]) * model_output) + float(c[5]) * noise

    def multistep_dpm_solver_second_order_update(self, model_output_list, sample=None, noise=None):
        c = sa_step_coefficients(self, self.step_index, 2)
        m0, m1 = model_output_list[-1], model_output_list[-2]
        D1 = float(c[6]) * (m0 - m1)
        return ((float(c[3]) * sample + float(c[4]) * m0) + (0.5 * float(c[4])) * D1) + float(c[5]) * noise


def sa_step_coefficients(sched, i, order, zero_z=False):
    """One row of the AED_SA_COEF_STRIDE table (include/aed.h) for step index i, fp32 0-dim-tensor arithmetic in the
    expression order of models.py:1238-1255 and the scheduler's precondition_* / update functions."""
    from ._lib import SA_COEF_STRIDE
    sig, sd = sched.sigmas, sched.config.sigma_data
    s_s, s_t = sig[i], sig[i + 1]
    c = torch.zeros(SA_COEF_STRIDE, dtype=torch.float32)
    c[0] = 1 / ((s_s ** 2 + sd ** 2) ** 0.5)
    c[1] = sd ** 2 / (s_s ** 2 + sd ** 2)
    if sched.config.prediction_type == "v_prediction":
        c[2] = -s_s * sd / (s_s ** 2 + sd ** 2) ** 0.5
    else:
        c[2] = s_s * sd / (s_s ** 2 + sd ** 2) ** 0.5
    h = torch.log(s_s) - torch.log(s_t)
    c[3] = s_t / s_s * torch.exp(-h)
    c[4] = 1 - torch.exp(-2.0 * h)
    c[5] = s_t * torch.sqrt(1.0 - torch.exp(-2 * h))
    if order == 2:
        h_0 = torch.log(sig[i - 1]) - torch.log(s_s)
        c[6] = 1.0 / (h_0 / h)
    c[7] = float(order)
    c[8] = 1.0 if zero_z else 0.0
    c[9] = (2 * np.pi * sched.timesteps[i].cpu().float())          # the DiT's Fourier-feature argument
    return c


def sa_step_orders(sched, start, n_steps, lower_order_nums, first_order=False):
    """Solver order of each of `n_steps` consecutive steps from step index `start`, following the branch conditions of
    models.py:1225-1242 / :1289-1319 (first order when forced, while no history exists, and at the final step)."""
    cfgs = sched.config
    n = len(sched.timesteps)
    out = []
    for i in range(start, start + n_steps):
        final = (i == n - 1) and (cfgs.euler_at_final or (cfgs.lower_order_final and n < 15)
                                  or cfgs.final_sigmas_type == "zero")
        first = first_order or cfgs.solver_order == 1 or lower_order_nums < 1 or final
        out.append(1 if first else 2)
        if lower_order_nums < cfgs.solver_order:
            lower_order_nums += 1
    return out


def sa_coefficient_table(sched, start, n_steps, lower_order_nums, first_order=False, invert=False):
    """[n_steps, AED_SA_COEF_STRIDE]; `invert` marks the step whose noise is defined as zero (models.py:1235-1236)."""
    n = len(sched.timesteps)
    orders = sa_step_orders(sched, start, n_steps, lower_order_nums, first_order)
    rows = [sa_step_coefficients(sched, start + k, orders[k],
                                 zero_z=invert and (start + k == n - 1) and sched.config.final_sigmas_type == "zero")
            for k in range(n_steps)]
    return torch.stack(rows).contiguous()

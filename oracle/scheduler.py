"""Oracle restatement of the DDIM scheduler tables the reference reads.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED at this
boundary: the arithmetic lives in third-party ``diffusers`` (DDIMScheduler,
un-vendored and un-pinned, /root/reference/requirements.txt:1).  Restated from
its published semantics and anchored on the reference's call sites:
  models.py:71-81    alphas_cumprod / timesteps
  models.py:96-97    prev_timestep = t - num_train_timesteps // num_inference_steps
  models.py:687-689  final_alpha_cumprod for prev_timestep < 0
  ddim_inversion.py:82, pc_drift.py:89,250   scheduler.step / _get_variance
  main_run_sdedit.py:86  add_noise
"""
from types import SimpleNamespace

import numpy as np
import torch


class OracleDDIMScheduler:
    def __init__(self, num_train_timesteps=1000, beta_start=0.0015, beta_end=0.0195,
                 beta_schedule="scaled_linear", set_alpha_to_one=False, steps_offset=1,
                 prediction_type="epsilon", timestep_spacing="leading", clip_sample=False):
        if beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps,
                                   dtype=torch.float32) ** 2
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        else:
            raise NotImplementedError(beta_schedule)
        self.betas = betas
        self.alphas = 1.0 - betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps,
                                      prediction_type=prediction_type,
                                      steps_offset=steps_offset,
                                      timestep_spacing=timestep_spacing,
                                      clip_sample=clip_sample,
                                      set_alpha_to_one=set_alpha_to_one,
                                      beta_start=beta_start, beta_end=beta_end,
                                      beta_schedule=beta_schedule)
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    def set_timesteps(self, num_inference_steps, device=None):
        n_train = self.config.num_train_timesteps
        self.num_inference_steps = num_inference_steps
        if self.config.timestep_spacing == "leading":
            step_ratio = n_train // num_inference_steps
            ts = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.int64)
            ts += self.config.steps_offset
        elif self.config.timestep_spacing == "trailing":
            step_ratio = n_train / num_inference_steps
            ts = np.round(np.arange(n_train, 0, -step_ratio)).astype(np.int64) - 1
        elif self.config.timestep_spacing == "linspace":
            ts = np.linspace(0, n_train - 1, num_inference_steps).round()[::-1].copy().astype(np.int64)
        else:
            raise ValueError(self.config.timestep_spacing)
        self.timesteps = torch.from_numpy(ts)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def _get_variance(self, timestep, prev_timestep):
        alpha_prod_t = self.alphas_cumprod[timestep]
        alpha_prod_t_prev = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        beta_prod_t = 1 - alpha_prod_t
        beta_prod_t_prev = 1 - alpha_prod_t_prev
        return (beta_prod_t_prev / beta_prod_t) * (1 - alpha_prod_t / alpha_prod_t_prev)

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False,
             generator=None, variance_noise=None):
        prev_timestep = timestep - self.config.num_train_timesteps // self.num_inference_steps
        alpha_prod_t = self.alphas_cumprod[timestep]
        alpha_prod_t_prev = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        beta_prod_t = 1 - alpha_prod_t
        if self.config.prediction_type == "epsilon":
            pred_original_sample = (sample - beta_prod_t ** 0.5 * model_output) / alpha_prod_t ** 0.5
            pred_epsilon = model_output
        elif self.config.prediction_type == "v_prediction":
            pred_original_sample = (alpha_prod_t ** 0.5) * sample - (beta_prod_t ** 0.5) * model_output
            pred_epsilon = (alpha_prod_t ** 0.5) * model_output + (beta_prod_t ** 0.5) * sample
        else:
            raise ValueError(self.config.prediction_type)
        variance = self._get_variance(timestep, prev_timestep)
        std_dev_t = eta * variance ** 0.5
        pred_sample_direction = (1 - alpha_prod_t_prev - std_dev_t ** 2) ** 0.5 * pred_epsilon
        prev_sample = alpha_prod_t_prev ** 0.5 * pred_original_sample + pred_sample_direction
        if eta > 0:
            if variance_noise is None:
                variance_noise = torch.randn(model_output.shape, generator=generator, dtype=model_output.dtype)
            prev_sample = prev_sample + std_dev_t * variance_noise
        return SimpleNamespace(prev_sample=prev_sample, pred_original_sample=pred_original_sample)

    def add_noise(self, original_samples, noise, timesteps):
        a = self.alphas_cumprod[timesteps] ** 0.5
        s = (1 - self.alphas_cumprod[timesteps]) ** 0.5
        while a.dim() < original_samples.dim():
            a = a.unsqueeze(-1)
            s = s.unsqueeze(-1)
        return a * original_samples + s * noise

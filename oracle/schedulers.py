"""
Oracle: restatement of diffusers' ``PNDMScheduler`` (PLMS branch) and
``DDPMScheduler`` -- the two schedulers the reference constructs for Stage 3
(scripts/stage2/train_gm_unet.py:171-176 PNDM for validation;
scripts/inference/generate_hdr.py:162 DDPM).  TEST INFRASTRUCTURE ONLY.
PARITY UNPINNED against real diffusers; formulas from SURVEY.md Appendix A.2,
pinned by closed-form known-answer tests (tests/test_oracle_schedulers.py).

Protocol used by the pipelines (stable_diffusion_gm.py:216-241, 715, 1037, 1048,
1071; stable_diffusion_dual_unet.py:1037, 1072): ``config``, ``set_timesteps``,
``timesteps``, ``order``, ``init_noise_sigma``, ``scale_model_input``, ``step``,
``alphas_cumprod``; must survive ``copy.deepcopy``.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import torch


def _betas(cfg):
    n = cfg["num_train_timesteps"]
    if cfg["beta_schedule"] == "linear":
        return torch.linspace(cfg["beta_start"], cfg["beta_end"], n, dtype=torch.float32)
    if cfg["beta_schedule"] == "scaled_linear":
        return torch.linspace(cfg["beta_start"] ** 0.5, cfg["beta_end"] ** 0.5, n, dtype=torch.float32) ** 2
    raise NotImplementedError(cfg["beta_schedule"])


class _Config(dict):
    """dict with attribute access (diffusers FrozenDict behaviour the pipelines rely on)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name) from None


class PNDMScheduler:
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 skip_prk_steps=True, set_alpha_to_one=False, prediction_type="epsilon",
                 timestep_spacing="leading", steps_offset=1, clip_sample=False):
        self.config = _Config(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                              beta_schedule=beta_schedule, skip_prk_steps=skip_prk_steps,
                              set_alpha_to_one=set_alpha_to_one, prediction_type=prediction_type,
                              timestep_spacing=timestep_spacing, steps_offset=steps_offset, clip_sample=clip_sample)
        assert skip_prk_steps and prediction_type == "epsilon" and timestep_spacing == "leading"
        self.betas = _betas(self.config)
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.ets, self.counter, self.cur_sample = [], 0, None
        self.timesteps = None

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        base = (np.arange(0, num_inference_steps) * ratio).round() + self.config.steps_offset
        plms = np.concatenate([base[:-1], base[-2:-1], base[-1:]])[::-1].copy()
        self.timesteps = torch.from_numpy(plms.astype(np.int64)).to(device)
        self.ets, self.counter, self.cur_sample = [], 0, None

    def scale_model_input(self, sample, timestep=None):
        return sample

    def step(self, model_output, timestep, sample, return_dict=True):
        timestep = int(timestep)
        ratio = self.config.num_train_timesteps // self.num_inference_steps
        prev = timestep - ratio
        if self.counter != 1:
            self.ets = self.ets[-3:]
            self.ets.append(model_output)
        else:
            prev = timestep
            timestep = timestep + ratio
        e = self.ets
        if len(e) == 1 and self.counter == 0:
            self.cur_sample = sample
        elif len(e) == 1 and self.counter == 1:
            model_output = (model_output + e[-1]) / 2
            sample, self.cur_sample = self.cur_sample, None
        elif len(e) == 2:
            model_output = (3 * e[-1] - e[-2]) / 2
        elif len(e) == 3:
            model_output = (23 * e[-1] - 16 * e[-2] + 5 * e[-3]) / 12
        else:
            model_output = (1 / 24) * (55 * e[-1] - 59 * e[-2] + 37 * e[-3] - 9 * e[-4])
        prev_sample = self._get_prev_sample(sample, timestep, prev, model_output)
        self.counter += 1
        return (prev_sample,) if not return_dict else SimpleNamespace(prev_sample=prev_sample)

    def _get_prev_sample(self, sample, timestep, prev_timestep, model_output):
        a_t = self.alphas_cumprod[timestep]
        a_prev = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        b_t, b_prev = 1 - a_t, 1 - a_prev
        sample_coeff = (a_prev / a_t) ** 0.5
        denom = a_t * b_prev ** 0.5 + (a_t * b_t * a_prev) ** 0.5
        return sample_coeff * sample - (a_prev - a_t) * model_output / denom


class DDPMScheduler:
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 variance_type="fixed_small", clip_sample=False, clip_sample_range=1.0, prediction_type="epsilon",
                 timestep_spacing="leading", steps_offset=1):
        self.config = _Config(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                              beta_schedule=beta_schedule, variance_type=variance_type, clip_sample=clip_sample,
                              clip_sample_range=clip_sample_range, prediction_type=prediction_type,
                              timestep_spacing=timestep_spacing, steps_offset=steps_offset)
        assert variance_type == "fixed_small" and prediction_type == "epsilon" and timestep_spacing == "leading"
        self.betas = _betas(self.config)
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)
        self.one = torch.tensor(1.0)
        self.init_noise_sigma = 1.0
        self.timesteps = None

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64) + self.config.steps_offset
        self.timesteps = torch.from_numpy(ts).to(device)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def _prev(self, t):
        return t - self.config.num_train_timesteps // self.num_inference_steps

    def step(self, model_output, timestep, sample, generator=None, return_dict=True):
        t = int(timestep)
        p = self._prev(t)
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[p] if p >= 0 else self.one
        b_t, b_p = 1 - a_t, 1 - a_p
        cur_a = a_t / a_p
        cur_b = 1 - cur_a
        x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
        if self.config.clip_sample:
            x0 = x0.clamp(-self.config.clip_sample_range, self.config.clip_sample_range)
        x0_coeff = (a_p ** 0.5 * cur_b) / b_t
        xt_coeff = cur_a ** 0.5 * b_p / b_t
        prev = x0_coeff * x0 + xt_coeff * sample
        if t > 0:
            noise = torch.randn(model_output.shape, generator=generator, dtype=model_output.dtype,
                                device=generator.device if generator is not None else model_output.device).to(model_output.device)
            var = torch.clamp(b_p / b_t * cur_b, min=1e-20)
            prev = prev + (var ** 0.5) * noise
        return (prev,) if not return_dict else SimpleNamespace(prev_sample=prev)

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


class DPMSolverMultistepScheduler:
    """diffusers ``DPMSolverMultistepScheduler`` restated for the configuration the reference reaches with
    ``DPMSolverMultistepScheduler.from_config(pipeline.scheduler.config)``
    (scripts/inference/experiments/formal_improved.py:195, scripts/stage2/experiments/scheduler_tuning.py:190-201):
    algorithm ``dpmsolver++``, ``solver_order`` 2, ``midpoint``, epsilon prediction, ``lower_order_final``,
    ``final_sigmas_type="zero"``, no Karras sigmas.  PARITY UNPINNED (diffusers not installed); pinned by the
    closed form: with an x0-consistent model the multistep update is exact and the sampler returns x0."""

    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 solver_order=2, timestep_spacing="leading", steps_offset=1, lower_order_final=True, clip_sample=False):
        self.config = _Config(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                              beta_schedule=beta_schedule, solver_order=solver_order, timestep_spacing=timestep_spacing,
                              steps_offset=steps_offset, lower_order_final=lower_order_final, clip_sample=clip_sample,
                              algorithm_type="dpmsolver++", solver_type="midpoint", prediction_type="epsilon",
                              final_sigmas_type="zero")
        assert solver_order in (1, 2)
        self.betas = _betas(self.config)
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)
        self.init_noise_sigma = 1.0
        self.timesteps = None

    def set_timesteps(self, num_inference_steps, device=None):
        c = self.config
        last = c.num_train_timesteps
        if c.timestep_spacing == "linspace":
            ts = np.linspace(0, last - 1, num_inference_steps + 1).round()[::-1][:-1].copy().astype(np.int64)
        elif c.timestep_spacing == "leading":
            ratio = last // (num_inference_steps + 1)
            ts = (np.arange(0, num_inference_steps + 1) * ratio).round()[::-1][:-1].copy().astype(np.int64) + c.steps_offset
        else:
            raise NotImplementedError(c.timestep_spacing)
        sig = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()
        sig = np.interp(ts, np.arange(0, len(sig)), sig)
        self.sigmas = torch.from_numpy(np.concatenate([sig, [0.0]]).astype(np.float32))
        self.timesteps = torch.from_numpy(ts).to(device)
        self.num_inference_steps = len(ts)
        self.model_outputs = [None] * c.solver_order
        self.lower_order_nums = 0
        self._step_index = None

    def scale_model_input(self, sample, timestep=None):
        return sample

    @staticmethod
    def _alpha_sigma(sigma):
        alpha_t = 1 / ((sigma ** 2 + 1) ** 0.5)
        return alpha_t, sigma * alpha_t

    def step(self, model_output, timestep, sample, generator=None, return_dict=True):
        if self._step_index is None:
            idx = (self.timesteps.cpu() == int(timestep)).nonzero()
            self._step_index = int(idx[1] if len(idx) > 1 else idx[0])
        i, n = self._step_index, len(self.timesteps)
        lower_final = i == n - 1  # final_sigmas_type == "zero"
        lower_second = i == n - 2 and self.config.lower_order_final and n < 15
        a_s0, s_s0 = self._alpha_sigma(self.sigmas[i])
        x0 = (sample - s_s0 * model_output) / a_s0  # convert_model_output (dpmsolver++, epsilon)
        for k in range(self.config.solver_order - 1):
            self.model_outputs[k] = self.model_outputs[k + 1]
        self.model_outputs[-1] = x0
        a_t, s_t = self._alpha_sigma(self.sigmas[i + 1])
        lam_t, lam_s0 = torch.log(a_t) - torch.log(s_t), torch.log(a_s0) - torch.log(s_s0)
        h = lam_t - lam_s0
        if self.config.solver_order == 1 or self.lower_order_nums < 1 or lower_final:
            prev = (s_t / s_s0) * sample - (a_t * (torch.exp(-h) - 1.0)) * x0
        else:
            a_s1, s_s1 = self._alpha_sigma(self.sigmas[i - 1])
            lam_s1 = torch.log(a_s1) - torch.log(s_s1)
            m0, m1 = self.model_outputs[-1], self.model_outputs[-2]
            h_0 = lam_s0 - lam_s1
            r0 = h_0 / h
            d0, d1 = m0, (1.0 / r0) * (m0 - m1)
            prev = (s_t / s_s0) * sample - (a_t * (torch.exp(-h) - 1.0)) * d0 - 0.5 * (a_t * (torch.exp(-h) - 1.0)) * d1
        if self.lower_order_nums < self.config.solver_order:
            self.lower_order_nums += 1
        self._step_index += 1
        return (prev,) if not return_dict else SimpleNamespace(prev_sample=prev)

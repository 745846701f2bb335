"""
Schedulers for the MI355X build: ``PNDMScheduler`` (PLMS, the "50 PNDM steps" configuration of
scripts/stage2/train_gm_unet.py:171-176) and ``DDPMScheduler``
(scripts/inference/generate_hdr.py:162).  In the reference both come from ``diffusers``; these
classes keep the protocol the pipelines rely on (stable_diffusion_gm.py:216-241, 610-625, 715,
1037, 1048, 1071; stable_diffusion_dual_unet.py:1037, 1072): ``config`` (dict-like, attribute
access), ``set_timesteps``, ``timesteps``, ``order``, ``init_noise_sigma``,
``scale_model_input``, ``step(...)``, ``alphas_cumprod``, ``from_config`` and survival under
``copy.deepcopy``.

The schedulers are host-side state machines; the per-element update runs in the
``gmd_latent_step`` HIP kernel for device tensors (same float32 operation order as the torch
expressions, so the two agree bit for bit) and in plain torch for host tensors, which is what the
reference itself executes on CPU tensors.
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np
import torch

from .. import hip_ops as ops
from .configuration import ConfigMixin
from .image_processor import _Output, randn_tensor


@dataclass
class SchedulerOutput(_Output):
    prev_sample: torch.Tensor


def _betas(beta_schedule, beta_start, beta_end, n, trained_betas=None):
    if trained_betas is not None:
        return torch.tensor(trained_betas, dtype=torch.float32)
    if beta_schedule == "linear":
        return torch.linspace(beta_start, beta_end, n, dtype=torch.float32)
    if beta_schedule == "scaled_linear":
        return torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2
    raise NotImplementedError(f"{beta_schedule} is not implemented")


class _SchedulerBase(ConfigMixin):
    config_name = "scheduler_config.json"
    order = 1

    @classmethod
    def from_pretrained(cls, path, subfolder=None, **overrides):
        d = os.path.join(path, subfolder) if subfolder else path
        return cls.from_config(cls.load_config(d), **overrides)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def __len__(self):
        return self.config.num_train_timesteps


class PNDMScheduler(_SchedulerBase):
    _defaults = dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                     trained_betas=None, skip_prk_steps=False, set_alpha_to_one=False, prediction_type="epsilon",
                     timestep_spacing="leading", steps_offset=0, clip_sample=False)

    def __init__(self, **kwargs):
        cfg = dict(self._defaults)
        bad = [k for k in kwargs if k not in cfg]
        if bad:
            raise TypeError(f"PNDMScheduler: unexpected arguments {bad}")
        cfg.update(kwargs)
        self.register_to_config(**cfg)
        if cfg["prediction_type"] != "epsilon":
            raise NotImplementedError("only epsilon prediction is implemented")
        self.betas = _betas(cfg["beta_schedule"], cfg["beta_start"], cfg["beta_end"], cfg["num_train_timesteps"], cfg["trained_betas"])
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if cfg["set_alpha_to_one"] else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.pndm_order = 4
        self.cur_model_output = 0
        self.counter = 0
        self.cur_sample = None
        self.ets = []
        self.num_inference_steps = None
        self._timesteps = np.arange(0, cfg["num_train_timesteps"])[::-1].copy()
        self.prk_timesteps = None
        self.plms_timesteps = None
        self.timesteps = None

    def set_timesteps(self, num_inference_steps, device=None):
        c = self.config
        self.num_inference_steps = num_inference_steps
        if c.timestep_spacing == "linspace":
            self._timesteps = np.linspace(0, c.num_train_timesteps - 1, num_inference_steps).round().astype(np.int64)
        elif c.timestep_spacing == "leading":
            ratio = c.num_train_timesteps // num_inference_steps
            self._timesteps = (np.arange(0, num_inference_steps) * ratio).round()
            self._timesteps += c.steps_offset
        elif c.timestep_spacing == "trailing":
            ratio = c.num_train_timesteps / num_inference_steps
            self._timesteps = np.round(np.arange(c.num_train_timesteps, 0, -ratio))[::-1].astype(np.int64)
            self._timesteps -= 1
        else:
            raise ValueError(f"{c.timestep_spacing} is not supported")
        if c.skip_prk_steps:
            self.prk_timesteps = np.array([])
            self.plms_timesteps = np.concatenate([self._timesteps[:-1], self._timesteps[-2:-1], self._timesteps[-1:]])[::-1].copy()
        else:
            raise NotImplementedError("Runge-Kutta warm-up (skip_prk_steps=False) is not implemented; SD-1.5 uses skip_prk_steps=True")
        timesteps = np.concatenate([self.prk_timesteps, self.plms_timesteps]).astype(np.int64)
        self.timesteps = torch.from_numpy(timesteps).to(device)
        self.ets = []
        self.counter = 0
        self.cur_model_output = 0
        self.cur_sample = None

    # ---- PLMS planning (host) ------------------------------------------------------------------
    def _plan(self, timestep):
        """Which PLMS branch the next step takes: (mode, effective timestep, previous timestep)."""
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        timestep = int(timestep)
        ratio = self.config.num_train_timesteps // self.num_inference_steps
        prev = timestep - ratio
        n_after = len(self.ets[-3:]) + 1 if self.counter != 1 else len(self.ets)
        if self.counter == 1:
            prev, timestep = timestep, timestep + ratio
        if n_after == 1 and self.counter == 0:
            mode = 0
        elif n_after == 1 and self.counter == 1:
            mode = 1
        else:
            mode = min(n_after, 4)
        return mode, timestep, prev

    def _coefs(self, timestep, prev_timestep):
        """diffusers ``_get_prev_sample`` coefficients, evaluated on float32 0-d tensors exactly as there."""
        a_t = self.alphas_cumprod[timestep]
        a_prev = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        b_t, b_prev = 1 - a_t, 1 - a_prev
        sample_coeff = (a_prev / a_t) ** 0.5
        denom = a_t * b_prev ** 0.5 + (a_t * b_t * a_prev) ** 0.5
        return sample_coeff, a_prev - a_t, denom

    def _commit(self, mode, eps, sample):
        if self.counter != 1:
            self.ets = self.ets[-3:]
            self.ets.append(eps)
        if mode == 0:
            self.cur_sample = sample
        elif mode == 1:
            self.cur_sample = None
        self.counter += 1

    def step(self, model_output, timestep, sample, return_dict=True):
        mode, t_eff, prev = self._plan(timestep)
        sc, ad, dn = self._coefs(t_eff, prev)
        if model_output.is_cuda:
            hist = [e for e in reversed(self.ets[-3:])] if self.counter != 1 else [self.ets[-1]]
            x = sample.contiguous()
            nh = {0: 0, 1: 1, 2: 1, 3: 2, 4: 3}[mode]
            eps_copy, prev_sample, _ = ops.latent_step(model_output.contiguous(), x, mode, (sc.item(), ad.item(), dn.item(), 1.0, 0.0),
                                                       False, 1.0, cur_sample=self.cur_sample, hist=hist[:nh])
            # keep the kernel's private copy in the history: `model_output` may be the static output buffer of a
            # captured graph that the next replay overwrites
            self._commit(mode, eps_copy, sample)
        else:
            e = (self.ets[-3:] + [model_output]) if self.counter != 1 else self.ets
            smp = sample
            if mode == 0:
                m = model_output
            elif mode == 1:
                m = (model_output + e[-1]) / 2
                smp = self.cur_sample
            elif mode == 2:
                m = (3 * e[-1] - e[-2]) / 2
            elif mode == 3:
                m = (23 * e[-1] - 16 * e[-2] + 5 * e[-3]) / 12
            else:
                m = (1 / 24) * (55 * e[-1] - 59 * e[-2] + 37 * e[-3] - 9 * e[-4])
            prev_sample = sc * smp - ad * m / dn
            self._commit(mode, model_output, sample)
        return (prev_sample,) if not return_dict else SchedulerOutput(prev_sample=prev_sample)

    def fused_step(self, eps_in, timestep, sample, do_cfg, guidance_scale, guidance_rescale=0.0, want_x0=False):
        """CFG combine (+rescale) + x0 + PLMS update in ONE HIP kernel pass (device tensors only).
        eps_in: raw UNet output ([2B,...] when do_cfg).  Returns (prev_sample, x0 | None)."""
        mode, t_eff, prev = self._plan(timestep)
        sc, ad, dn = self._coefs(t_eff, prev)
        a = self.alphas_cumprod[int(timestep)]  # dual_unet.py:1072 uses the loop timestep
        ratio = None
        if do_cfg and guidance_rescale > 0.0:
            ratio = ops.cfg_std_ratio(eps_in, guidance_scale)
        hist = [e for e in reversed(self.ets[-3:])] if self.counter != 1 else [self.ets[-1]]
        nh = {0: 0, 1: 1, 2: 1, 3: 2, 4: 3}[mode]
        eps, prev_sample, x0 = ops.latent_step(eps_in, sample.contiguous(), mode,
                                               (sc.item(), ad.item(), dn.item(), a.sqrt().item(), (1 - a).sqrt().item()),
                                               do_cfg, guidance_scale, cur_sample=self.cur_sample, hist=hist[:nh], ratio=ratio,
                                               guidance_rescale=guidance_rescale, want_x0=want_x0)
        self._commit(mode, eps, sample)
        return prev_sample, x0


class DDPMScheduler(_SchedulerBase):
    _defaults = dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear", trained_betas=None,
                     variance_type="fixed_small", clip_sample=True, prediction_type="epsilon", thresholding=False,
                     dynamic_thresholding_ratio=0.995, clip_sample_range=1.0, sample_max_value=1.0,
                     timestep_spacing="leading", steps_offset=0, rescale_betas_zero_snr=False)

    def __init__(self, **kwargs):
        cfg = dict(self._defaults)
        bad = [k for k in kwargs if k not in cfg]
        if bad:
            raise TypeError(f"DDPMScheduler: unexpected arguments {bad}")
        cfg.update(kwargs)
        self.register_to_config(**cfg)
        if cfg["prediction_type"] != "epsilon" or cfg["thresholding"] or cfg["rescale_betas_zero_snr"]:
            raise NotImplementedError("only epsilon prediction without thresholding / zero-SNR rescale is implemented")
        if cfg["variance_type"] not in ("fixed_small", "fixed_small_log", "fixed_large"):
            raise NotImplementedError(cfg["variance_type"])
        self.betas = _betas(cfg["beta_schedule"], cfg["beta_start"], cfg["beta_end"], cfg["num_train_timesteps"], cfg["trained_betas"])
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.one = torch.tensor(1.0)
        self.init_noise_sigma = 1.0
        self.custom_timesteps = False
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, cfg["num_train_timesteps"])[::-1].copy())

    def set_timesteps(self, num_inference_steps=None, device=None, timesteps=None):
        c = self.config
        if timesteps is not None:
            timesteps = np.array(timesteps, dtype=np.int64)
            self.custom_timesteps = True
            self.num_inference_steps = len(timesteps)
        else:
            if num_inference_steps > c.num_train_timesteps:
                raise ValueError("num_inference_steps cannot exceed num_train_timesteps")
            self.num_inference_steps = num_inference_steps
            self.custom_timesteps = False
            if c.timestep_spacing == "linspace":
                timesteps = np.linspace(0, c.num_train_timesteps - 1, num_inference_steps).round()[::-1].copy().astype(np.int64)
            elif c.timestep_spacing == "leading":
                ratio = c.num_train_timesteps // num_inference_steps
                timesteps = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
                timesteps += c.steps_offset
            elif c.timestep_spacing == "trailing":
                ratio = c.num_train_timesteps / num_inference_steps
                timesteps = np.round(np.arange(c.num_train_timesteps, 0, -ratio)).astype(np.int64) - 1
            else:
                raise ValueError(f"{c.timestep_spacing} is not supported")
        self.timesteps = torch.from_numpy(timesteps).to(device)

    def previous_timestep(self, timestep):
        if self.custom_timesteps:
            idx = (self.timesteps == timestep).nonzero(as_tuple=True)[0][0]
            return torch.tensor(-1) if idx == self.timesteps.shape[0] - 1 else self.timesteps[idx + 1]
        n = self.num_inference_steps if self.num_inference_steps else self.config.num_train_timesteps
        return timestep - self.config.num_train_timesteps // n

    def _get_variance(self, t):
        prev_t = self.previous_timestep(t)
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        cur_b = 1 - a_t / a_p
        variance = torch.clamp((1 - a_p) / (1 - a_t) * cur_b, min=1e-20)
        vt = self.config.variance_type
        if vt == "fixed_small_log":
            variance = torch.exp(0.5 * torch.log(variance))
        elif vt == "fixed_large":
            variance = cur_b
        return variance

    def draws_noise(self, timestep):
        """True when ``step`` at this timestep consumes the generator (every step but t == 0)."""
        return int(timestep) > 0

    def _device_step(self, eps_in, timestep, sample, do_cfg, guidance_scale, guidance_rescale, want_x0, generator, noise=None):
        """One HIP kernel pass (gmd_ddpm_step): CFG combine (+rescale), pipeline x0, clipped x0 prediction, posterior mean
        and the variance noise.  The noise is drawn HERE with ``randn_tensor`` exactly where ``step`` draws it, so the
        generator the dual pipeline shares between its two schedulers (stable_diffusion_dual_unet.py:1015, 1077, 1093) is
        consumed in the reference's order."""
        t = int(timestep)
        prev_t = int(self.previous_timestep(t))
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        b_t, b_p = 1 - a_t, 1 - a_p
        cur_a = a_t / a_p
        cur_b = 1 - cur_a
        x0_coeff = (a_p ** 0.5 * cur_b) / b_t
        xt_coeff = cur_a ** 0.5 * b_p / b_t
        scale = 0.0  # (noise: the caller's pre-drawn tensor, or drawn below)
        if t > 0:
            if noise is None:  # (the pipelines pre-draw a CPU generator's noise for all steps, in call order: see fused_step)
                noise = randn_tensor(sample.shape, generator=generator, device=sample.device, dtype=torch.float32)
            v = self._get_variance(t)
            scale = (v if self.config.variance_type == "fixed_small_log" else v ** 0.5).item()
        ratio = ops.cfg_std_ratio(eps_in, guidance_scale) if (do_cfg and guidance_rescale > 0.0) else None
        return ops.ddpm_step(eps_in.contiguous(), sample.contiguous(),
                             ((a_t ** 0.5).item(), (b_t ** 0.5).item(), x0_coeff.item(), xt_coeff.item(), scale,
                              a_t.sqrt().item(), (1 - a_t).sqrt().item()),
                             do_cfg, guidance_scale, noise=noise, ratio=ratio, guidance_rescale=guidance_rescale,
                             clip_range=self.config.clip_sample_range if self.config.clip_sample else None, want_x0=want_x0)

    def fused_step(self, eps_in, timestep, sample, do_cfg, guidance_scale, guidance_rescale=0.0, want_x0=False, generator=None, noise=None):
        """Same contract as ``PNDMScheduler.fused_step`` plus the generator (device float32 tensors only).
        ``noise``: this step's variance noise already drawn from ``generator`` by the caller (a CPU generator forces a
        synchronous host draw + copy per step; the pipelines draw all steps up front, in the order the steps consume
        them, so the host keeps running ahead of the GPU).  Returns (prev_sample, x0 | None)."""
        return self._device_step(eps_in, timestep, sample, do_cfg, guidance_scale, guidance_rescale, want_x0, generator, noise)

    def step(self, model_output, timestep, sample, generator=None, return_dict=True, noise=None):
        if model_output.is_cuda and model_output.dtype == torch.float32 and sample.dtype == torch.float32:
            prev, _ = self._device_step(model_output, timestep, sample, False, 1.0, 0.0, False, generator, noise)
            return (prev,) if not return_dict else SchedulerOutput(prev_sample=prev)
        return self._host_step(model_output, timestep, sample, generator, return_dict)

    def _host_step(self, model_output, timestep, sample, generator=None, return_dict=True):
        """The torch expressions of diffusers' ``DDPMScheduler.step`` (host tensors; also the reference for the kernel test)."""
        t = int(timestep)
        prev_t = int(self.previous_timestep(t))
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        b_t, b_p = 1 - a_t, 1 - a_p
        cur_a = a_t / a_p
        cur_b = 1 - cur_a
        dev = model_output.device
        x0 = (sample - (b_t ** 0.5).to(dev) * model_output) / (a_t ** 0.5).to(dev)
        if self.config.clip_sample:
            x0 = x0.clamp(-self.config.clip_sample_range, self.config.clip_sample_range)
        x0_coeff = ((a_p ** 0.5 * cur_b) / b_t).to(dev)
        xt_coeff = (cur_a ** 0.5 * b_p / b_t).to(dev)
        prev = x0_coeff * x0 + xt_coeff * sample
        if t > 0:
            noise = randn_tensor(model_output.shape, generator=generator, device=dev, dtype=model_output.dtype)
            v = self._get_variance(t).to(dev)
            prev = prev + (v * noise if self.config.variance_type == "fixed_small_log" else (v ** 0.5) * noise)
        return (prev,) if not return_dict else SchedulerOutput(prev_sample=prev)


class DPMSolverMultistepScheduler(_SchedulerBase):
    """DPM-Solver++ multistep scheduler: the one the reference swaps in for the dual-UNet text->HDR runs
    (``DPMSolverMultistepScheduler.from_config(pipeline.scheduler.config)``,
    scripts/inference/experiments/formal_improved.py:195; scripts/stage2/experiments/scheduler_tuning.py:190-201).
    Implements diffusers' ``dpmsolver++`` / ``midpoint`` / epsilon-prediction path with ``solver_order`` 1-2,
    ``lower_order_final`` and ``final_sigmas_type`` zero / sigma_min; Karras / exponential / beta sigma schedules,
    SDE variants and thresholding raise NotImplementedError.  A host-side state machine over the scheduler protocol:
    ``step`` / ``fused_step`` run as ONE HIP kernel (gmd_dpm_step) for float32 device tensors, as the same torch
    expressions on the host otherwise."""

    _defaults = dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear", trained_betas=None,
                     solver_order=2, prediction_type="epsilon", thresholding=False, dynamic_thresholding_ratio=0.995,
                     sample_max_value=1.0, algorithm_type="dpmsolver++", solver_type="midpoint", lower_order_final=True,
                     euler_at_final=False, use_karras_sigmas=False, use_exponential_sigmas=False, use_beta_sigmas=False,
                     final_sigmas_type="zero", timestep_spacing="linspace", steps_offset=0, clip_sample=False)

    def __init__(self, **kwargs):
        cfg = dict(self._defaults)
        bad = [k for k in kwargs if k not in cfg]
        if bad:
            raise TypeError(f"DPMSolverMultistepScheduler: unexpected arguments {bad}")
        cfg.update(kwargs)
        self.register_to_config(**cfg)
        if (cfg["algorithm_type"] != "dpmsolver++" or cfg["solver_type"] != "midpoint" or cfg["prediction_type"] != "epsilon"
                or cfg["thresholding"] or cfg["use_karras_sigmas"] or cfg["use_exponential_sigmas"] or cfg["use_beta_sigmas"]
                or cfg["euler_at_final"] or cfg["solver_order"] not in (1, 2)):
            raise NotImplementedError("only dpmsolver++ / midpoint / epsilon, solver_order <= 2, plain sigmas are implemented")
        self.betas = _betas(cfg["beta_schedule"], cfg["beta_start"], cfg["beta_end"], cfg["num_train_timesteps"], cfg["trained_betas"])
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.alpha_t = torch.sqrt(self.alphas_cumprod)
        self.sigma_t = torch.sqrt(1 - self.alphas_cumprod)
        self.lambda_t = torch.log(self.alpha_t) - torch.log(self.sigma_t)
        self.sigmas = ((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.linspace(0, cfg["num_train_timesteps"] - 1, cfg["num_train_timesteps"], dtype=np.float32)[::-1].copy())
        self.model_outputs = [None] * cfg["solver_order"]
        self.lower_order_nums = 0
        self._step_index = None

    @property
    def step_index(self):
        return self._step_index

    def set_timesteps(self, num_inference_steps=None, device=None):
        c = self.config
        last = c.num_train_timesteps
        if c.timestep_spacing == "linspace":
            ts = np.linspace(0, last - 1, num_inference_steps + 1).round()[::-1][:-1].copy().astype(np.int64)
        elif c.timestep_spacing == "leading":
            ratio = last // (num_inference_steps + 1)
            ts = (np.arange(0, num_inference_steps + 1) * ratio).round()[::-1][:-1].copy().astype(np.int64)
            ts += c.steps_offset
        elif c.timestep_spacing == "trailing":
            ratio = c.num_train_timesteps / num_inference_steps
            ts = np.arange(last, 0, -ratio).round().copy().astype(np.int64) - 1
        else:
            raise ValueError(f"{c.timestep_spacing} is not supported")
        sig = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()
        sig = np.interp(ts, np.arange(0, len(sig)), sig)
        if c.final_sigmas_type == "sigma_min":
            sigma_last = ((1 - self.alphas_cumprod[0]) / self.alphas_cumprod[0]) ** 0.5
        elif c.final_sigmas_type == "zero":
            sigma_last = 0
        else:
            raise ValueError(f"`final_sigmas_type` must be one of 'zero', or 'sigma_min', but got {c.final_sigmas_type}")
        self.sigmas = torch.from_numpy(np.concatenate([sig, [sigma_last]]).astype(np.float32))
        self.timesteps = torch.from_numpy(ts).to(device=device, dtype=torch.int64)
        self._ts_host = [int(v) for v in ts]
        self.num_inference_steps = len(ts)
        self.model_outputs = [None] * c.solver_order
        self.lower_order_nums = 0
        self._step_index = None

    @staticmethod
    def _sigma_to_alpha_sigma_t(sigma):
        alpha_t = 1 / ((sigma ** 2 + 1) ** 0.5)
        return alpha_t, sigma * alpha_t

    def _init_step_index(self, timestep):
        t = int(timestep)
        idx = [k for k, v in enumerate(self._ts_host) if v == t]
        if not idx:
            self._step_index = len(self._ts_host) - 1
        else:
            self._step_index = idx[1] if len(idx) > 1 else idx[0]

    def _plan_step(self, timestep):
        """Host side of one step: (order, float32 0-dim coefficient tensors) computed exactly as diffusers computes them."""
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        if self._step_index is None:
            self._init_step_index(timestep)
        c = self.config
        i, n = self._step_index, len(self._ts_host)
        lower_order_final = (i == n - 1) and (c.euler_at_final or (c.lower_order_final and n < 15) or c.final_sigmas_type == "zero")
        lower_order_second = (i == n - 2) and c.lower_order_final and n < 15
        alpha_s0, sigma_s0 = self._sigma_to_alpha_sigma_t(self.sigmas[i])
        alpha_t, sigma_t = self._sigma_to_alpha_sigma_t(self.sigmas[i + 1])
        lambda_t = torch.log(alpha_t) - torch.log(sigma_t)
        lambda_s0 = torch.log(alpha_s0) - torch.log(sigma_s0)
        h = lambda_t - lambda_s0
        first = c.solver_order == 1 or self.lower_order_nums < 1 or lower_order_final
        r0 = None
        if not first:
            assert c.solver_order == 2 or self.lower_order_nums < 2 or lower_order_second
            alpha_s1, sigma_s1 = self._sigma_to_alpha_sigma_t(self.sigmas[i - 1])
            lambda_s1 = torch.log(alpha_s1) - torch.log(sigma_s1)
            r0 = (lambda_s0 - lambda_s1) / h
        return first, alpha_s0, sigma_s0, alpha_t, sigma_t, h, r0

    def _advance(self, x0_pred):
        c = self.config
        for k in range(c.solver_order - 1):
            self.model_outputs[k] = self.model_outputs[k + 1]
        self.model_outputs[-1] = x0_pred
        if self.lower_order_nums < c.solver_order:
            self.lower_order_nums += 1
        self._step_index += 1

    def _device_step(self, eps_in, timestep, sample, do_cfg, guidance_scale, guidance_rescale, want_x0):
        """One HIP kernel pass (gmd_dpm_step): CFG combine (+rescale), pipeline x0, x0 prediction and the multistep update."""
        first, alpha_s0, sigma_s0, alpha_t, sigma_t, h, r0 = self._plan_step(timestep)
        c_x = sigma_t / sigma_s0
        c_m = alpha_t * (torch.exp(-h) - 1.0)
        c_h = 0.5 * c_m
        inv_r0 = (1.0 / r0) if r0 is not None else torch.tensor(0.0)
        a = self.alphas_cumprod[int(timestep)]  # the pipeline's own x0 (dual_unet.py:1072) uses the loop timestep
        ratio = ops.cfg_std_ratio(eps_in, guidance_scale) if (do_cfg and guidance_rescale > 0.0) else None
        m1 = None if first else self.model_outputs[-1]
        m0, prev, x0 = ops.dpm_step(eps_in.contiguous(), sample.contiguous(), 1 if first else 2,
                                    (sigma_s0.item(), alpha_s0.item(), c_x.item(), c_m.item(), c_h.item(), inv_r0.item(),
                                     a.sqrt().item(), (1 - a).sqrt().item()),
                                    do_cfg, guidance_scale, m1=m1, ratio=ratio, guidance_rescale=guidance_rescale, want_x0=want_x0)
        self._advance(m0)
        return prev, x0

    def fused_step(self, eps_in, timestep, sample, do_cfg, guidance_scale, guidance_rescale=0.0, want_x0=False, generator=None):
        """Same contract as ``PNDMScheduler.fused_step`` (device tensors only; ``generator`` is accepted because ``step`` has
        the parameter, and unused: this solver is deterministic).  Returns (prev_sample, x0 | None)."""
        return self._device_step(eps_in, timestep, sample, do_cfg, guidance_scale, guidance_rescale, want_x0)

    def step(self, model_output, timestep, sample, generator=None, variance_noise=None, return_dict=True):
        if model_output.is_cuda and model_output.dtype == torch.float32 and sample.dtype == torch.float32:
            prev, _ = self._device_step(model_output, timestep, sample, False, 1.0, 0.0, False)
            return (prev,) if not return_dict else SchedulerOutput(prev_sample=prev)
        first, alpha_s0, sigma_s0, alpha_t, sigma_t, h, r0 = self._plan_step(timestep)
        x0_pred = (sample - sigma_s0 * model_output) / alpha_s0  # convert_model_output: dpmsolver++, epsilon
        m1 = self.model_outputs[-1]
        sample = sample.to(torch.float32)
        if first:
            prev = (sigma_t / sigma_s0) * sample - (alpha_t * (torch.exp(-h) - 1.0)) * x0_pred
        else:
            D0, D1 = x0_pred, (1.0 / r0) * (x0_pred - m1)
            prev = ((sigma_t / sigma_s0) * sample - (alpha_t * (torch.exp(-h) - 1.0)) * D0
                    - 0.5 * (alpha_t * (torch.exp(-h) - 1.0)) * D1)
        self._advance(x0_pred)
        prev = prev.to(model_output.dtype)
        return (prev,) if not return_dict else SchedulerOutput(prev_sample=prev)

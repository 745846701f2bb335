"""
Oracle: the two Stage-3 denoising loops restated from the reference, as plain
functions over pre-computed prompt embeddings.  TEST INFRASTRUCTURE ONLY.

* :func:`gm_loop`   follows gm_diffusion/pipelines/stable_diffusion_gm.py:1003-1091
* :func:`dual_loop` follows gm_diffusion/pipelines/stable_diffusion_dual_unet.py:1001-1113
  with the batched GM-embedding slice of
  scripts/inference/experiments/visualize_latents.py:274
  (``prompt_embeds[negative_prompt_embeds.shape[0]:]``; equals the reference's
  ``prompt_embeds[1:]`` when B == 1 with CFG).
* :func:`decode_tail` follows scripts/inference/generate_hdr.py:225-265.
"""
from __future__ import annotations

import copy
import inspect

import numpy as np
import torch

from . import hdr_ops


def rescale_noise_cfg(noise_cfg, noise_pred_text, guidance_rescale=0.0):
    """stable_diffusion_gm.py:71-94"""
    std_text = noise_pred_text.std(dim=list(range(1, noise_pred_text.ndim)), keepdim=True)
    std_cfg = noise_cfg.std(dim=list(range(1, noise_cfg.ndim)), keepdim=True)
    rescaled = noise_cfg * (std_text / std_cfg)
    return guidance_rescale * rescaled + (1 - guidance_rescale) * noise_cfg


def _cfg(noise_pred, guidance_scale, guidance_rescale):
    u, t = noise_pred.chunk(2)
    out = u + guidance_scale * (t - u)
    if guidance_rescale > 0.0:
        out = rescale_noise_cfg(out, t, guidance_rescale)
    return out


@torch.no_grad()
def gm_loop(unet, scheduler, sdr_latent, prompt_embeds, negative_prompt_embeds, latents,
            num_inference_steps=50, guidance_scale=7.5, guidance_rescale=0.0, generator=None, record=None):
    """Single-UNet GM denoise conditioned on ``sdr_latent`` (A1)."""
    do_cfg = guidance_scale > 1
    embeds = torch.cat([negative_prompt_embeds, prompt_embeds]) if do_cfg else prompt_embeds
    scheduler.set_timesteps(num_inference_steps)
    latents = latents * scheduler.init_noise_sigma
    step_kw = {"generator": generator} if "generator" in inspect.signature(scheduler.step).parameters else {}
    for t in scheduler.timesteps:
        cat_latents = torch.cat([sdr_latent, latents], dim=1)  # gm.py:1045 conditioning first
        x = torch.cat([cat_latents] * 2) if do_cfg else cat_latents
        x = scheduler.scale_model_input(x, t)
        noise_pred = unet(x, t, encoder_hidden_states=embeds, return_dict=False)[0]
        if do_cfg:
            noise_pred = _cfg(noise_pred, guidance_scale, guidance_rescale)
        latents = scheduler.step(noise_pred, t, latents, **step_kw, return_dict=False)[0]
        if record is not None:
            record.append(latents.clone())
    return latents


@torch.no_grad()
def dual_loop(unet, gm_unet, scheduler, prompt_embeds, negative_prompt_embeds, latents,
              num_inference_steps=50, guidance_scale=7.5, guidance_rescale=0.0, generator=None, record=None, added_cond=None):
    """Joint SDR-UNet (CFG) + GM-UNet (conditional only, fed the SDR x0 prediction) loop (A2).

    ``added_cond`` (SDXL-style UNets only; the reference has no such path, BASELINE.json configs[4]): dict with ``text_embeds`` /
    ``time_ids`` of the prompts and ``negative_text_embeds`` / ``negative_time_ids`` (default: the same time_ids) of the
    unconditional half, batched exactly like the prompt embeddings: [negative; positive] for the SDR UNet, positive for the GM UNet."""
    do_cfg = guidance_scale > 1
    embeds = torch.cat([negative_prompt_embeds, prompt_embeds]) if do_cfg else prompt_embeds
    gm_embeds = embeds[negative_prompt_embeds.shape[0]:] if do_cfg else embeds  # vis.py:274
    sdr_kw, gm_kw = {}, {}
    if added_cond is not None:
        pos = dict(text_embeds=added_cond["text_embeds"], time_ids=added_cond["time_ids"])
        gm_kw = dict(added_cond_kwargs=pos)
        if do_cfg:
            neg_ids = added_cond.get("negative_time_ids", added_cond["time_ids"])
            sdr_kw = dict(added_cond_kwargs=dict(text_embeds=torch.cat([added_cond["negative_text_embeds"], pos["text_embeds"]]),
                                                 time_ids=torch.cat([neg_ids, pos["time_ids"]])))
        else:
            sdr_kw = dict(added_cond_kwargs=pos)
    scheduler.set_timesteps(num_inference_steps)
    latents = latents * scheduler.init_noise_sigma
    gm_latents = latents.clone()  # dual.py:1012: both streams start from the same noise
    gm_scheduler = copy.deepcopy(scheduler)  # dual.py:1037, after set_timesteps
    step_kw = {"generator": generator} if "generator" in inspect.signature(scheduler.step).parameters else {}
    for t in scheduler.timesteps:
        x = torch.cat([latents] * 2) if do_cfg else latents
        x = scheduler.scale_model_input(x, t)
        gm_latents = gm_scheduler.scale_model_input(gm_latents, t)
        eps = unet(x, t, encoder_hidden_states=embeds, return_dict=False, **sdr_kw)[0]
        if do_cfg:
            eps = _cfg(eps, guidance_scale, guidance_rescale)
        a = scheduler.alphas_cumprod.to(eps.device)[t].view(-1, 1, 1, 1)  # dual.py:1072
        x0 = (latents - (1 - a).sqrt() * eps) / a.sqrt()  # pre-step latents
        latents = scheduler.step(eps, t, latents, **step_kw, return_dict=False)[0]
        gm_in = torch.cat([x0, gm_latents], dim=1)  # dual.py:1080
        gm_eps = gm_unet(gm_in, t, encoder_hidden_states=gm_embeds, return_dict=False, **gm_kw)[0]
        gm_latents = gm_scheduler.step(gm_eps, t, gm_latents, **step_kw, return_dict=False)[0]
        if record is not None:
            record.append((latents.clone(), gm_latents.clone()))
    return latents, gm_latents


@torch.no_grad()
def decode_tail(vae, sdr_latent, gm_latent, qmax=99, clamp=False):
    """generate_hdr.py:225-265 / formal_improved.py:272-303: decode both latents,
    denorm + clamp, u8 PNG bytes, Eq.1 (numpy variant: no clamp), /(qmax+1)."""
    sf = vae.config.scaling_factor
    sdr_dec = vae.decode(1 / sf * sdr_latent, return_dict=False)[0]  # generate_hdr.py:225
    gm_dec = vae.decode(1 / sf * gm_latent, return_dict=False)[0]  # generate_hdr.py:230
    out = hdr_ops.hdr_tail(sdr_dec.float().numpy(), gm_dec.float().numpy(), qmax=qmax, clamp=clamp)
    out["sdr_dec"] = sdr_dec.float().numpy()
    out["gm_dec"] = gm_dec.float().numpy()
    return out


def latent_rms(a, b):
    a = torch.as_tensor(np.asarray(a), dtype=torch.float64)
    b = torch.as_tensor(np.asarray(b), dtype=torch.float64)
    return float(((a - b) ** 2).mean().sqrt())

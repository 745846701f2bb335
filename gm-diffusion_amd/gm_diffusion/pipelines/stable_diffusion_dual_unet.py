# Portions of this file keep API-surface text (the ``__call__`` / constructor signatures, ``encode_prompt``, ``check_inputs``
# error messages, the property block and the output tail) of diffusers' ``StableDiffusionPipeline``, which the reference's
# pipeline files are copies of with ~40 changed lines each (their header, reproduced as the Apache License 2.0 requires):
#
#     Copyright 2024 The HuggingFace Team. All rights reserved.
#
#     Licensed under the Apache License, Version 2.0 (the "License");
#     you may not use this file except in compliance with the License.
#     You may obtain a copy of the License at
#
#         http://www.apache.org/licenses/LICENSE-2.0
#
#     Unless required by applicable law or agreed to in writing, software
#     distributed under the License is distributed on an "AS IS" BASIS,
#     WITHOUT WARRANTIES OR CONDITIONS OF ANY KIND, either express or implied.
#     See the License for the specific language governing permissions and
#     limitations under the License.
#
# Changes from that text: the LoRA / textual-inversion / IP-adapter branches are removed, the denoising loop, latent step,
# UNet / VAE calls and the decode tail are new (hand-written HIP kernels behind gm_diffusion.hip_ops).
"""
``StableDiffusionDualUNetPipeline`` -- the Stage-3 text->HDR path: an SDR UNet (with
classifier-free guidance) and a GM UNet (conditional only, fed the SDR x0-prediction) are stepped
jointly with two scheduler instances.  Drop-in mirror of the reference class at
gm_diffusion/pipelines/stable_diffusion_dual_unet.py:156 (constructor :202-214, ``__call__``
:782-1140).

Per iteration (reference lines):
  1045-1060  SDR UNet on cat([latents]*2)                      -> pack kernel (CFG duplicate) + HIP UNet
  1063-1069  CFG combine (+ rescale_noise_cfg)                 \\
  1071-1075  x0 = (x - sqrt(1-a) eps) / sqrt(a)                  > ONE fused HIP kernel (gmd_latent_step)
  1077       latents = scheduler.step(eps, t, latents)         /
  1080       gm_in = cat([x0, gm_latents], dim=1)              -> folded into the GM UNet input pack
  1083-1092  GM UNet on gm_in with the conditional embeddings  -> HIP UNet
  1093       gm_latents = gm_scheduler.step(...)               -> gmd_latent_step (no CFG)

Generalisations over the reference as written: the GM UNet receives the
conditional half ``prompt_embeds[negative.shape[0]:]`` (scripts/inference/experiments/
visualize_latents.py:274), which equals the reference's ``prompt_embeds[1:]`` for its only working
case (batch 1 + CFG) and also works for batch > 1 and without CFG; with a non-latent
``output_type`` BOTH latents are decoded (the reference decodes neither correctly,
dual_unet.py:1117-1132).  As in the reference the result is always the bare tuple
``(sdr, gm)`` and ``return_dict`` is ignored (:1132); ``callback_on_step_end`` is accepted and
ignored (:1095-1103 is commented out there).
"""
from __future__ import annotations

import copy
from typing import Any, Callable, Dict, List, Optional, Union

import torch

from .. import hip_ops as ops
from .stable_diffusion_gm import _GMPipelineBase, rescale_noise_cfg, retrieve_timesteps

__all__ = ["StableDiffusionDualUNetPipeline", "rescale_noise_cfg", "retrieve_timesteps"]


class StableDiffusionDualUNetPipeline(_GMPipelineBase):
    # Run the GM UNet on a second HIP stream, one step behind the SDR UNet (GM(i) needs only x0_i, SDR(i+1) only
    # latents_{i+1}).  Measured on MI355X (bench.py): eager launches 3.04 vs 3.04 images/s (the host, ~10 us per launch,
    # cannot keep two streams fed); with each forward replayed from a captured HIP graph 3.04 -> 3.86 images/s.
    overlap_streams = True
    # Launch-plan family of the two UNet forwards (hip_ops.plan_family): None = by regime -- the co-running family when the forwards
    # overlap on two streams, the launch-by-launch plans when they share one stream; True / False pin it (bench.py pins True for
    # its instrumented single-stream step so that it times, and compares bit for bit, the kernels the shipped path runs)
    co_run_plans = None
    _step_probe = None  # test hook: callable(i, sdr_latents, gm_latents) after every loop iteration

    def __init__(self, vae, text_encoder, tokenizer, unet, gm_unet, scheduler, safety_checker, feature_extractor,
                 image_encoder=None, requires_safety_checker: bool = True):
        self._init_common(vae, text_encoder, tokenizer, unet, scheduler, safety_checker, feature_extractor, image_encoder,
                          requires_safety_checker, gm_unet=gm_unet)

    @staticmethod
    def _batched_added_cond(added_cond, do_cfg, device):
        """(SDR UNet's, GM UNet's) added_cond_kwargs: [negative; positive] rows for the guidance batch, the positive rows for the
        GM UNet (it runs the conditional half only, like its prompt embeddings); (None, None) without added conditioning."""
        if not added_cond:
            return None, None
        pos = {k: added_cond[k].to(device) for k in ("text_embeds", "time_ids")}
        if not do_cfg:
            return pos, pos
        if "negative_text_embeds" not in added_cond:
            raise ValueError("classifier-free guidance with an SDXL-style UNet needs added_cond_kwargs['negative_text_embeds']")
        neg_ids = added_cond.get("negative_time_ids", added_cond["time_ids"]).to(device)
        return dict(text_embeds=torch.cat([added_cond["negative_text_embeds"].to(device), pos["text_embeds"]]),
                    time_ids=torch.cat([neg_ids, pos["time_ids"]])), pos

    def _gm_stream(self, device):
        return ops.side_stream(device)  # one per device for the whole process, not per pipeline object (see there)

    @torch.no_grad()
    def __call__(
        self,
        prompt: Union[str, List[str]] = None,
        height: Optional[int] = None,
        width: Optional[int] = None,
        num_inference_steps: int = 50,
        timesteps: List[int] = None,
        sigmas: List[float] = None,
        guidance_scale: float = 7.5,
        negative_prompt: Optional[Union[str, List[str]]] = None,
        num_images_per_prompt: Optional[int] = 1,
        eta: float = 0.0,
        generator: Optional[Union[torch.Generator, List[torch.Generator]]] = None,
        latents: Optional[torch.Tensor] = None,
        prompt_embeds: Optional[torch.Tensor] = None,
        negative_prompt_embeds: Optional[torch.Tensor] = None,
        ip_adapter_image=None,
        ip_adapter_image_embeds: Optional[List[torch.Tensor]] = None,
        output_type: Optional[str] = "pil",
        return_dict: bool = True,
        cross_attention_kwargs: Optional[Dict[str, Any]] = None,
        guidance_rescale: float = 0.0,
        clip_skip: Optional[int] = None,
        callback_on_step_end: Optional[Callable[[Any, int, Any, Dict], Dict]] = None,
        callback_on_step_end_tensor_inputs: List[str] = ["latents"],
        **kwargs,
    ):
        callback, callback_steps = self._pop_legacy_callbacks(kwargs)
        # EXTENSION beyond the reference (which has no SDXL path: its added_cond_kwargs carry IP-adapter image embeddings only,
        # stable_diffusion_gm.py:1022-1026): for UNets with addition_embed_type "text_time" the caller passes
        # added_cond_kwargs = {"text_embeds", "time_ids"[, "negative_text_embeds", "negative_time_ids"]}; every other unknown
        # keyword is ignored as in the reference
        added_cond = kwargs.pop("added_cond_kwargs", None)
        if hasattr(callback_on_step_end, "tensor_inputs"):
            callback_on_step_end_tensor_inputs = callback_on_step_end.tensor_inputs
        height, width = self._default_hw(height, width)
        self.check_inputs(prompt, height, width, callback_steps, negative_prompt, prompt_embeds, negative_prompt_embeds,
                          ip_adapter_image, ip_adapter_image_embeds, callback_on_step_end_tensor_inputs)
        self._guidance_scale = guidance_scale
        self._guidance_rescale = guidance_rescale
        self._clip_skip = clip_skip
        self._cross_attention_kwargs = cross_attention_kwargs
        self._interrupt = False

        if prompt is not None and isinstance(prompt, str):
            batch_size = 1
        elif prompt is not None and isinstance(prompt, list):
            batch_size = len(prompt)
        else:
            batch_size = prompt_embeds.shape[0]
        device = self._execution_device
        lora_scale = self.cross_attention_kwargs.get("scale", None) if self.cross_attention_kwargs is not None else None
        prompt_embeds, negative_prompt_embeds = self.encode_prompt(
            prompt, device, num_images_per_prompt, self.do_classifier_free_guidance, negative_prompt,
            prompt_embeds=prompt_embeds, negative_prompt_embeds=negative_prompt_embeds, lora_scale=lora_scale,
            clip_skip=self.clip_skip)
        do_cfg = self.do_classifier_free_guidance
        n_neg = negative_prompt_embeds.shape[0] if do_cfg else 0
        if do_cfg:
            prompt_embeds = torch.cat([negative_prompt_embeds, prompt_embeds])
        gm_prompt_embeds = prompt_embeds[n_neg:]  # conditional half (vis.py:274)

        timesteps, num_inference_steps = retrieve_timesteps(self.scheduler, num_inference_steps, device, timesteps, sigmas)

        num_channels_latents = self.unet.config.in_channels
        latents = self.prepare_latents(batch_size * num_images_per_prompt, num_channels_latents, height, width,
                                       self._latent_dtype(prompt_embeds, device), device, generator, latents)
        gm_latents = latents.clone()  # dual.py:1012: both streams start from the same (scaled) noise
        extra_step_kwargs = self.prepare_extra_step_kwargs(generator, eta)

        num_warmup_steps = len(timesteps) - num_inference_steps * self.scheduler.order
        self._num_timesteps = len(timesteps)
        self.gm_scheduler = copy.deepcopy(self.scheduler)  # dual.py:1037, after set_timesteps

        sdr_added, gm_added = self._batched_added_cond(added_cond, do_cfg, latents.device)
        fused = self._use_fused(latents, self.unet, self.scheduler) and self._use_fused(latents, self.gm_unet, self.gm_scheduler)
        if fused:
            ctx = self.unet.prepare_context(prompt_embeds)
            gm_ctx = self.gm_unet.prepare_context(gm_prompt_embeds)
            h, w = latents.shape[-2:]
            # constant over the loop: written once into the persistent buffers the (captured) forwards read
            self.unet.set_added_cond(sdr_added, (2 if do_cfg else 1) * latents.shape[0])
            self.gm_unet.set_added_cond(gm_added, latents.shape[0])
            # The GM UNet of step i needs only x0_i; the SDR UNet of step i+1 needs only latents_{i+1}: the two are
            # independent, so the GM stream runs on its own HIP stream one step behind the SDR stream and their
            # kernels overlap (the batch-B GM kernels alone cannot fill 256 CUs).
            sdr_stream = torch.cuda.current_stream(latents.device)
            gm_stream = self._gm_stream(latents.device) if self.overlap_streams else sdr_stream
            ts_host = [int(v) for v in timesteps.tolist()]          # host copy: no device sync inside the loop
            ts_dev = timesteps.to(device=latents.device, dtype=torch.float32)
            g_sdr = g_gm = None
            shared = self._cfg_shared(self.unet, do_cfg)
            nb_sdr = (2 if do_cfg else 1) * latents.shape[0]
            # two forwards in flight side by side: their contractions take the co-running plan family (fewest L2 -> LDS bytes per
            # product; -2.2 % per batch against the launch-by-launch plans, +8.6 % if the streams were serialised:
            # profiles/r05_ab_plan_default.txt), a single-stream run keeps the plans that are fastest alone
            co_run = (gm_stream is not sdr_stream) if self.co_run_plans is None else bool(self.co_run_plans)
            if self._graphs_ok():
                g_sdr = self.unet.graphed_forward(nb_sdr, h, w, ctx, cfg_shared=shared, co_run=co_run)
                g_gm = self.gm_unet.graphed_forward(latents.shape[0], h, w, gm_ctx, co_run=co_run)
            pre = self._predraw_step_noise([self.scheduler, self.gm_scheduler], ts_host, latents.shape, generator, latents.device)
            gm_stream.wait_stream(sdr_stream)
            if gm_stream is not sdr_stream:
                # allocated on the caller's stream, consumed (and released) by step 0 on the GM stream: without this the
                # allocator may hand the block to the SDR stream's step 1 while GM step 0 still reads it
                gm_latents.record_stream(gm_stream)

        with self.progress_bar(total=num_inference_steps) as progress_bar:
            for i, t in enumerate(timesteps):
                if self.interrupt:
                    continue
                if fused:
                    x = self.unet.pack_input(latents, dup=2 if (do_cfg and not shared) else 1, out=g_sdr.x if g_sdr else None)
                    self.unet.set_timestep_from(ts_dev, i)
                    if g_sdr:
                        sdr_noise_pred = g_sdr.replay()
                    else:
                        with ops.plan_family(co_run):
                            sdr_noise_pred = self.unet.forward_packed(x, nb_sdr, h, w, ctx, cfg_shared=shared)
                    pre_step = latents
                    latents, x0_latent = self.scheduler.fused_step(sdr_noise_pred, ts_host[i], pre_step, do_cfg, self.guidance_scale,
                                                                   self.guidance_rescale if do_cfg else 0.0, want_x0=True,
                                                                   **self._fused_step_kwargs(extra_step_kwargs),
                                                                   **({"noise": pre[0][i]} if pre else {}))
                    with torch.cuda.stream(gm_stream):
                        gm_stream.wait_stream(sdr_stream)  # x0_i is ready
                        x0_latent.record_stream(gm_stream)
                        gx = self.gm_unet.pack_input((x0_latent, gm_latents), dup=1, out=g_gm.x if g_gm else None)
                        self.gm_unet.set_timestep_from(ts_dev, i)
                        if g_gm:
                            gm_noise_pred = g_gm.replay()
                        else:
                            with ops.plan_family(co_run):
                                gm_noise_pred = self.gm_unet.forward_packed(gx, gx.shape[0], h, w, gm_ctx)
                        gm_latents = self.gm_scheduler.step(gm_noise_pred, ts_host[i], gm_latents,
                                                            **self._fused_step_kwargs(extra_step_kwargs),
                                                            **({"noise": pre[1][i]} if pre else {}), return_dict=False)[0]
                else:
                    latent_model_input = torch.cat([latents] * 2) if do_cfg else latents
                    latent_model_input = self.scheduler.scale_model_input(latent_model_input, t)
                    gm_latents = self.gm_scheduler.scale_model_input(gm_latents, t)
                    sdr_noise_pred = self.unet(latent_model_input, t, encoder_hidden_states=prompt_embeds, timestep_cond=None,
                                               cross_attention_kwargs=self.cross_attention_kwargs, added_cond_kwargs=sdr_added,
                                               return_dict=False)[0]
                    if do_cfg:
                        sdr_noise_pred_uncond, sdr_noise_pred_text = sdr_noise_pred.chunk(2)
                        sdr_noise_pred = sdr_noise_pred_uncond + self.guidance_scale * (sdr_noise_pred_text - sdr_noise_pred_uncond)
                    if do_cfg and self.guidance_rescale > 0.0:
                        sdr_noise_pred = rescale_noise_cfg(sdr_noise_pred, sdr_noise_pred_text, guidance_rescale=self.guidance_rescale)
                    # x0-prediction (dual.py:1071-1075), from the PRE-step latents
                    alphas_cumprod = self.scheduler.alphas_cumprod.to(sdr_noise_pred.device)[t].view(-1, 1, 1, 1)
                    sqrt_alpha_cumprod = alphas_cumprod.sqrt()
                    sqrt_one_minus_alpha_cumprod = (1 - alphas_cumprod).sqrt()
                    x0_latent = (latents - sqrt_one_minus_alpha_cumprod * sdr_noise_pred) / sqrt_alpha_cumprod
                    latents = self.scheduler.step(sdr_noise_pred, t, latents, **extra_step_kwargs, return_dict=False)[0]
                    gm_latent_input = torch.cat([x0_latent, gm_latents], dim=1)
                    gm_noise_pred = self.gm_unet(gm_latent_input, t, encoder_hidden_states=gm_prompt_embeds, timestep_cond=None,
                                                 cross_attention_kwargs=self.cross_attention_kwargs, added_cond_kwargs=gm_added,
                                                 return_dict=False)[0]
                    gm_latents = self.gm_scheduler.step(gm_noise_pred, t, gm_latents, **extra_step_kwargs, return_dict=False)[0]

                if self._step_probe is not None:
                    # test hook (tests/test_northstar_gpu.py): the reference's callback block is commented out
                    # (stable_diffusion_dual_unet.py:1095-1103) and its legacy callback sees the SDR latents only; the probe
                    # gets BOTH latents of iteration i after joining the two streams (this serialises them: not for timing)
                    if fused:
                        torch.cuda.synchronize(latents.device)
                    self._step_probe(i, latents, gm_latents)
                if i == len(timesteps) - 1 or ((i + 1) > num_warmup_steps and (i + 1) % self.scheduler.order == 0):
                    progress_bar.update()
                    if callback is not None and i % callback_steps == 0:
                        step_idx = i // getattr(self.scheduler, "order", 1)
                        callback(step_idx, t, latents)

        if fused:
            sdr_stream.wait_stream(gm_stream)  # join: the caller sees both results on its own stream
            gm_latents.record_stream(sdr_stream)
        self._check_f32_range("StableDiffusionDualUNetPipeline latents", self.unet, latents)
        self._check_f32_range("StableDiffusionDualUNetPipeline GM latents", self.gm_unet, gm_latents)
        if output_type == "latent":
            return (latents, gm_latents)
        sf = self.vae.config.scaling_factor
        sdr_image = self.vae.decode(latents / sf, return_dict=False, generator=generator)[0]
        gm_image = self.vae.decode(gm_latents / sf, return_dict=False, generator=generator)[0]
        self._check_f32_range("StableDiffusionDualUNetPipeline decoded images", self.vae, sdr_image, gm_image)
        do_denormalize = [True] * sdr_image.shape[0]
        sdr_out = self.image_processor.postprocess(sdr_image, output_type=output_type, do_denormalize=do_denormalize)
        gm_out = self.image_processor.postprocess(gm_image, output_type=output_type, do_denormalize=do_denormalize)
        self.maybe_free_model_hooks()
        return (sdr_out, gm_out)

"""
Oracle fixtures: seeded models + synthetic inputs shared by tests, smoke and the
bench's cpu_baseline leg.  TEST INFRASTRUCTURE ONLY.

Inputs follow SURVEY.md §8(d): weights from default torch initialisers under
``torch.manual_seed(1234)``; embeddings N(0,1) from Generator(1); initial latents
from Generator(42), drawn for the FULL batch (then sharded by the caller).
"""
from __future__ import annotations

import numpy as np
import torch

from . import pipelines, schedulers, unet as ounet, vae as ovae

WEIGHT_SEED = 1234
EMBED_SEED = 1
LATENT_SEED = 42


def build_unet(kind="tiny", in_channels=4, seed=WEIGHT_SEED, **over):
    torch.manual_seed(seed + in_channels)  # SDR (4ch) and GM (8ch) UNets get different weights
    cfg = ounet.tiny_unet_config(in_channels) if kind == "tiny" else dict(in_channels=in_channels)
    cfg.update(over)
    return ounet.UNet2DConditionModel(**cfg).eval().requires_grad_(False)


def build_vae(kind="tiny", seed=WEIGHT_SEED, with_encoder=False, **over):
    torch.manual_seed(seed + 100)
    cfg = ovae.tiny_vae_config() if kind == "tiny" else {}
    cfg.update(over)
    return ovae.AutoencoderKL(with_encoder=with_encoder, **cfg).eval().requires_grad_(False)


def make_inputs(batch, h, w, cross_dim=768, seq=77):
    ge = torch.Generator("cpu").manual_seed(EMBED_SEED)
    pos = torch.randn(batch, seq, cross_dim, generator=ge)
    neg = torch.randn(batch, seq, cross_dim, generator=ge)
    gl = torch.Generator("cpu").manual_seed(LATENT_SEED)
    latents = torch.randn(batch, 4, h, w, generator=gl)
    return pos, neg, latents


def _np(d):
    return {k: (v.numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def fixture_gm_tiny():
    """BASELINE config-1 shape at reduced width: single 8-ch UNet, 1 prompt, 256x256
    (latent 32x32), 10 PNDM steps, fp32."""
    unet = build_unet("tiny", 8)
    pos, neg, lat = make_inputs(1, 32, 32, cross_dim=unet.config.cross_attention_dim)
    gs = torch.Generator("cpu").manual_seed(7)
    sdr_latent = torch.randn(1, 4, 32, 32, generator=gs) * 0.7
    rec = []
    out = pipelines.gm_loop(unet, schedulers.PNDMScheduler(), sdr_latent, pos, neg, lat,
                            num_inference_steps=10, guidance_scale=7.5, record=rec)
    return _np(dict(prompt_embeds=pos, negative_prompt_embeds=neg, latents=lat, sdr_latent=sdr_latent,
                    out=out, per_step=torch.stack(rec)))


def fixture_dual_tiny():
    """Dual-UNet loop, B=2 (exercises the batched GM-embedding slice), 16x16 latents, 10 PNDM steps."""
    unet, gm_unet = build_unet("tiny", 4), build_unet("tiny", 8)
    pos, neg, lat = make_inputs(2, 16, 16, cross_dim=unet.config.cross_attention_dim)
    rec = []
    sdr, gm = pipelines.dual_loop(unet, gm_unet, schedulers.PNDMScheduler(), pos, neg, lat,
                                  num_inference_steps=10, guidance_scale=7.5, record=rec)
    vae = build_vae("tiny")
    tail = pipelines.decode_tail(vae, sdr, gm, qmax=99)
    d = dict(prompt_embeds=pos, negative_prompt_embeds=neg, latents=lat, sdr_out=sdr, gm_out=gm,
             sdr_per_step=torch.stack([r[0] for r in rec]), gm_per_step=torch.stack([r[1] for r in rec]))
    d.update({"tail_" + k: v for k, v in tail.items()})
    return _np(d)


def fixture_dual_tiny_rescale():
    """Same as dual_tiny with guidance_rescale=0.7 and 6 steps (covers rescale_noise_cfg and the
    low-order PLMS branches only)."""
    unet, gm_unet = build_unet("tiny", 4), build_unet("tiny", 8)
    pos, neg, lat = make_inputs(2, 16, 16, cross_dim=unet.config.cross_attention_dim)
    sdr, gm = pipelines.dual_loop(unet, gm_unet, schedulers.PNDMScheduler(), pos, neg, lat,
                                  num_inference_steps=6, guidance_scale=5.0, guidance_rescale=0.7)
    return _np(dict(prompt_embeds=pos, negative_prompt_embeds=neg, latents=lat, sdr_out=sdr, gm_out=gm))


DUAL_SD15_RECORD_EVERY = 5  # iterations 0, 5, ..., 50 of the 51 are kept


def fixture_dual_sd15_512(steps=50):
    """The north-star path at its own width (BASELINE config 2 with one prompt): SD-1.5-width SDR + GM UNets, 64x64 latent
    (512x512), ``steps`` PNDM steps (steps + 1 iterations), CFG 7.5, float32 -- stable_diffusion_dual_unet.py:1040-1093 as
    formal_improved.py:199 runs it.  ~150 CPU UNet evaluations: minutes.  Inputs are reproducible by seed (build_unet("sd15", 4 | 8),
    make_inputs(1, 64, 64), build_vae("sd15")), so only outputs are stored: the final latent pair, every 5th iteration's pair and a
    128x128 crop of the decoded tail (generate_hdr.py:225-265) from the SD-1.5-width VAE decoder."""
    unet, gm_unet = build_unet("sd15", 4), build_unet("sd15", 8)
    pos, neg, lat = make_inputs(1, 64, 64)
    rec = []
    sdr, gm = pipelines.dual_loop(unet, gm_unet, schedulers.PNDMScheduler(), pos, neg, lat,
                                  num_inference_steps=steps, guidance_scale=7.5, record=rec)
    idx = list(range(0, len(rec), DUAL_SD15_RECORD_EVERY))
    d = dict(steps=np.int64(steps), record_index=np.asarray(idx, dtype=np.int64), sdr_out=sdr, gm_out=gm,
             sdr_per_step=torch.stack([rec[i][0] for i in idx]), gm_per_step=torch.stack([rec[i][1] for i in idx]),
             latents_checksum=np.float64(lat.double().sum().item()), embeds_checksum=np.float64(pos.double().sum().item()))
    del unet, gm_unet
    tail = pipelines.decode_tail(build_vae("sd15"), sdr, gm, qmax=99)
    c = slice(192, 320)
    for k in ("sdr_dec", "gm_dec"):  # NCHW
        d["tail_" + k + "_crop"] = tail[k][:, :, c, c]
    for k in ("sdr", "gm", "hdr", "hdr_norm", "sdr_u8", "gm_u8"):  # NHWC
        if k in tail:
            d["tail_" + k + "_crop"] = tail[k][:, c, c, :]
    return _np(d)


PIPELINE_FIXTURES = {
    "gm_tiny": fixture_gm_tiny,
    "dual_tiny": fixture_dual_tiny,
    "dual_tiny_rescale": fixture_dual_tiny_rescale,
}
# minutes of CPU each: only regenerated on request (make_golden.py --slow)
SLOW_PIPELINE_FIXTURES = {
    "dual_sd15_512": fixture_dual_sd15_512,
}

#!/usr/bin/env python
"""Drift of the full-size dual-UNet pipeline (SD-1.5 widths, synthetic weights) over the whole 50-step PNDM trajectory: bfloat16,
float16 and the exact float32 FMA kernels against the float32 pipeline on the matrix cores (the tolerance path); latents, per step and
in the decoded images.  Diagnostic: the parity gate itself is float32 vs the CPU oracle (tests/test_pipeline_gpu.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gm-diffusion_amd")):
    sys.path.insert(0, p)
import torch
from gm_diffusion import hdr
from gm_diffusion.components import AutoencoderKL, PNDMScheduler, UNet2DConditionModel
from gm_diffusion.pipelines import StableDiffusionDualUNetPipeline


def build(dt):
    u = UNet2DConditionModel(in_channels=4).init_random(1234).to("cuda", dt)
    g = UNet2DConditionModel(in_channels=8).init_random(1238).to("cuda", dt)
    v = AutoencoderKL().init_random(1334).to("cuda", dt)
    s = PNDMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", skip_prk_steps=True, steps_offset=1, set_alpha_to_one=False)
    p = StableDiffusionDualUNetPipeline(vae=v, text_encoder=None, tokenizer=None, unet=u, gm_unet=g, scheduler=s, safety_checker=None,
                                        feature_extractor=None, requires_safety_checker=False)
    p.set_progress_bar_config(disable=True)
    return p


steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
g = torch.Generator().manual_seed(0)
pe, ne = torch.randn(1, 77, 768, generator=g).cuda(), torch.randn(1, 77, 768, generator=g).cuda()
lat = torch.randn(1, 4, 64, 64, generator=g).cuda()
res = {}
from gm_diffusion import hip_ops
# "f32" = the float32 pipeline on the matrix cores (three float16 products per float32 product: the tolerance path, reference of
# the drift figures); "f32 exact" = the float32 FMA kernels
for name, dt, mode in (("f32", torch.float32, "split"), ("f32 exact", torch.float32, "exact"), ("f16", torch.float16, "split"), ("bf16", torch.bfloat16, "split")):
    prev = hip_ops.set_f32_mode(mode)
    pipe = build(dt)
    if mode == "exact":
        pipe.use_hip_graphs = False
    rec = []
    sdr, gm = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, latents=lat, height=512, width=512, num_inference_steps=steps,
                   guidance_scale=7.5, output_type="latent", callback=lambda i, t, x: rec.append(x.float().clone()), callback_steps=1)
    out = hdr.decode_to_hdr(pipe.vae, sdr, gm, qmax=99.0, want=("sdr", "gm", "hdr"))
    res[name] = (sdr.float(), gm.float(), {k: v.float() for k, v in out.items()}, rec)
    del pipe
    torch.cuda.empty_cache()
    hip_ops.set_f32_mode(prev)
rms = lambda a, b: float(((a.double() - b.double()) ** 2).mean().sqrt())
a = res["f32"]
lr_s, lr_g = float(a[0].pow(2).mean().sqrt()), float(a[1].pow(2).mean().sqrt())
print(f"steps={steps}, 1 prompt, 512x512, SD-1.5 widths, synthetic weights; reference = float32 on the matrix cores; latent RMS: sdr {lr_s:.3f} gm {lr_g:.3f}")
for name in ("f32 exact", "f16", "bf16"):
    b = res[name]
    print(f"{name:10s} latent RMS vs f32: sdr {rms(a[0], b[0]):.3e} (rel {rms(a[0], b[0]) / lr_s:.2e})  gm {rms(a[1], b[1]):.3e} (rel {rms(a[1], b[1]) / lr_g:.2e})")
    print(f"{'':10s} per-step SDR latent RMS (every 5th):", " ".join(f"{rms(x, y):.1e}" for x, y in list(zip(a[3], b[3]))[::5]))
    for k in ("sdr", "gm", "hdr"):
        print(f"{'':10s} decoded {k}: RMS diff {rms(a[2][k], b[2][k]):.3e}  (mean |value| {float(a[2][k].abs().mean()):.3f})")

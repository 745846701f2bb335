#!/usr/bin/env python
"""Concurrent timeline of the shipped two-stream pipeline (BASELINE config 2: SD-1.5 dual UNet, 512x512, batch 4, bf16, captured graphs,
GM stream one step behind): a one-thread stamp kernel (gmd_stamp: the device's 100 MHz counter) sits at every block boundary of BOTH
UNet forwards -- recorded into the captured graphs, so the replays stamp themselves -- and writes into row i of a table for loop
iteration i.  rocprofv3 cannot show this (its tracing serialises the two streams); the stamps cost 11 one-thread
launches per forward.  Prints, for a few iterations in the middle of the loop, when each UNet entered / left each block on the common
device clock, the overlap of the two forwards, and the per-block durations with the other stream running beside them against the
same blocks with the streams serialised (--no-overlap run of the same table)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gm-diffusion_amd")):
    sys.path.insert(0, p)
import torch
from gm_diffusion import hdr
from gm_diffusion.components import AutoencoderKL, PNDMScheduler, UNet2DConditionModel
from gm_diffusion.pipelines import StableDiffusionDualUNetPipeline

dev = torch.device("cuda", 0)
B, steps = 4, 50
NAMES = ["down0 (64x64)", "down1 (32x32)", "down2 (16x16)", "down3 (8x8)", "mid (8x8)", "up0 (8x8)", "up1 (16x16)", "up2 (32x32)", "up3 (64x64)", "norm_out+conv_out"]


def build():
    unet = UNet2DConditionModel(in_channels=4).init_random(1234, device=dev).to(dev, torch.bfloat16)
    gm = UNet2DConditionModel(in_channels=8).init_random(1238, device=dev).to(dev, torch.bfloat16)
    vae = AutoencoderKL().init_random(1334, device=dev).to(dev, torch.bfloat16)
    sched = PNDMScheduler(num_train_timesteps=1000, skip_prk_steps=True, set_alpha_to_one=False, beta_start=0.00085, beta_end=0.012,
                          beta_schedule="scaled_linear", steps_offset=1)
    p = StableDiffusionDualUNetPipeline(vae=vae, text_encoder=None, tokenizer=None, unet=unet, gm_unet=gm, scheduler=sched, safety_checker=None,
                                        feature_extractor=None, requires_safety_checker=False)
    p.set_progress_bar_config(disable=True)
    for m in (unet, gm):
        m._stamp_buf = torch.zeros(steps + 1, 16, dtype=torch.int64, device=dev)
        m._stamp_row = torch.zeros(1, dtype=torch.int32, device=dev)
    return p


def run(p, overlap):
    p.overlap_streams = overlap
    ge = torch.Generator("cpu").manual_seed(1)
    pos, neg = torch.randn(B, 77, 768, generator=ge).to(dev), torch.randn(B, 77, 768, generator=ge).to(dev)
    lat = torch.randn(B, 4, 64, 64, generator=torch.Generator("cpu").manual_seed(42)).to(dev)
    for _ in range(2):  # capture, then one warm replayed run
        sdr, gm = p(prompt_embeds=pos, negative_prompt_embeds=neg, latents=lat, height=512, width=512, num_inference_steps=steps,
                    guidance_scale=7.5, output_type="latent")
    torch.cuda.synchronize()
    return p.unet._stamp_buf.cpu().numpy().astype("float64") / 100.0, p.gm_unet._stamp_buf.cpu().numpy().astype("float64") / 100.0  # us


p = build()
s_ov, g_ov = run(p, True)
s_se, g_se = run(p, False)
print("Two-stream pipeline, device clock (us); SDR UNet at CFG batch 8, GM UNet at batch 4, one HIP graph replay each per iteration")
for it in (20, 21, 22):
    t0 = s_ov[it, 0]
    print(f"iteration {it}: SDR forward {s_ov[it, 0] - t0:8.1f} .. {s_ov[it, 10] - t0:8.1f}   GM forward of iteration {it - 1} {g_ov[it - 1, 0] - t0:8.1f} .. {g_ov[it - 1, 10] - t0:8.1f}"
          f"   GM forward of iteration {it} {g_ov[it, 0] - t0:8.1f} .. {g_ov[it, 10] - t0:8.1f}   next SDR starts {s_ov[it + 1, 0] - t0:8.1f}")
its = range(5, 45)
per_iter_ov = (s_ov[44, 0] - s_ov[5, 0]) / 39
per_iter_se = (s_se[44, 0] - s_se[5, 0]) / 39
print(f"\nloop period: {per_iter_ov:.1f} us per iteration with the two streams, {per_iter_se:.1f} us with the streams serialised (--no-overlap)")
print(f"{'block':22s} {'SDR overlapped':>15s} {'SDR serial':>11s} {'GM overlapped':>14s} {'GM serial':>10s}   (mean us over iterations 5..44)")
for k, nm in enumerate(NAMES):
    so = sum(s_ov[i, k + 1] - s_ov[i, k] for i in its) / len(its)
    ss = sum(s_se[i, k + 1] - s_se[i, k] for i in its) / len(its)
    go = sum(g_ov[i, k + 1] - g_ov[i, k] for i in its) / len(its)
    gs = sum(g_se[i, k + 1] - g_se[i, k] for i in its) / len(its)
    print(f"{nm:22s} {so:15.1f} {ss:11.1f} {go:14.1f} {gs:10.1f}")
so = sum(s_ov[i, 10] - s_ov[i, 0] for i in its) / len(its); ss = sum(s_se[i, 10] - s_se[i, 0] for i in its) / len(its)
go = sum(g_ov[i, 10] - g_ov[i, 0] for i in its) / len(its); gs = sum(g_se[i, 10] - g_se[i, 0] for i in its) / len(its)
print(f"{'whole forward':22s} {so:15.1f} {ss:11.1f} {go:14.1f} {gs:10.1f}")
# fraction of the SDR forward's span during which a GM forward is also in flight
ov = 0.0
for i in its:
    a0, a1 = s_ov[i, 0], s_ov[i, 10]
    for j in (i - 1, i):
        b0, b1 = g_ov[j, 0], g_ov[j, 10]
        ov += max(0.0, min(a1, b1) - max(a0, b0))
print(f"a GM forward is in flight during {ov / sum(s_ov[i, 10] - s_ov[i, 0] for i in its) * 100:.0f} % of the SDR forwards' time")

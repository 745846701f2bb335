#!/usr/bin/env python
"""Probe: does the chip absorb more than two concurrent UNet streams?  Runs P independent dual-UNet pipelines (own model
objects, own streams, batch B/P each, one host thread each) against one pipeline at batch B.  Diagnostic tool."""
import argparse, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gm-diffusion_amd")):
    sys.path.insert(0, p)
import torch
from gm_diffusion.components import AutoencoderKL, PNDMScheduler, UNet2DConditionModel
from gm_diffusion.pipelines import StableDiffusionDualUNetPipeline


def build(seed):
    dev, dt = "cuda", torch.bfloat16
    u = UNet2DConditionModel(in_channels=4).init_random(seed).to(dev, dt)
    g = UNet2DConditionModel(in_channels=8).init_random(seed + 1).to(dev, dt)
    v = AutoencoderKL().init_random(seed + 2).to(dev, dt)
    s = PNDMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", skip_prk_steps=True, steps_offset=1, set_alpha_to_one=False)
    p = StableDiffusionDualUNetPipeline(vae=v, text_encoder=None, tokenizer=None, unet=u, gm_unet=g, scheduler=s, safety_checker=None,
                                        feature_extractor=None, requires_safety_checker=False)
    p.set_progress_bar_config(disable=True)
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--pipes", type=int, default=2)
    ap.add_argument("--steps", type=int, default=50)
    a = ap.parse_args()
    P, b = a.pipes, a.batch // a.pipes
    pipes = [build(100 + 10 * i) for i in range(P)]
    g = torch.Generator().manual_seed(0)
    ins = [(torch.randn(b, 77, 768, generator=g).cuda(), torch.randn(b, 77, 768, generator=g).cuda(), torch.randn(b, 4, 64, 64, generator=g).cuda()) for _ in range(P)]
    streams = [torch.cuda.Stream() for _ in range(P)]

    def run(i):
        with torch.cuda.stream(streams[i]):
            pe, ne, lat = ins[i]
            pipes[i](prompt_embeds=pe, negative_prompt_embeds=ne, latents=lat, height=512, width=512, num_inference_steps=a.steps,
                     guidance_scale=7.5, output_type="latent")

    def round_():
        th = [threading.Thread(target=run, args=(i,)) for i in range(P)]
        [t.start() for t in th]; [t.join() for t in th]
        torch.cuda.synchronize()

    for i in range(P):  # graph capture must not race with another thread's launches: warm up serially
        run(i)
    torch.cuda.synchronize()
    round_()
    t0 = time.perf_counter(); round_(); round_(); dt = (time.perf_counter() - t0) / 2
    print(f"pipes={P} batch/pipe={b}: {dt*1e3:.1f} ms per {a.batch} images (loop only, no VAE) -> {a.batch/dt:.3f} img/s")


if __name__ == "__main__":
    main()

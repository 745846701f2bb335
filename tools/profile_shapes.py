#!/usr/bin/env python
"""Per-shape kernel timing of one SDR-UNet eval (batch 2B), one GM-UNet eval (batch B) and one VAE decode at the
bench configuration (HIP events on the launch stream).  Diagnostic tool: prints a table sorted by total time."""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gm-diffusion_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from gm_diffusion import hip_ops as ops, profiling  # noqa: E402
from gm_diffusion.components import AutoencoderKL, UNet2DConditionModel  # noqa: E402


class ShapeTimer(profiling.KernelTimer):
    pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--what", default="unet4,unet8,vae")
    ap.add_argument("--lead-cycles", type=float, default=3e7)
    a = ap.parse_args()
    dev = "cuda"
    h = a.res // 8
    # monkey-patch the ops to tag shapes
    orig_gemm, orig_conv, orig_attn, orig_gn = ops.gemm_nt, ops.conv3x3, ops.attention, ops.groupnorm_scale_shift
    tags = []

    def wrap(kind, fn, tagger):
        def f(*args, **kw):
            tm = profiling.active()
            n0 = len(tm.records) if tm else 0
            out = fn(*args, **kw)
            if tm and len(tm.records) > n0:
                k, fl, by, s, e = tm.records[-1]
                tm.records[-1] = (f"{kind} {tagger(*args, **kw)}", fl, by, s, e)
            return out
        return f

    ops.gemm_nt = wrap("gemm", orig_gemm, lambda a_, w, **kw: f"M={a_.shape[-2]} N={w.shape[-2]} K={a_.shape[-1]} b={(a_.shape[0] if a_.dim()==3 else (w.shape[0] if w.dim()==3 else 1))}")
    ops.conv3x3 = wrap("conv", orig_conv, lambda x, w, B, H, W, **kw: f"B={B} {H}x{W} Cin={x.shape[-1]} Cout={w.shape[0]} s={kw.get('stride',1)} up={int(kw.get('upsample',False))}")
    ops.attention = wrap("attn", orig_attn, lambda q, k, vt, heads, nk, scale, **kw: f"B={q.shape[0]} Nq={q.shape[1]} Nk={nk} d={vt.shape[1]//heads}")
    ops.groupnorm = wrap("gn", ops.groupnorm, lambda x, B, G, *a_, **kw: f"B={B} rows={x.numel() // (B * x.shape[-1])} C={x.shape[-1]} silu={int(kw.get('silu', False))}")
    ops.layernorm = wrap("ln", ops.layernorm, lambda x, *a_, **kw: f"rows={x.numel() // x.shape[-1]} C={x.shape[-1]}")
    ops.concat_channels = wrap("cat", ops.concat_channels, lambda a_, b_: f"rows={a_.numel() // a_.shape[-1]} C={a_.shape[-1]}+{b_.shape[-1]}")
    import gm_diffusion.components.unet_2d_condition as U
    import gm_diffusion.components.autoencoder_kl as V
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(0)
    timer = ShapeTimer()
    jobs = []
    if "unet4" in a.what:
        u4 = UNet2DConditionModel(in_channels=4).init_random(1).to(dev, dtype)
        x4 = torch.randn(2 * a.batch, 4, h, h, generator=g).to(dev)
        c4 = torch.randn(2 * a.batch, 77, 768, generator=g).to(dev)
        jobs.append(("unet4", lambda: u4(x4, 500, encoder_hidden_states=c4, return_dict=False)))
    if "unet8" in a.what:
        u8 = UNet2DConditionModel(in_channels=8).init_random(2).to(dev, dtype)
        x8 = torch.randn(a.batch, 8, h, h, generator=g).to(dev)
        c8 = torch.randn(a.batch, 77, 768, generator=g).to(dev)
        jobs.append(("unet8", lambda: u8(x8, 500, encoder_hidden_states=c8, return_dict=False)))
    if "vae" in a.what:
        vae = AutoencoderKL().init_random(3).to(dev, dtype)
        z = torch.randn(a.batch, 4, h, h, generator=g).to(dev)
        jobs.append(("vae", lambda: vae.decode_nhwc(z)))
    for name, fn in jobs:
        fn()
        torch.cuda.synchronize()
        timer.records.clear()
        profiling.set_timer(timer)
        ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(int(a.lead_cycles))  # let the host run ahead: otherwise short kernels also count launch latency
        ev0.record()
        for _ in range(a.reps):
            fn()
        ev1.record()
        profiling.set_timer(None)
        torch.cuda.synchronize()
        total = ev0.elapsed_time(ev1) / a.reps
        summ = timer.summary()
        ksum = sum(v["ms"] for v in summ.values()) / a.reps
        print(f"\n=== {name}: {total:.3f} ms per call; timed gemm/conv/attn kernels {ksum:.3f} ms ===")
        for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])[:90]:
            print(f"{k:62s} n={v['launches']//a.reps:3d} avg_us={v['avg_us']:9.1f} tot_ms={v['ms']/a.reps:8.3f} TF/s={v['tflops']:7.1f} GB/s={v['gbps']:7.0f}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""LayerNorm micro-benchmark over the UNet's token shapes (diagnostic; device-side lead so launches are back to back)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gm-diffusion_amd"))
import torch
from gm_diffusion import hip_ops as ops
g = torch.Generator().manual_seed(0)
for rows, C in [(32768, 320), (8192, 640), (2048, 1280), (16384, 320), (4096, 640), (1024, 1280), (512, 1280)]:
    x = torch.randn(rows, C, generator=g).bfloat16().cuda()
    ga, be = torch.ones(C).cuda(), torch.zeros(C).cuda()
    f = lambda: ops.layernorm(x, ga, be, 1e-5)
    f(); f(); torch.cuda.synchronize(); torch.cuda._sleep(int(2e7))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): f()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 200 * 1e3
    print(f"LN rows={rows} C={C}: {t:6.1f} us  {rows * C * 4 / t / 1e3:7.0f} GB/s (read+write)")

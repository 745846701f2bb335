#!/usr/bin/env python
"""Attention micro-benchmark over the bench workload's shapes (diagnostic). GMD_LIB_OVERRIDE selects a library build."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gm-diffusion_amd"))
import torch
from gm_diffusion import hip_ops as ops

SHAPES = [(8, 4096, 4096, 8, 40), (4, 4096, 4096, 8, 40), (8, 1024, 1024, 8, 80), (4, 1024, 1024, 8, 80), (8, 256, 256, 8, 160),
          (8, 4096, 77, 8, 40), (8, 1024, 77, 8, 80), (8, 64, 64, 8, 160)]
g = torch.Generator().manual_seed(0)
DT = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[os.environ.get("GMD_ONE_DTYPE", "bf16")]  # f32 = attention_split.hip
print("lib:", os.environ.get("GMD_LIB_OVERRIDE", "prod"), "dtype:", DT)
for B, Nq, Nk, H, D in SHAPES:
    C = H * D
    q = torch.randn(B, Nq, C, generator=g).to(DT).cuda(); k = torch.randn(B, Nk, C, generator=g).to(DT).cuda()
    nkp = (Nk + 7) // 8 * 8
    vt = torch.randn(B, C, nkp, generator=g).to(DT).cuda()
    f = lambda: ops.attention(q, k, vt, H, Nk, D ** -0.5)
    f(); f(); torch.cuda.synchronize(); torch.cuda._sleep(int(1e7))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    print(f"attn B={B} Nq={Nq} Nk={Nk} H={H} d={D}: {us:8.1f} us  {4.0 * B * H * Nq * Nk * D / us / 1e6:7.1f} TF/s")

#!/usr/bin/env python
"""V^T projection (operand-swapped, batched) micro-benchmark: V^T[b] = W_v . x_b^T (diagnostic)."""
import os, sys
os.environ.setdefault("GMD_TUNING", "1")  # kernel-plan overrides are a debug facility (include/gmd_hip.h)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gm-diffusion_amd"))
import torch
from gm_diffusion import hip_ops as ops
g = torch.Generator().manual_seed(0)
print("variant:", os.environ.get("GMD_GEMM_FORCE", "heuristic"))
for B, N, C in ((8, 4096, 320), (8, 1024, 640), (8, 256, 1280), (8, 64, 1280), (4, 4096, 320), (4, 1024, 640)):
    x = torch.randn(B, N, C, generator=g).bfloat16().cuda(); w = (torch.randn(C, C, generator=g) * C ** -0.5).bfloat16().cuda()
    ld = (N + 7) // 8 * 8
    f = lambda: ops.gemm_nt(w, x, ldc=ld)
    f(); f(); torch.cuda.synchronize(); torch.cuda._sleep(int(1e7))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 100 * 1e3
    print(f"vt B={B} N={N} C={C}: {us:8.1f} us  {2.0 * B * N * C * C / us / 1e6:7.1f} TF/s")

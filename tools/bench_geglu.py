#!/usr/bin/env python
"""ff1 (GEGLU-epilogue) GEMM micro-benchmark over the three transformer widths (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gm-diffusion_amd"))
import torch
from gm_diffusion import hip_ops as ops
g = torch.Generator().manual_seed(0)
print("lib:", os.environ.get("GMD_LIB_OVERRIDE", "prod"))
for M, C in ((32768, 320), (8192, 640), (2048, 1280), (16384, 320), (4096, 640), (1024, 1280)):
    x = torch.randn(M, C, generator=g).bfloat16().cuda(); w = (torch.randn(8 * C, C, generator=g) * C ** -0.5).bfloat16().cuda(); b = torch.randn(8 * C, generator=g).cuda()
    f = lambda: ops.gemm_nt(x, w, bias=b, act=ops.ACT_GEGLU)
    f(); f(); torch.cuda.synchronize(); torch.cuda._sleep(int(1e7))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f"ff1 GEGLU M={M} C={C}: {us:8.1f} us  {2.0 * M * 8 * C * C / us / 1e6:7.1f} TF/s")

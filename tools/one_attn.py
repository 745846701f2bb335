#!/usr/bin/env python
"""Run one attention shape a few times (target for rocprofv3 --pmc)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gm-diffusion_amd"))
import torch
from gm_diffusion import hip_ops as ops
B, Nq, Nk, H, D = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (8, 4096, 4096, 8, 40))]
g = torch.Generator().manual_seed(0)
C = H * D
q = torch.randn(B, Nq, C, generator=g).bfloat16().cuda(); k = torch.randn(B, Nk, C, generator=g).bfloat16().cuda()
vt = torch.randn(B, C, (Nk + 7) // 8 * 8, generator=g).bfloat16().cuda()
for _ in range(4):
    ops.attention(q, k, vt, H, Nk, D ** -0.5)
torch.cuda.synchronize()
print("done")

#!/usr/bin/env python
"""Run the level-0 fused GEGLU feed-forward (gmd_ff_geglu_fused, C = 320) a few times on M rows (target for rocprofv3 --pmc)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gm-diffusion_amd"))
import torch
from gm_diffusion import hip_ops as ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
C = 320
g = torch.Generator().manual_seed(0)
x = torch.randn(M, C, generator=g).bfloat16().cuda(); r = torch.randn(M, C, generator=g).bfloat16().cuda()
w1 = (torch.randn(8 * C, C, generator=g) * 0.05).bfloat16().cuda(); b1 = torch.randn(8 * C, generator=g).cuda()
w2 = (torch.randn(C, 4 * C, generator=g) * 0.03).bfloat16().cuda(); b2 = torch.randn(C, generator=g).cuda()
for _ in range(5):
    ops.ff_geglu_fused(x, w1, b1, w2, b2, r)
torch.cuda.synchronize()
print("done")

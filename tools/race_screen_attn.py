#!/usr/bin/env python
"""Race screen for the LDS-DMA attention kernel (d = 40): the same inputs launched many times must give bit-identical outputs,
also while a second stream keeps the chip busy with other kernels (diagnostic; the kernel's ordering argument is in attention.hip)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gm-diffusion_amd"))
import torch
from gm_diffusion import hip_ops as ops

g = torch.Generator().manual_seed(3)
side = torch.cuda.Stream()
xa = torch.randn(16384, 320, generator=g).bfloat16().cuda()
wa = torch.randn(320, 320, generator=g).bfloat16().cuda()
bad = 0
for dtype in (torch.bfloat16, torch.float16):
    for B, Nq, Nk, H in [(8, 4096, 4096, 8), (8, 4096, 77, 8), (3, 1000, 130, 8), (2, 200, 64, 4), (1, 16384, 16384, 2)]:
        D, C = 40, H * 40
        q = torch.randn(B, Nq, C, generator=g).to(dtype).cuda()
        k = torch.randn(B, Nk, C, generator=g).to(dtype).cuda()
        vt = torch.randn(B, C, (Nk + 7) // 8 * 8, generator=g).to(dtype).cuda()
        ref = ops.attention(q, k, vt, H, Nk, D ** -0.5).clone()
        n = 20 if Nk >= 16384 else 150
        diff = 0
        for it in range(n):
            if it % 3 == 0:
                with torch.cuda.stream(side):
                    for _ in range(4):
                        ops.gemm_nt(xa.to(torch.bfloat16), wa)
            out = ops.attention(q, k, vt, H, Nk, D ** -0.5)
            diff += int(not torch.equal(out, ref))
        torch.cuda.synchronize()
        print(f"{str(dtype):16s} B={B} Nq={Nq} Nk={Nk} H={H}: {diff} of {n} launches differ")
        bad += diff
print("RACE SCREEN", "FAILED" if bad else "clean")
sys.exit(1 if bad else 0)

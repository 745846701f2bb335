#!/usr/bin/env python
"""Is a kernel held back by the clock the chip keeps under load?  The same launches on random and on all-zero operands: identical
instruction streams and memory traffic, but zero operands switch far fewer transistors in the matrix pipes, so the chip holds a higher
clock (MI355X_MICROARCH.md, DVFS give-back: +15...21 % on a tuned GEMM).  A kernel that speeds up on zeros by that much is bound by
power / clock, not by its schedule."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gm-diffusion_amd")):
    sys.path.insert(0, p)
import torch
from gm_diffusion import hip_ops as ops


def timeit(fn, reps=60):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    torch.cuda._sleep(int(1e7))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


g = torch.Generator().manual_seed(0)
for B, H, ci, co in [(8, 64, 640, 320), (8, 64, 320, 320), (8, 32, 640, 640), (8, 32, 1280, 640), (8, 16, 1280, 1280)]:
    row = []
    for zero in (False, True):
        x = torch.randn(B, H * H, ci, generator=g).bfloat16().cuda()
        w = (torch.randn(co, 9 * ci, generator=g) * 0.02).bfloat16().cuda()
        if zero:
            x.zero_(); w.zero_()
        b = torch.zeros(co).cuda()
        row.append(timeit(lambda: ops.conv3x3(x, w, B, H, H, bias=b)))
    fl = 2.0 * B * H * H * co * 9 * ci
    print(f"conv B={B} {H}x{H} {ci}->{co}: random {row[0]:7.1f} us ({fl / row[0] / 1e6:5.0f} TF/s)   zeros {row[1]:7.1f} us ({fl / row[1] / 1e6:5.0f} TF/s)   x{row[0] / row[1]:.2f}", flush=True)
for M, N, K in [(32768, 320, 1280), (8192, 640, 2560), (2048, 1280, 5120), (8192, 5120, 640)]:
    row = []
    for zero in (False, True):
        a = torch.randn(M, K, generator=g).bfloat16().cuda()
        w = (torch.randn(N, K, generator=g) * 0.02).bfloat16().cuda()
        if zero:
            a.zero_(); w.zero_()
        row.append(timeit(lambda: ops.gemm_nt(a, w)))
    fl = 2.0 * M * N * K
    print(f"gemm M={M} N={N} K={K}: random {row[0]:7.1f} us ({fl / row[0] / 1e6:5.0f} TF/s)   zeros {row[1]:7.1f} us ({fl / row[1] / 1e6:5.0f} TF/s)   x{row[0] / row[1]:.2f}", flush=True)

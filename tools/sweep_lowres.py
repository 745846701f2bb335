#!/usr/bin/env python
"""Sweep of tile / split-K plans on the low-resolution convolutions (16x16 / 8x8 UNet levels) against the heuristic of make_plan
(csrc/gemm.hip): is there a plan the heuristic misses?  Uses the debug-only plan override (GMD_TUNING=1)."""
import os, sys
os.environ.setdefault("GMD_TUNING", "1")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gm-diffusion_amd"))
import torch
from gm_diffusion import hip_ops as ops
from gm_diffusion._native import lib


def timeit(fn, reps=50):
    fn(); fn(); torch.cuda.synchronize()
    torch.cuda._sleep(int(1e7))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


g = torch.Generator().manual_seed(0)
SHAPES = [(8, 16, 1280, 1280), (4, 16, 1280, 1280), (8, 16, 2560, 1280), (8, 8, 1280, 1280), (4, 8, 1280, 1280), (8, 8, 2560, 1280), (8, 16, 640, 1280),
          (8, 32, 1280, 640), (4, 32, 640, 640)]
PLANS = [(128, 160, 0, k) for k in (1, 2, 3, 4, 6, 8, 12, 16)] + [(128, 128, 0, k) for k in (2, 4, 8)] + [(64, 64, 9, k) for k in (1, 2, 3, 4, 6, 8)]
for B, H, ci, co in SHAPES:
    x = torch.randn(B, H * H, ci, generator=g).bfloat16().cuda()
    w = (torch.randn(co, 9 * ci, generator=g) * 0.02).bfloat16().cuda()
    b = torch.randn(co, generator=g).cuda()
    fn = lambda: ops.conv3x3(x, w, B, H, H, bias=b)
    lib().gmd_gemm_plan_override(0, 0, 0, 0)
    timeit(fn, 20)  # (the first timing after allocating the operands reads high)
    base = timeit(fn)
    res = []
    for bm, bn, pf, ks in PLANS:
        if lib().gmd_gemm_plan_override(bm, bn, pf, ks) != 0:
            continue
        try:
            res.append((timeit(fn), bm, bn, ks))
        except Exception as e:  # plan refused for this shape
            pass
    lib().gmd_gemm_plan_override(0, 0, 0, 0)
    res.sort()
    fl = 2.0 * B * H * H * co * 9 * ci
    best = ", ".join(f"{bm}x{bn} ks={ks}: {t:.1f}" for t, bm, bn, ks in res[:4])
    print(f"conv B={B} {H}x{H} {ci}->{co}: heuristic {base:6.1f} us ({fl / base / 1e6:5.0f} TF/s)   best forced: {best}", flush=True)

#!/usr/bin/env python
"""Tile / K-slice sweep of the loader/consumer kernel (gemm_lc_kernel, plan code 244) on the launches that cannot put one 128-row tile on
every CU -- the 16x16 / 8x8 convolutions and the deep-K linears of the lower UNet levels -- against the ring kernels' plans and the
heuristic's choice.  Debug-only plan override (GMD_TUNING=1)."""
import os, sys
os.environ.setdefault("GMD_TUNING", "1")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gm-diffusion_amd"))
import torch
from gm_diffusion import hip_ops as ops
from gm_diffusion._native import lib


def timeit(fn, reps=40):
    fn(); fn(); torch.cuda.synchronize()
    torch.cuda._sleep(int(1e7))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


g = torch.Generator().manual_seed(0)
CONVS = [(8, 16, 1280, 1280), (4, 16, 1280, 1280), (8, 16, 2560, 1280), (4, 16, 2560, 1280), (8, 16, 640, 1280), (4, 16, 640, 1280), (8, 8, 1280, 1280),
         (4, 8, 1280, 1280), (8, 8, 2560, 1280), (4, 8, 2560, 1280), (4, 32, 1280, 640), (4, 32, 320, 640), (8, 32, 320, 640)]
GEMMS = [(1024, 1280, 5120), (512, 1280, 5120), (256, 1280, 5120), (1024, 1280, 1280), (512, 1280, 1280), (512, 2560, 1280), (1024, 2560, 1280)]
PLANS = ([(128, 160, 9, k) for k in (1, 2, 4, 8, 16)] + [(64, 64, 9, k) for k in (1, 2, 4)] +
         [(128, 160, 244, k) for k in (1, 2, 3, 4, 6, 8)] + [(64, 160, 244, k) for k in (1, 2, 3, 4, 6, 8)])


def sweep(name, fn, flops, info):
    lib().gmd_gemm_plan_override(0, 0, 0, 0)
    timeit(fn, 10)
    base = timeit(fn)
    res = []
    for bm, bn, pf, ks in PLANS:
        if lib().gmd_gemm_plan_override(bm, bn, pf, ks) != 0:
            continue
        try:
            res.append((timeit(fn), bm, bn, pf, ks))
        except Exception:
            pass
    lib().gmd_gemm_plan_override(0, 0, 0, 0)
    res.sort()
    best = ", ".join(f"{'lc' if pf == 244 else 'ring'} {bm}x{bn} ks={ks}: {t:.1f}" for t, bm, bn, pf, ks in res[:5])
    print(f"{name}: heuristic {info} {base:6.1f} us ({flops / base / 1e6:5.0f} TF/s)   best forced: {best}", flush=True)


for B, H, ci, co in CONVS:
    x = torch.randn(B, H * H, ci, generator=g).bfloat16().cuda()
    w = (torch.randn(co, 9 * ci, generator=g) * 0.02).bfloat16().cuda()
    b = torch.randn(co, generator=g).cuda()
    sweep(f"conv B={B} {H}x{H} {ci}->{co}", lambda: ops.conv3x3(x, w, B, H, H, bias=b), 2.0 * B * H * H * co * 9 * ci,
          ops.gemm_plan_info(torch.bfloat16, B * H * H, co, 9 * ci))
for M, N, K in GEMMS:
    a = torch.randn(M, K, generator=g).bfloat16().cuda()
    w = (torch.randn(N, K, generator=g) * 0.02).bfloat16().cuda()
    b = torch.randn(N, generator=g).cuda()
    r = torch.randn(M, N, generator=g).bfloat16().cuda()
    sweep(f"gemm M={M} N={N} K={K}", lambda: ops.gemm_nt(a, w, bias=b, residual=r), 2.0 * M * N * K, ops.gemm_plan_info(torch.bfloat16, M, N, K))

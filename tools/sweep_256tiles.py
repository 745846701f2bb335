#!/usr/bin/env python
"""Plans for the launches that put exactly one 4-wave workgroup on each CU (224..256 tiles of 128x160: the GM UNet's level-0
convolutions, the SDR UNet's 32x32 level): every instantiated tile / ring / split-K variant against the heuristic."""
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
PLANS = [(128, 160, 0, 1), (128, 160, 0, 2), (128, 160, 0, 3), (128, 160, 123, 1), (128, 160, 124, 1), (128, 160, 143, 1), (128, 128, 0, 1), (128, 128, 0, 2),
         (64, 64, 9, 1), (64, 64, 103, 1), (64, 64, 104, 1), (64, 128, 103, 1), (64, 128, 104, 1)]


def sweep(name, fn, flops):
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
    best = ", ".join(f"{bm}x{bn} pf{pf} ks{ks}: {t:.1f}" for t, bm, bn, pf, ks in res[:5])
    print(f"{name}: heuristic {base:6.1f} us ({flops / base / 1e6:5.0f} TF/s)   forced: {best}", flush=True)


for B, H, ci, co in [(4, 64, 320, 320), (4, 64, 640, 320), (4, 64, 960, 320), (8, 32, 640, 640), (8, 32, 1280, 640), (8, 32, 1920, 640), (8, 32, 320, 640)]:
    x = torch.randn(B, H * H, ci, generator=g).bfloat16().cuda()
    w = (torch.randn(co, 9 * ci, generator=g) * 0.02).bfloat16().cuda()
    b = torch.randn(co, generator=g).cuda()
    sweep(f"conv B={B} {H}x{H} {ci}->{co}", lambda: ops.conv3x3(x, w, B, H, H, bias=b), 2.0 * B * H * H * co * 9 * ci)
for M, N, K in [(8192, 640, 640), (8192, 640, 2560), (16384, 320, 1280), (8192, 1280, 640)]:
    xs = [torch.randn(M, K, generator=g).bfloat16().cuda() for _ in range(3)]
    rs = [torch.randn(M, N, generator=g).bfloat16().cuda() for _ in range(3)]
    w = (torch.randn(N, K, generator=g) * 0.02).bfloat16().cuda()
    b = torch.randn(N, generator=g).cuda()
    i = [0]

    def fn():
        i[0] = (i[0] + 1) % 3
        return ops.gemm_nt(xs[i[0]], w, bias=b, residual=rs[i[0]])

    sweep(f"gemm M={M} N={N} K={K}", fn, 2.0 * M * N * K)

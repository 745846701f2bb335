#!/usr/bin/env python
"""Correctness + speed of GEMM kernel variants (gmd_gemm_plan_override) against the default kernel, in one process."""
import os, sys
os.environ.setdefault("GMD_TUNING", "1")  # kernel-plan overrides are a debug facility (include/gmd_hip.h)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gm-diffusion_amd"))
import torch
from gm_diffusion import hip_ops as ops
from gm_diffusion._native import lib


def force(v=None):
    lib().gmd_gemm_plan_override(*([int(x) for x in v.split(",")] if v else [0, 0, 0, 0]))

variants = sys.argv[1:] or ["128,160,123,1", "128,160,124,1", "256,160,143,1", "128,160,122,1"]
g = torch.Generator().manual_seed(0)


def timeit(fn, reps=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


cases = []
for B, H, W, ci, co, kw in [(8, 64, 64, 320, 320, {}), (8, 64, 64, 640, 320, {}), (8, 32, 32, 640, 640, {}), (2, 33, 17, 128, 320, {}),
                            (8, 32, 32, 640, 640, dict(upsample=True)), (8, 64, 64, 320, 320, dict(stride=2)), (8, 16, 16, 1280, 1280, {}),
                            (4, 128, 128, 512, 512, {}), (4, 256, 256, 256, 256, {}), (4, 64, 64, 320, 320, {}), (4, 32, 32, 640, 640, {}),
                            (8, 32, 32, 1280, 640, {}), (4, 16, 16, 1280, 1280, {}), (8, 8, 8, 1280, 1280, {})]:
    x = torch.randn(B, H * W, ci, generator=g).bfloat16().cuda()
    w = (torch.randn(co, 9 * ci, generator=g) * 0.02).bfloat16().cuda()
    b = torch.randn(co, generator=g).cuda()
    cases.append((f"conv B={B} {H}x{W} {ci}->{co} {kw}", 2.0 * B * H * W * co * 9 * ci / (kw.get('stride', 1) ** 2) * (4 if kw.get('upsample') else 1),
                  lambda x=x, w=w, b=b, B=B, H=H, W=W, kw=kw: ops.conv3x3(x, w, B, H, W, bias=b, **kw)[0]))
for M, N, K in [(32768, 320, 320), (32768, 2560, 320), (32768, 320, 1280), (8192, 640, 640), (8192, 5120, 640), (8192, 640, 2560), (2048, 1280, 1280),
                (2048, 1280, 5120), (1000, 320, 192), (16384, 320, 320), (4096, 640, 640), (1024, 1280, 1280), (8192, 1280, 640), (2048, 2560, 1280),
                (4096, 640, 2560), (512, 1280, 1280), (16384, 320, 1280)]:
    a = torch.randn(M, K, generator=g).bfloat16().cuda()
    w = (torch.randn(N, K, generator=g) * 0.02).bfloat16().cuda()
    b = torch.randn(N, generator=g).cuda()
    r = torch.randn(M, N, generator=g).bfloat16().cuda()
    cases.append((f"gemm M={M} N={N} K={K}", 2.0 * M * N * K, lambda a=a, w=w, b=b, r=r: ops.gemm_nt(a, w, bias=b, residual=r)))

for M, N, K in [(32768, 2560, 320), (8192, 5120, 640), (2048, 10240, 1280)]:
    a = torch.randn(M, K, generator=g).bfloat16().cuda()
    w = (torch.randn(N, K, generator=g) * 0.02).bfloat16().cuda()
    b = torch.randn(N, generator=g).cuda()
    cases.append((f"geglu M={M} N={N} K={K}", 2.0 * M * N * K, lambda a=a, w=w, b=b: ops.gemm_nt(a, w, bias=b, act=ops.ACT_GEGLU)))
x = torch.randn(8, 64 * 64, 320, generator=g).bfloat16().cuda()
w3 = (torch.randn(320, 9 * 320, generator=g) * 0.02).bfloat16().cuda()
tb = torch.randn(8, 320, generator=g).cuda()
rs = torch.randn(8, 64 * 64, 320, generator=g).bfloat16().cuda()
cases.append(("conv B=8 64x64 320->320 rowbias+res", 2.0 * 32768 * 320 * 2880, lambda: ops.conv3x3(x, w3, 8, 64, 64, bias=tb[0].contiguous(), rowbias=tb, residual=rs)[0]))

print(f"{'case':52s} {'default':>16s} " + " ".join(f"{v:>22s}" for v in variants))
for name, fl, fn in cases:
    force()
    ref = fn().float()
    t0 = timeit(fn)
    row = f"{name:52s} {t0:7.1f}us {fl / t0 / 1e6:6.0f}TF"
    for v in variants:
        bn_ok = True
        force(v)
        try:
            out = fn().float()
            err = float((out - ref).abs().max())
            t = timeit(fn)
            row += f" {t:7.1f}us {fl / t / 1e6:6.0f}TF e={err:.0e}"
        except Exception as e:
            row += f" {'n/a':>22s}"
    print(row, flush=True)
force()

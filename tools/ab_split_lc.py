#!/usr/bin/env python
"""float32 matrix-core path: the loader / converter kernel (gemm_split_lc_kernel: activation rows split once, in LDS, by the loader
waves) against the round-3 kernel (every wave splits its fragments in registers) on the pipeline's float32 shapes, pre-split weights:
time per launch, and bit-identity of the results (same products in the same order)."""
import os, sys
os.environ.setdefault("GMD_TUNING", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gm-diffusion_amd")):
    sys.path.insert(0, p)
import torch
from gm_diffusion import hip_ops as ops
from gm_diffusion._native import lib

ops.set_f32_mode("split")


def timeit(fn, reps=30):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    torch.cuda._sleep(int(2e7))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def ab(name, fl, fn):
    lib().gmd_gemm_plan_override(0, 0, 9, 0)
    y0 = fn(); t0 = timeit(fn)
    lib().gmd_gemm_plan_override(0, 0, 0, 0)   # the split planner's own choice
    y1 = fn(); t1 = timeit(fn)
    lib().gmd_gemm_plan_override(0, 0, 244, 0)  # loader / converter kernel wherever instantiated
    y2 = fn(); t2 = timeit(fn)
    lib().gmd_gemm_plan_override(0, 0, 0, 0)
    print(f"{name:44s} ring {t0:7.1f} us ({fl / t0 / 1e6:4.0f} TF/s)   default {t1:7.1f}   forced lc {t2:7.1f} ({fl / t2 / 1e6:4.0f} TF/s)  x{t0 / t2:.2f}   "
          f"bit-identical: {bool(torch.equal(y0, y2))}", flush=True)


g = torch.Generator().manual_seed(0)
for B, H, ci, co in [(8, 64, 320, 320), (8, 64, 640, 320), (8, 32, 640, 640), (8, 32, 1280, 640), (4, 64, 320, 320), (4, 32, 640, 640), (8, 16, 1280, 1280)]:
    x = torch.randn(B, H * H, ci, generator=g).cuda()
    w = ops.split_weights((torch.randn(co, 9 * ci, generator=g) * 0.02).cuda())
    b = torch.randn(co, generator=g).cuda()
    ab(f"conv B={B} {H}x{H} {ci}->{co}", 2.0 * B * H * H * co * 9 * ci, lambda: ops.conv3x3(x, w, B, H, H, bias=b)[0])
for M, N, K, mode in [(32768, 320, 320, "res"), (32768, 640, 320, "bias"), (32768, 320, 1280, "res"), (8192, 640, 640, "res"), (8192, 640, 2560, "res"),
                      (32768, 2560, 320, "geglu"), (8192, 5120, 640, "geglu"), (2048, 10240, 1280, "geglu"), (2048, 2560, 1280, "bias"), (16384, 320, 320, "res")]:
    a = torch.randn(M, K, generator=g).cuda()
    w = ops.split_weights((torch.randn(N, K, generator=g) * 0.02).cuda())
    b = torch.randn(N, generator=g).cuda()
    kw = dict(bias=b)
    if mode == "res": kw["residual"] = torch.randn(M, N, generator=g).cuda()
    if mode == "geglu": kw["act"] = ops.ACT_GEGLU
    ab(f"gemm M={M} N={N} K={K} {mode}", 2.0 * M * N * K, lambda: ops.gemm_nt(a, w, **kw))

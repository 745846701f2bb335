#!/usr/bin/env python
"""float32-on-matrix-cores (three float16 products) against the exact float32 FMA kernel and the float16 kernel, on the
GEMM / conv3x3 shapes of the bench workload: time per launch, effective TFLOP/s and error against float64 (small shapes).
Usage: bench_split.py [--gemm-only|--conv-only] [--err]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gm-diffusion_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from gm_diffusion import hip_ops as ops  # noqa: E402

DEV = "cuda"
CONVS = [(8, 64, 64, 320, 320), (8, 64, 64, 640, 320), (8, 32, 32, 640, 640), (8, 32, 32, 1280, 640), (8, 16, 16, 1280, 1280),
         (8, 16, 16, 2560, 1280), (8, 8, 8, 1280, 1280), (8, 8, 8, 2560, 1280), (4, 64, 64, 320, 320), (4, 128, 128, 512, 512),
         (4, 512, 512, 128, 128)]
GEMMS = [(32768, 320, 320), (32768, 2560, 320), (32768, 320, 1280), (8192, 640, 640), (8192, 5120, 640), (8192, 640, 2560),
         (2048, 1280, 1280), (2048, 10240, 1280), (2048, 1280, 5120), (512, 1280, 1280), (16384, 320, 320), (8, 1280, 1280)]


def timeit(fn, reps=30):
    fn(); fn()
    torch.cuda.synchronize()
    torch.cuda._sleep(int(2e7))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


def rel(a, b):
    return float((a.double() - b).norm() / b.norm())


def main():
    g = torch.Generator().manual_seed(0)
    convs, gemms = CONVS, GEMMS
    if "--gemm-only" in sys.argv:
        convs = []
    if "--conv-only" in sys.argv:
        gemms = []
    want_err = "--err" in sys.argv
    tot = dict(exact=0.0, split=0.0, presplit=0.0, f16=0.0)
    print(f"{'shape':38s} {'exact us':>9s} {'split us':>9s} {'presplit':>9s} {'f16 us':>8s}   TF/s(presplit)  x vs f16")
    for B, H, W, ci, co in convs:
        x = torch.randn(B, H * W, ci, generator=g).to(DEV)
        w = (torch.randn(co, 9 * ci, generator=g) * 0.02).to(DEV)
        b = torch.randn(co, generator=g).to(DEV)
        ops.set_f32_mode("split")
        ws = ops.split_weights(w)
        t_s = timeit(lambda: ops.conv3x3(x, w, B, H, W, bias=b))
        t_p = timeit(lambda: ops.conv3x3(x, ws, B, H, W, bias=b))
        ops.set_f32_mode("exact")
        t_e = timeit(lambda: ops.conv3x3(x, w, B, H, W, bias=b), reps=3) if B * H * W * co * ci < 8 * 64 * 64 * 640 * 320 + 1 else float("nan")
        ops.set_f32_mode("split")
        xh, wh = x.half(), w.half()
        t_h = timeit(lambda: ops.conv3x3(xh, wh, B, H, W, bias=b))
        fl = 2.0 * B * H * W * co * 9 * ci
        for k, v in (("exact", t_e), ("split", t_s), ("presplit", t_p), ("f16", t_h)):
            tot[k] += v
        print(f"conv B={B} {H}x{W} {ci}->{co}".ljust(38) + f" {t_e:9.1f} {t_s:9.1f} {t_p:9.1f} {t_h:8.1f}   {fl / t_p / 1e6:7.1f}        {t_p / t_h:5.2f}")
    for M, N, K in gemms:
        a = torch.randn(M, K, generator=g).to(DEV)
        w = (torch.randn(N, K, generator=g) * 0.02).to(DEV)
        b = torch.randn(N, generator=g).to(DEV)
        ops.set_f32_mode("split")
        ws = ops.split_weights(w)
        t_s = timeit(lambda: ops.gemm_nt(a, w, bias=b))
        t_p = timeit(lambda: ops.gemm_nt(a, ws, bias=b))
        ops.set_f32_mode("exact")
        t_e = timeit(lambda: ops.gemm_nt(a, w, bias=b), reps=3)
        if want_err and M * N <= 2048 * 10240:
            ref = a.double() @ w.double().t() + b.double()
            e_exact = rel(ops.gemm_nt(a, w, bias=b), ref)
            ops.set_f32_mode("split")
            e_split = rel(ops.gemm_nt(a, ws, bias=b), ref)
            e_half = rel(ops.gemm_nt(a.half(), w.half(), bias=b, out_dtype=torch.float32), ref)
            errs = f"  err exact {e_exact:.1e} split {e_split:.1e} f16 {e_half:.1e}"
        else:
            errs = ""
        ops.set_f32_mode("split")
        ah, wh = a.half(), w.half()
        t_h = timeit(lambda: ops.gemm_nt(ah, wh, bias=b))
        for k, v in (("exact", t_e), ("split", t_s), ("presplit", t_p), ("f16", t_h)):
            tot[k] += v
        print(f"gemm M={M} N={N} K={K}".ljust(38) + f" {t_e:9.1f} {t_s:9.1f} {t_p:9.1f} {t_h:8.1f}   {2.0 * M * N * K / t_p / 1e6:7.1f}        {t_p / t_h:5.2f}{errs}")
    print("sum us:", {k: round(v, 1) for k, v in tot.items()})


if __name__ == "__main__":
    main()

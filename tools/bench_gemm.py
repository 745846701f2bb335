#!/usr/bin/env python
"""GEMM / conv3x3 micro-benchmark over the shapes of the bench workload (diagnostic tool).
Set GMD_GEMM_FORCE="bm,bn,pf,ksplit" to pin a kernel variant (0 = heuristic)."""
import os
import sys
os.environ.setdefault("GMD_TUNING", "1")  # kernel-plan overrides are a debug facility (include/gmd_hip.h)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gm-diffusion_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from gm_diffusion import hip_ops as ops  # noqa: E402

DEV = "cuda"
CONVS = [  # B, H, W, Cin, Cout
    (8, 64, 64, 320, 320), (8, 64, 64, 640, 320), (8, 32, 32, 640, 640), (8, 32, 32, 1280, 640), (8, 16, 16, 1280, 1280),
    (8, 16, 16, 2560, 1280), (8, 8, 8, 1280, 1280), (8, 8, 8, 2560, 1280), (4, 8, 8, 1280, 1280), (4, 16, 16, 1280, 1280),
    (4, 32, 32, 640, 640), (4, 64, 64, 320, 320), (4, 128, 128, 512, 512), (4, 256, 256, 256, 256), (4, 512, 512, 128, 128),
]
GEMMS = [  # M, N, K
    (32768, 320, 320), (32768, 2560, 320), (32768, 320, 1280), (8192, 640, 640), (8192, 5120, 640), (8192, 640, 2560),
    (2048, 1280, 1280), (2048, 10240, 1280), (2048, 1280, 5120), (512, 1280, 1280), (16384, 320, 320), (8, 1280, 1280),
    (4096, 640, 640), (1024, 1280, 1280), (32768, 640, 320), (8192, 1280, 640), (16384, 320, 1280), (4096, 640, 2560),
]


def timeit(fn, reps=100):
    fn(); fn()
    torch.cuda.synchronize()
    torch.cuda._sleep(int(1e7))  # ~0.1 s device-side lead so the host is ahead of the GPU (launch latency not counted)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


def main():
    g = torch.Generator().manual_seed(0)
    global CONVS, GEMMS
    if "--gemm-only" in sys.argv:
        CONVS = []
    if "--conv-only" in sys.argv:
        GEMMS = []
    print("variant:", os.environ.get("GMD_GEMM_FORCE", "heuristic"))
    tot = 0.0
    for B, H, W, ci, co in CONVS:
        x = torch.randn(B, H * W, ci, generator=g).bfloat16().to(DEV)
        w = (torch.randn(co, 9 * ci, generator=g) * 0.02).bfloat16().to(DEV)
        b = torch.randn(co, generator=g).to(DEV)
        us = timeit(lambda: ops.conv3x3(x, w, B, H, W, bias=b))
        fl = 2.0 * B * H * W * co * 9 * ci
        tot += us
        print(f"conv B={B} {H}x{W} {ci}->{co}: {us:9.1f} us  {fl / us / 1e6:7.1f} TF/s")
    for M, N, K in GEMMS:
        a = torch.randn(M, K, generator=g).bfloat16().to(DEV)
        w = (torch.randn(N, K, generator=g) * 0.02).bfloat16().to(DEV)
        b = torch.randn(N, generator=g).to(DEV)
        us = timeit(lambda: ops.gemm_nt(a, w, bias=b))
        tot += us
        print(f"gemm M={M} N={N} K={K}: {us:9.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s")
    print(f"sum {tot:.1f} us")


if __name__ == "__main__":
    main()

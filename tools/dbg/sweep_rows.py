"""Plan sweep on the yardstick rows still behind the vendor library (same timing method as tools/vs_library_gemm.py)."""
import os, sys
os.environ.setdefault("GMD_TUNING", "1")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gm-diffusion_amd"))
import torch
from gm_diffusion import hip_ops as ops
from gm_diffusion._native import lib
def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); torch.cuda._sleep(int(1e7))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
g = torch.Generator().manual_seed(0)
PLANS = [(256, 160, 283, 1), (256, 128, 283, 1), (128, 160, 244, 1), (128, 128, 244, 1), (64, 160, 244, 1), (64, 128, 244, 1), (128, 160, 0, 1), (128, 128, 0, 1), (128, 160, 123, 1),
         (64, 64, 0, 1), (64, 64, 9, 1), (64, 128, 103, 1), (64, 160, 244, 2), (128, 160, 244, 2), (256, 160, 283, 2), (256, 128, 283, 2), (64, 64, 0, 2)]
for M, N, K in [(1024, 1280, 1280), (2048, 640, 640), (4096, 512, 512), (16384, 128, 128), (1536, 1280, 1280), (1280, 1280, 1280)]:
    xs = [torch.randn(M, K, generator=g).bfloat16().cuda() for _ in range(3)]
    w = (torch.randn(N, K, generator=g) * 0.02).bfloat16().cuda()
    b32 = torch.randn(N, generator=g).cuda()
    i = [0]
    def mine():
        i[0] = (i[0] + 1) % 3
        return ops.gemm_nt(xs[i[0]], w, bias=b32)
    lib().gmd_gemm_plan_override(0, 0, 0, 0)
    timeit(mine, 20)
    base = timeit(mine)
    res = []
    for bm, bn, pf, ks in PLANS:
        if lib().gmd_gemm_plan_override(bm, bn, pf, ks) != 0:
            continue
        try:
            res.append((timeit(mine), bm, bn, pf, ks))
        except Exception as e:
            pass
    lib().gmd_gemm_plan_override(0, 0, 0, 0)
    res.sort()
    print(f"M={M} N={N} K={K}: default {ops.gemm_plan_info(torch.bfloat16, M, N, K)} {base:.1f} us | " + "  ".join(f"{t:.1f}:{bm}x{bn}/{pf}/{ks}" for t, bm, bn, pf, ks in res[:8]), flush=True)

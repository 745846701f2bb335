#!/usr/bin/env python
"""GroupNorm fed by producer statistics (gn_apply_cs_kernel) and the two-launch split path on the level-0 / level-1 shapes:
device time per launch.  GMD_LIB_OVERRIDE selects a library build."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gm-diffusion_amd"))
import torch
from gm_diffusion import hip_ops as ops

def timeit(fn, reps=100):
    fn(); fn(); torch.cuda.synchronize()
    torch.cuda._sleep(int(2e7))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

g = torch.Generator().manual_seed(0)
print("lib:", os.environ.get("GMD_LIB_OVERRIDE", "prod"))
for B, HW, C in [(8, 4096, 320), (4, 4096, 320), (8, 1024, 640), (8, 4096, 640)]:
    x = torch.randn(B * HW, C, generator=g).bfloat16().cuda()
    w = (torch.randn(C, C, generator=g) * 0.05).bfloat16().cuda()
    ga, be = torch.randn(C, generator=g).cuda(), torch.randn(C, generator=g).cuda()
    y = ops.gemm_nt(x, w, colstats=True)
    has = getattr(y, "_colstats", None) is not None
    yv = ops.carry_colstats(y.view(B, HW, C), y)
    t_cs = timeit(lambda: ops.groupnorm(yv, B, 32, ga, be, 1e-5, True)) if has else float("nan")
    plain = y.view(B, HW, C).clone()
    t_split = timeit(lambda: ops.groupnorm(plain, B, 32, ga, be, 1e-5, True))
    mb = 2 * B * HW * C * 2 / 1e6
    print(f"B={B} HW={HW} C={C}: from producer statistics {t_cs:6.1f} us ({mb / t_cs:.2f} TB/s)   two-launch {t_split:6.1f} us")

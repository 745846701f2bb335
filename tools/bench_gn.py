import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gm-diffusion_amd"))
import torch
from gm_diffusion import hip_ops as ops

def timeit(fn, reps=100):
    fn(); fn(); torch.cuda.synchronize()
    torch.cuda._sleep(int(2e7))  # device-side lead: the host gets ahead, launch latency is not counted
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

g = torch.Generator().manual_seed(0)
for B in (8, 4):
    for HW, C in [(4096, 320), (4096, 640), (4096, 960), (1024, 640), (1024, 1280), (1024, 1920), (256, 1280), (256, 1920), (256, 2560), (64, 1280), (64, 1920), (64, 2560)]:
        x = torch.randn(B, HW, C, generator=g).bfloat16().cuda()
        ga, be = torch.randn(C, generator=g).cuda(), torch.randn(C, generator=g).cuda()
        slab = HW * (C // 32) * 2
        ts = timeit(lambda: ops.groupnorm_split(x, B, 32, ga, be, 1e-5, True))
        t2 = timeit(lambda: ops.groupnorm(x, B, 32, ga, be, 1e-5, True))
        tf = float("nan")
        if slab <= 128 * 1024:
            y = torch.empty_like(x)
            from gm_diffusion._native import lib, check
            def f():
                check(lib().gmd_groupnorm_fused(x.data_ptr(), y.data_ptr(), 1, B, HW, C, 32, 1e-5, ga.data_ptr(), be.data_ptr(), 1, torch.cuda.current_stream().cuda_stream), "f")
            tf = timeit(f)
        print(f"B={B} HW={HW} C={C} slab={slab//1024}KB split3={ts:6.1f}us fused={tf:6.1f}us dispatch={t2:6.1f}us")

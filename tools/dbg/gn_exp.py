import os, sys
sys.path.insert(0, "/root/repo/gm-diffusion_amd")
import torch
from gm_diffusion import hip_ops as ops
def timeit(fn, reps=200):
    fn(); fn(); torch.cuda.synchronize()
    torch.cuda._sleep(int(2e7))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
g = torch.Generator().manual_seed(0)
print("U:", os.environ.get("GMD_GN_U", "auto"))
for B, HW, C in [(8, 4096, 320), (4, 4096, 320), (8, 1024, 640), (4, 1024, 640), (8, 4096, 640)]:
    x = torch.randn(B * HW, C, generator=g).bfloat16().cuda()
    w = (torch.randn(C, C, generator=g) * 0.05).bfloat16().cuda()
    ga, be = torch.randn(C, generator=g).cuda(), torch.randn(C, generator=g).cuda()
    y = ops.gemm_nt(x, w, colstats=True)
    yv = ops.carry_colstats(y.view(B, HW, C), y)
    t1 = timeit(lambda: ops.groupnorm(yv, B, 32, ga, be, 1e-5, True))
    t0 = timeit(lambda: ops.groupnorm(yv, B, 32, ga, be, 1e-5, False))
    print(f"B={B} HW={HW} C={C}: silu {t1:6.1f} us   no silu {t0:6.1f} us")

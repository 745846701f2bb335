import os, sys
sys.path.insert(0, "/root/repo/gm-diffusion_amd")
import torch
from gm_diffusion import hip_ops as ops
def t(M, N, K, reps=200, **kw):
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
    for _ in range(3): ops.gemm_nt(a, w, **kw)
    torch.cuda.synchronize(); torch.cuda._sleep(int(1e7))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ops.gemm_nt(a, w, **kw)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for M, N in ((32768, 320), (8192, 640), (32768, 640)):
    r = torch.randn(M, N, device="cuda").bfloat16()
    print(f"M={M} N={N} K=64: bf16 out {t(M,N,64):5.1f}  f32 out {t(M,N,64,out_dtype=torch.float32):5.1f}  bf16+residual {t(M,N,64,residual=r):5.1f} us")
    print(f"M={M} N={N} K=320: bf16 out {t(M,N,320):5.1f}  f32 out {t(M,N,320,out_dtype=torch.float32):5.1f}  bf16+residual {t(M,N,320,residual=r):5.1f} us")
x = torch.randn(32768, 320, device="cuda").bfloat16()
torch.cuda.synchronize(); torch.cuda._sleep(int(1e7))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): y = x.clone()
e1.record(); torch.cuda.synchronize()
print("torch clone of 21 MB: %.1f us" % (e0.elapsed_time(e1) / 200 * 1e3))

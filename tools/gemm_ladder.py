"""Fixed cost vs per-K-step cost of the GEMM kernels (diagnostic): time(K) ladder at fixed M, N."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gm-diffusion_amd"))
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

for M, N in ((8192, 640), (32768, 320), (2048, 1280), (128, 160), (65536, 320)):
    print(f"M={M} N={N}: " + "  ".join(f"K={K}:{t(M, N, K):6.1f}" for K in (64, 128, 320, 640, 1280, 2560)))
x = torch.zeros(64, device="cuda")
torch.cuda.synchronize(); torch.cuda._sleep(int(1e7))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): ops.cast(x, torch.bfloat16) if hasattr(ops, "cast") else x.add_(1)
e1.record(); torch.cuda.synchronize()
print("trivial launch us:", e0.elapsed_time(e1) / 200 * 1e3)

"""Measure what a torch.cuda.Event pair adds around one launch on a busy stream (diagnostic for bench.py's roofline)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gm-diffusion_amd"))
import torch
from gm_diffusion import hip_ops as ops

x = torch.zeros(64, device="cuda")
big = torch.randn(8192, 640, device="cuda").bfloat16(); w = torch.randn(640, 640, device="cuda").bfloat16()
def pairs(fn, n=400):
    torch.cuda.synchronize(); torch.cuda._sleep(int(3e7))
    ev = []
    for _ in range(n):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); ev.append((a, b))
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return t[len(t) // 2], sum(t) / len(t)
def bulk(fn, n=400):
    torch.cuda.synchronize(); torch.cuda._sleep(int(3e7))
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n
print("empty pair       median/mean us:", pairs(lambda: None))
print("tiny kernel pair median/mean us:", pairs(lambda: x.add_(1.0)), " bulk:", bulk(lambda: x.add_(1.0)))
print("gemm 8192x640x640 pair:", pairs(lambda: ops.gemm_nt(big, w)), " bulk:", bulk(lambda: ops.gemm_nt(big, w)))

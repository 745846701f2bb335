#!/usr/bin/env python
"""Fixed vs per-K-step cost of gmd_gemm_nt: device time (HIP-graph replay, rotated buffers) against K at fixed M, N."""
import os, sys
os.environ.setdefault("GMD_TUNING", "1")  # kernel-plan overrides are a debug facility (include/gmd_hip.h)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gm-diffusion_amd")):
    sys.path.insert(0, p)
import torch
from gm_diffusion import hip_ops as ops
from gm_diffusion._native import lib
if len(sys.argv) > 1:
    lib().gmd_gemm_plan_override(*[int(v) for v in sys.argv[1].split(",")])
dev = "cuda"
g = torch.Generator().manual_seed(0)
bf = lambda *s: (torch.randn(*s, generator=g) * 0.5).bfloat16().to(dev)


def graph_time(fns, reps=24):
    for f in fns: f()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    ws = ops.new_workspace(dev)
    with torch.cuda.stream(s):
        with ops.workspace_scope(ws), torch.cuda.graph(gr):
            for r in range(reps): fns[r % len(fns)]()
    torch.cuda.synchronize(); gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


for M, N in [(512, 1280), (2048, 1280), (8192, 640), (32768, 320)]:
    row = []
    for K in (64, 128, 320, 640, 1280, 2560):
        fns = []
        for i in range(6):
            x, w, b = bf(M, K), bf(N, K) * 0.05, torch.randn(N, generator=g).to(dev)
            fns.append(lambda x=x, w=w, b=b: ops.gemm_nt(x, w, bias=b))
        row.append(f"K={K}: {graph_time(fns):6.1f}")
    print(f"M={M:6d} N={N:5d}  " + "  ".join(row), flush=True)
# an empty-ish kernel for the boundary cost: LayerNorm of 64 rows
x, ga, be = bf(64, 320), torch.ones(320, device=dev), torch.zeros(320, device=dev)
print("layernorm 64x320 (launch boundary reference): %.1f us" % graph_time([lambda: ops.layernorm(x, ga, be)]))

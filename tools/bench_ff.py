#!/usr/bin/env python
"""GEGLU feed-forward at level 0 (C = 320): the fused kernel (csrc/ff_fused.hip) against the two GEMM launches it replaces,
device time inside a HIP graph with rotating buffers, interleaved rounds in one process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gm-diffusion_amd")):
    sys.path.insert(0, p)
import torch
from gm_diffusion import hip_ops as ops

dev = "cuda"
g = torch.Generator().manual_seed(0)
C = 320


def graph_time(fns, reps=20):
    for f in fns:
        f()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(gr):
            for r in range(reps):
                fns[r % len(fns)]()
    torch.cuda.synchronize()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    return min(ts), sorted(ts)[len(ts) // 2]


for dtype in (torch.bfloat16, torch.float16):
    for M in (32768, 16384, 65536):
        sets = []
        for i in range(4):
            x = (torch.randn(M, C, generator=g)).to(dtype).to(dev)
            r = (torch.randn(M, C, generator=g)).to(dtype).to(dev)
            w1 = (torch.randn(8 * C, C, generator=g) * 0.05).to(dtype).to(dev)
            b1 = torch.randn(8 * C, generator=g).to(dev)
            w2 = (torch.randn(C, 4 * C, generator=g) * 0.03).to(dtype).to(dev)
            b2 = torch.randn(C, generator=g).to(dev)
            sets.append((x, r, w1, b1, w2, b2))
        fused = [(lambda s=s: ops.ff_geglu_fused(s[0], s[2], s[3], s[4], s[5], s[1])) for s in sets]
        two = [(lambda s=s: ops.gemm_nt(ops.gemm_nt(s[0], s[2], bias=s[3], act=ops.ACT_GEGLU), s[4], bias=s[5], residual=s[1])) for s in sets]
        fl = 2.0 * M * C * 8 * C + 2.0 * M * 4 * C * C
        res = {}
        for rnd in range(2):
            for name, fns in (("fused", fused), ("two", two)):
                res.setdefault(name, []).append(graph_time(fns))
        f_ = min(v[0] for v in res["fused"]); t_ = min(v[0] for v in res["two"])
        print(f"{str(dtype):16s} M={M:6d}: fused {f_:7.1f} us ({fl / f_ / 1e6:6.0f} TF/s)   two launches {t_:7.1f} us ({fl / t_ / 1e6:6.0f} TF/s)   ratio {f_ / t_:.3f}")

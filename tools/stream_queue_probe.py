#!/usr/bin/env python
"""Which torch pool streams really run beside the current (null) stream?  HIP multiplexes its streams onto a few hardware queues;
two streams that share one are serialised.  For each of the first N pool streams: a half-chip GEMM chain on the null stream and
the same chain on the pool stream, wall time against the one-stream time (1.0 = serialised, ~0.5 = side by side)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gm-diffusion_amd"))
import torch
from gm_diffusion import hip_ops as ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
a = torch.randn(16384, 1280, device="cuda").bfloat16()  # 128 row blocks of 128: half of the chip's workgroup slots
w = (torch.randn(1280, 1280, device="cuda") * 0.03).bfloat16()
b = a.clone()


def chain(x, reps=40):
    for _ in range(reps):
        ops.gemm_nt(x, w)


def wall(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


chain(a); chain(b)
one = min(wall(lambda: chain(a)) for _ in range(3))
print(f"one stream, one chain: {one:.2f} ms; two chains on the null stream: {min(wall(lambda: (chain(a), chain(b))) for _ in range(3)):.2f} ms")
cur = torch.cuda.current_stream()
for k in range(n):
    s = torch.cuda.Stream()

    def both():
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            chain(b)
        chain(a)
        cur.wait_stream(s)

    both()
    t = min(wall(both) for _ in range(3))
    print(f"pool stream {k:2d} (id {s.stream_id:#x}, ptr {s.cuda_stream:#x}): two chains {t:.2f} ms = {t / (2 * one):.2f} of serial")

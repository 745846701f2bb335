#!/usr/bin/env python
"""CFG duplication of a shared-prefix tensor: gmd_dup_batch against the two runtime device-to-device copies it replaces."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gm-diffusion_amd")):
    sys.path.insert(0, p)
import torch
from gm_diffusion import hip_ops as ops

def timeit(fn, reps=50):
    fn(); fn(); torch.cuda.synchronize()
    torch.cuda._sleep(int(2e7))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for shape in ((4, 4096, 320), (2, 4096, 320), (8, 4096, 320)):
    t = torch.randn(*shape, device="cuda").bfloat16()
    out = torch.empty((2 * shape[0],) + shape[1:], dtype=t.dtype, device="cuda")
    def two():
        out[: shape[0]].copy_(t); out[shape[0]:].copy_(t)
    a = timeit(two); b = timeit(lambda: ops.dup_batch(t))
    mb = t.numel() * 2 / 1e6
    print(f"{shape}: {mb:.1f} MB  two copies {a:.1f} us   dup kernel {b:.1f} us ({3 * mb / b / 1e3:.2f} TB/s)")

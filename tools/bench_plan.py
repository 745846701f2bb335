#!/usr/bin/env python
"""Plan check: time selected GEMM shapes of the UNet under the heuristic and forced variants (diagnostic)."""
import os, sys, subprocess
os.environ.setdefault("GMD_TUNING", "1")  # kernel-plan overrides are a debug facility (include/gmd_hip.h)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(2048, 2560, 1280), (2048, 1280, 5120), (8192, 640, 2560), (8192, 1280, 640), (1024, 1280, 5120), (4096, 640, 2560), (512, 1280, 5120),
          (512, 2560, 1280), (1024, 2560, 1280), (4096, 1280, 640), (16384, 640, 320), (2048, 1280, 2560), (512, 1280, 2560)]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    for p in (ROOT, os.path.join(ROOT, "gm-diffusion_amd")):
        sys.path.insert(0, p)
    import torch
    from gm_diffusion import hip_ops as ops
    g = torch.Generator().manual_seed(0)
    for M, N, K in SHAPES:
        a = torch.randn(M, K, generator=g).bfloat16().cuda(); w = (torch.randn(N, K, generator=g) * 0.02).bfloat16().cuda(); b = torch.randn(N, generator=g).cuda()
        r = torch.randn(M, N, generator=g).bfloat16().cuda()
        f = lambda: ops.gemm_nt(a, w, bias=b, residual=r)
        f(); f(); torch.cuda.synchronize(); torch.cuda._sleep(int(1e7))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): f()
        e1.record(); torch.cuda.synchronize()
        print(f"{M}x{N}x{K} {e0.elapsed_time(e1) / 100 * 1e3:.1f}")
else:
    variants = ["0,0,0,0", "64,64,9,1", "128,160,0,1", "128,160,0,2", "128,160,0,4", "128,128,0,1"]
    res = {}
    for v in variants:
        out = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, GMD_GEMM_FORCE=v), capture_output=True, text=True).stdout
        for l in out.splitlines():
            if "x" in l and " " in l:
                k, t = l.split()
                res.setdefault(k, {})[v] = float(t)
    print(" " * 18 + " ".join(f"{v:>13s}" for v in variants))
    for k, d in res.items():
        print(f"{k:18s}" + " ".join(f"{d.get(v, 0):13.1f}" for v in variants))

#!/usr/bin/env python
"""Where a wave of gemm_pp_kernel spends its loop time: reads the stamp sums the DIAGNOSTIC build (tools/dbg/build_ppdiag.sh,
GMD_LIB_OVERRIDE=tools/dbg/libgmd_ppdiag.so) leaves in the workspace.  Shares only -- the stamped build is never timed."""
import os, sys
os.environ.setdefault("GMD_TUNING", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("GMD_LIB_OVERRIDE", os.path.join(ROOT, "tools", "dbg", "libgmd_ppdiag.so"))
sys.path.insert(0, os.path.join(ROOT, "gm-diffusion_amd"))
import numpy as np
import torch
from gm_diffusion import hip_ops as ops
from gm_diffusion._native import lib

plan = sys.argv[1] if len(sys.argv) > 1 else "256,160,283,1"
lib().gmd_gemm_plan_override(*[int(x) for x in plan.split(",")])
g = torch.Generator().manual_seed(0)
NAMES = ["reads + wait", "dma issue", "-", "barrier after R / loader barrier", "mfma issue", "barrier after C", "vmcnt wait"]
NWV = 12  # 8 consumer waves + 4 loader waves


def report(name, fn, nwg):
    ws = ops._workspace(torch.device("cuda", torch.cuda.current_device()))
    for _ in range(20): fn()  # warm clocks
    torch.cuda.synchronize()
    ws.view(torch.uint8)[: nwg * NWV * 10 * 8].zero_()
    fn(); torch.cuda.synchronize()
    d = ws.view(torch.uint8)[: nwg * NWV * 10 * 8].cpu().numpy().view(np.uint64).reshape(nwg, NWV, 10).astype(np.float64)
    nk = d[0, 0, 9]
    print(f"{name}: {nwg} workgroups, {int(nk)} K steps; per K step and wave, shader cycles (early consumers 0-3 | late consumers 4-7 | loaders 8-11); every stamp adds ~40")
    for k, nm in enumerate(NAMES):
        print(f"   {nm:34s} {d[:, :4, k].mean() / nk:8.1f} | {d[:, 4:8, k].mean() / nk:8.1f} | {d[:, 8:, k].mean() / nk:8.1f}")
    tot = d[:, :, 8].mean() / nk
    clk = (d[:, :, 8] / d[:, :, 7]).mean() * 100.0  # MHz: shader cycles per 100 MHz tick
    print(f"   loop total         {tot:8.1f} cycles per K step; in-kernel clock {clk:.0f} MHz (stamped build)")


for B, H, ci, co in [(8, 64, 640, 320), (8, 64, 320, 320)]:
    x = torch.randn(B, H * H, ci, generator=g).bfloat16().cuda()
    w = (torch.randn(co, 9 * ci, generator=g) * 0.02).bfloat16().cuda()
    b = torch.randn(co, generator=g).cuda()
    bn = int(plan.split(",")[1])
    report(f"conv B={B} {H}x{H} {ci}->{co}", lambda: ops.conv3x3(x, w, B, H, H, bias=b), (B * H * H // 256) * ((co + bn - 1) // bn))
for M, N, K in [(32768, 320, 1280), (32768, 320, 320)]:
    a = torch.randn(M, K, generator=g).bfloat16().cuda()
    w = (torch.randn(N, K, generator=g) * 0.02).bfloat16().cuda()
    b = torch.randn(N, generator=g).cuda()
    bn = int(plan.split(",")[1])
    report(f"gemm M={M} N={N} K={K}", lambda: ops.gemm_nt(a, w, bias=b), (M // 256) * ((N + bn - 1) // bn))

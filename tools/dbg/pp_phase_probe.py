#!/usr/bin/env python
"""Phase medians (kernel entry -> K loop | K loop | barrier + reduction | epilogue) of single gemm_pp launches from the stamped
diagnostic library (GMD_LIB_OVERRIDE=tools/dbg/libgmd_wgtrace_pp*.so), under the co-running plan family."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GMD_LIB_OVERRIDE", os.path.join(ROOT, "tools", "dbg", "libgmd_wgtrace_pp.so"))
sys.path.insert(0, os.path.join(ROOT, "gm-diffusion_amd"))
import numpy as np, torch
from gm_diffusion import hip_ops as ops
from gm_diffusion._native import lib
SH, CAP = 2048, 4096
en = lib().gmd_wg_trace_enable; en.argtypes = [ctypes.c_void_p]; en.restype = ctypes.c_int
lib().gmd_gemm_plan_family(1)
g = torch.Generator().manual_seed(0)
for M, N, K, res in [(32768, 320, 320, 0), (32768, 320, 320, 1), (8192, 640, 640, 1), (8192, 640, 640, 0), (32768, 960, 320, 0), (2048, 1280, 1280, 1)]:
    a = torch.randn(M, K, generator=g).bfloat16().cuda(); w = (torch.randn(N, K, generator=g) * 0.03).bfloat16().cuda()
    b = torch.randn(N, generator=g).cuda(); r = torch.randn(M, N, generator=g).bfloat16().cuda() if res else None
    for _ in range(5): ops.gemm_nt(a, w, bias=b, residual=r)
    torch.cuda.synchronize()
    ring = torch.zeros(16 + 16 * SH + 4 * SH * CAP, dtype=torch.int64, device="cuda"); ring[0] = SH; ring[1] = CAP
    torch.cuda.synchronize(); assert en(ring.data_ptr()) == 0
    for _ in range(10): ops.gemm_nt(a, w, bias=b, residual=r)
    torch.cuda.synchronize(); en(None)
    cnt = ring[16:16 + 16 * SH:16].cpu().numpy(); body = ring[16 + 16 * SH:].view(SH, CAP, 4)
    rec = np.concatenate([body[i, :min(int(n), CAP)].cpu().numpy() for i, n in enumerate(cnt) if n > 0]).view(np.uint64)
    tag = (rec[:, 2] >> np.uint64(32)).astype(np.int64)
    ph = rec[((((tag >> 8) & 255) & 0x80) != 0) & (((tag >> 16) & 255) == 255)]
    nm = rec[(((tag >> 8) & 255) & 0x80) == 0]
    f = lambda col, hi: ((ph[:, col] >> np.uint64(32)) if hi else (ph[:, col] & np.uint64(0xffffffff))).astype(np.int64) / 100.0
    whole = (nm[:, 1].astype(np.int64) - nm[:, 0].astype(np.int64)) / 100.0
    print(f"M={M} N={N} K={K} res={res} plan={ops.gemm_plan_info(torch.bfloat16, M, N, K)}: prologue {np.median(f(0,0)):5.2f}  loop {np.median(f(0,1)):5.2f}  sync {np.median(f(1,0)):5.2f}  "
          f"epilogue {np.median(f(1,1)):5.2f}  | whole workgroup {np.median(whole):5.2f} us ({len(ph)} workgroups)", flush=True)
    ep = rec[((((tag >> 8) & 255) & 0x80) != 0) & (((tag >> 16) & 255) == 254)]
    if len(ep):
        q = lambda col, hi: ((ep[:, col] >> np.uint64(32)) if hi else (ep[:, col] & np.uint64(0xffffffff))).astype(np.int64) / 100.0
        print(f"      epilogue of wave 0: stage half 0 {np.median(q(0,0)):5.2f} | items half 0 {np.median(q(0,1)):5.2f} | stage half 1 {np.median(q(1,0)):5.2f} | items half 1 {np.median(q(1,1)):5.2f} us", flush=True)
    del ring, body

import os, sys
os.environ["GMD_TUNING"] = "1"
ROOT = "/root/repo"
sys.path.insert(0, ROOT + "/gm-diffusion_amd")
import torch
from gm_diffusion import hip_ops as ops
from gm_diffusion._native import lib
g = torch.Generator().manual_seed(0)
def t(M, N, K, res, plan, reps=24, sets=6):
    lib().gmd_gemm_plan_override(*plan)
    bufs = []
    for i in range(sets):
        a = torch.randn(M, K, generator=g).bfloat16().cuda(); w = (torch.randn(N, K, generator=g) * 0.03).bfloat16().cuda()
        b = torch.randn(N, generator=g).cuda(); r = torch.randn(M, N, generator=g).bfloat16().cuda() if res else None
        bufs.append((a, w, b, r))
    fns = [(lambda s=s: ops.gemm_nt(s[0], s[1], bias=s[2], residual=s[3])) for s in bufs]
    for f in fns: f()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph(); st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    ws = ops.new_workspace("cuda")
    with torch.cuda.stream(st):
        with ops.workspace_scope(ws), torch.cuda.graph(gr):
            for i in range(reps): fns[i % sets]()
    torch.cuda.synchronize(); gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    lib().gmd_gemm_plan_override(0, 0, 0, 0)
    return best
for M, N, K in [(8192, 640, 640), (32768, 640, 320), (16384, 640, 320), (8192, 1280, 640), (2048, 1280, 1280), (32768, 640, 640)]:
    for res in (False, True):
        a = t(M, N, K, res, (256, 160, 283, 1)); b = t(M, N, K, res, (256, 128, 283, 1))
        print(f"M={M} N={N} K={K} res={int(res)}: 256x160 tiles {a:6.1f} us   256x128 tiles {b:6.1f} us   ({(M//256)*(N//160)} vs {(M//256)*(N//128)} workgroups)")

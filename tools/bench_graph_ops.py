#!/usr/bin/env python
"""Device-side time of single ops as the product runs them: `reps` launches captured into ONE HIP graph (no host launch
overhead between them; the eager Python front end costs ~10-15 us per call and hides anything shorter), operands rotated
over `sets` buffer sets so consecutive launches do not find their inputs in L2 / Infinity Cache.
Usage: bench_graph_ops.py [--batch 8] [--only gemm,conv,ln,gn,attn] [--force bm,bn,pf,ks]"""
import argparse, os, sys
os.environ.setdefault("GMD_TUNING", "1")  # kernel-plan overrides are a debug facility (include/gmd_hip.h)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gm-diffusion_amd")):
    sys.path.insert(0, p)
import torch
from gm_diffusion import hip_ops as ops
from gm_diffusion._native import lib

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--only", default="gemm,conv,ln,gn,attn")
ap.add_argument("--force", default="")
ap.add_argument("--reps", type=int, default=24)
ap.add_argument("--sets", type=int, default=6)
ap.add_argument("--family", type=int, default=0, help="launch-plan family (gmd_gemm_plan_family): 1 = the co-running family of the dual pipeline")
a = ap.parse_args()
lib().gmd_gemm_plan_family(a.family)
if a.force:
    lib().gmd_gemm_plan_override(*[int(v) for v in a.force.split(",")])
dev = "cuda"
g = torch.Generator().manual_seed(0)
B = a.batch
bf = lambda *s: (torch.randn(*s, generator=g) * 0.5).bfloat16().to(dev)


def graph_time(make, reps=a.reps, sets=a.sets):
    try:
        return _graph_time(make, reps, sets)
    except Exception as e:  # a forced plan that has no kernel for this op (e.g. GEGLU on an odd-TN tile)
        return float("nan")


def _graph_time(make, reps, sets):
    """make(i) -> closure launching the op on buffer set i.  Returns us per launch."""
    fns = [make(i) for i in range(sets)]
    for f in fns:
        f()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    ws = ops.new_workspace(dev)
    with torch.cuda.stream(s):
        with ops.workspace_scope(ws), torch.cuda.graph(gr):
            for r in range(reps):
                fns[r % sets]()
    torch.cuda.synchronize()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


rows = []
levels = [(64, 320), (32, 640), (16, 1280), (8, 1280)]
want = set(a.only.split(","))
for hw, C in levels:
    M = B * hw * hw
    if "gemm" in want and hw > 8 or ("gemm" in want and hw == 8):
        def mk_lin(N, K, res=True, act=ops.ACT_NONE):
            def make(i):
                x, w, b = bf(M, K), bf(N, K) * 0.05, torch.randn(N, generator=g).to(dev)
                r = bf(M, N) if res else None
                return lambda: ops.gemm_nt(x, w, bias=b, residual=r, act=act)
            return make
        rows.append((f"gemm  M={M:6d} N={C:5d} K={C:5d} +res (o1/o2/pout)", 2.0 * M * C * C, graph_time(mk_lin(C, C))))
        rows.append((f"gemm  M={M:6d} N={C:5d} K={C:5d}      (q2/pin)", 2.0 * M * C * C, graph_time(mk_lin(C, C, res=False))))
        rows.append((f"gemm  M={M:6d} N={2*C:5d} K={C:5d}      (qk)", 2.0 * M * 2 * C * C, graph_time(mk_lin(2 * C, C, res=False))))
        rows.append((f"gemm  M={M:6d} N={8*C:5d} K={C:5d} geglu (ff1)", 2.0 * M * 8 * C * C, graph_time(mk_lin(8 * C, C, res=False, act=ops.ACT_GEGLU))))
        rows.append((f"gemm  M={M:6d} N={C:5d} K={4*C:5d} +res (ff2)", 2.0 * M * C * 4 * C, graph_time(mk_lin(C, 4 * C))))
        def mk_vt(i):
            x, w = bf(B, hw * hw, C), bf(C, C) * 0.05
            return lambda: ops.gemm_nt(w, x, ldc=hw * hw)
        rows.append((f"gemm  vT b={B} M={C:5d} N={hw*hw:5d} K={C:5d}", 2.0 * M * C * C, graph_time(mk_vt)))
    if "ln" in want:
        def mk_ln(i):
            x, ga, be = bf(M, C), torch.ones(C, device=dev), torch.zeros(C, device=dev)
            return lambda: ops.layernorm(x, ga, be)
        t = graph_time(mk_ln)
        rows.append((f"ln    rows={M:6d} C={C:5d}   [{2 * M * C * 2 / t / 1e3:6.0f} GB/s]", 0.0, t))
    if "gn" in want:
        for Cg in ({320: (320, 640, 960), 640: (640, 1280, 1920), 1280: (1280, 2560)}[C] if hw > 8 else (1280, 2560)):
            def mk_gn(i, Cg=Cg):
                x, ga, be = bf(B, hw * hw, Cg), torch.ones(Cg, device=dev), torch.zeros(Cg, device=dev)
                return lambda: ops.groupnorm(x, B, 32, ga, be, 1e-5, silu=True)
            t = graph_time(mk_gn)
            rows.append((f"gn    B={B} rows={hw*hw:5d} C={Cg:5d} [{2 * M * Cg * 2 / t / 1e3:6.0f} GB/s]", 0.0, t))
    if "conv" in want:
        for ci, co in ({320: ((320, 320), (640, 320), (960, 320)), 640: ((640, 640), (320, 640), (1280, 640), (1920, 640)),
                        1280: ((1280, 1280), (640, 1280), (2560, 1280), (1920, 1280))}[C] if hw > 8 else ((1280, 1280), (2560, 1280))):
            def mk_conv(i, ci=ci, co=co):
                x, w, b, r = bf(B, hw * hw, ci), bf(co, 9 * ci) * 0.02, torch.randn(co, generator=g).to(dev), bf(B, hw * hw, co)
                return lambda: ops.conv3x3(x, w, B, hw, hw, bias=b, residual=r)
            rows.append((f"conv  B={B} {hw:3d}x{hw:<3d} {ci:4d}->{co:4d} +res", 2.0 * M * co * 9 * ci, graph_time(mk_conv, sets=3)))
    if "attn" in want and hw > 8:
        N = hw * hw
        d = C // 8
        def mk_attn(i):
            qk, vt = bf(B, N, 2 * C), bf(B, C, N)
            return lambda: ops.attention(qk, qk, vt, 8, N, d ** -0.5, k_col=C)
        rows.append((f"attn  self B={B} N={N:5d} d={d:3d}", 4.0 * B * N * N * C, graph_time(mk_attn, sets=3)))
        def mk_xattn(i):
            q, k, vt = bf(B, N, C), bf(B, 77, C), bf(B, C, 80)
            return lambda: ops.attention(q, k, vt, 8, 77, d ** -0.5)
        rows.append((f"attn  cross B={B} Nq={N:5d} Nk=77 d={d:3d}", 4.0 * B * N * 77 * C, graph_time(mk_xattn)))
print(f"{'op':58s} {'us':>8s} {'TF/s':>8s}")
for name, fl, t in rows:
    print(f"{name:58s} {t:8.1f} {fl / t / 1e6 if fl else 0:8.0f}")

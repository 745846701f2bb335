"""Stress test of the producer column statistics (gemm_shared.h: colstats_pass / colstats_store) under a second stream's load.

GEMM / conv launches with ``colstats=True`` are replayed from a HIP graph on one stream while the side stream replays another graph
(BG = unet | conv | gemm | attn | gn | copy | none; unet = a whole SD-1.5-width float32 UNet forward, the load that exposed the fault
described in DESIGN.md §4.5).  Every launch's output and statistics are compared bit for bit with a quiet run; exits 1 on a difference.

    DTYPE=f32|bf16  BG=unet  NREP=150  INNER=12  python tools/stress_colstats.py
    GMD_LIB_OVERRIDE=<other build>/libgmd_hip.so ...   # A/B against another build of the library"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gm-diffusion_amd")):
    sys.path.insert(0, p)
import torch
from gm_diffusion import hip_ops as ops
ops.set_f32_mode("split")
DTYPE = os.environ.get("DTYPE", "f32")
DT = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[DTYPE]
prep = (lambda w: ops.split_weights(w)) if DTYPE == "f32" else (lambda w: w.to(DT))
act = lambda t: t.to(DT)
DEV = "cuda"
BG = os.environ.get("BG", "unet")
NREP = int(os.environ.get("NREP", "150"))
g = torch.Generator().manual_seed(0)
# foreground: the launches the pipeline probe flagged (transformer proj_out at 32x32 and 64x64) + a conv
a1 = act(torch.randn(8192, 640, generator=g).to(DEV)); w1 = prep((torch.randn(640, 640, generator=g) * 0.04).to(DEV))
b1 = torch.randn(640, generator=g).to(DEV); r1 = act(torch.randn(8192, 640, generator=g).to(DEV))
a2 = act(torch.randn(32768, 320, generator=g).to(DEV)); w2 = prep((torch.randn(320, 320, generator=g) * 0.05).to(DEV))
b2 = torch.randn(320, generator=g).to(DEV); r2 = act(torch.randn(32768, 320, generator=g).to(DEV))
x3 = act(torch.randn(8, 1024, 640, generator=g).to(DEV)); w3 = prep((torch.randn(640, 9 * 640, generator=g) * 0.02).to(DEV))
INNER = int(os.environ.get('INNER', '12'))
def _st(y):
    st = getattr(y, '_colstats', None)
    if st is None:
        raise SystemExit('this launch takes a plan without producer statistics: ' + str(tuple(y.shape)))
    return st[0]
def fg():
    outs = []
    for _ in range(INNER):
        y = ops.gemm_nt(a1, w1, bias=b1, residual=r1, colstats=True); outs.append((y, _st(y)))
        y = ops.gemm_nt(a2, w2, bias=b2, residual=r2, colstats=True); outs.append((y, _st(y)))
        y, _, _ = ops.conv3x3(x3, w3, 8, 32, 32, bias=b1, colstats=True); outs.append((y, _st(y)))
    return outs
def capture(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    ws = ops.new_workspace(DEV)
    with ops.workspace_scope(ws), torch.cuda.graph(gr):
        out = fn()
    return gr, out, ws
def make_bg():
    if BG == "none":
        return None
    if BG == "unet":
        from gm_diffusion.components import UNet2DConditionModel
        u = UNet2DConditionModel(in_channels=8).init_random(8, device=DEV).to(DEV, DT)
        ehs = torch.randn(8, 77, 768, generator=g).to(DEV)
        gf = u.graphed_forward(8, 64, 64, u.prepare_context(ehs), cfg_shared=u.supports_cfg_shared())
        gf.x.normal_()
        u.set_timestep(500)
        return lambda: gf.replay()
    if BG == "conv":
        x = torch.randn(8, 4096, 320, generator=g).to(DEV); w = ops.split_weights((torch.randn(320, 2880, generator=g) * 0.02).to(DEV))
        fn = lambda: [ops.conv3x3(x, w, 8, 64, 64)[0] for _ in range(20)]
    elif BG == "gemm":
        a = torch.randn(32768, 1280, generator=g).to(DEV); w = ops.split_weights((torch.randn(320, 1280, generator=g) * 0.02).to(DEV))
        fn = lambda: [ops.gemm_nt(a, w) for _ in range(40)]
    elif BG == "attn":
        qk = torch.randn(8, 4096, 640, generator=g).to(DEV); vt = torch.randn(8, 320, 4096, generator=g).to(DEV)
        fn = lambda: [ops.attention(qk, qk, vt, 8, 4096, 40 ** -0.5, k_col=320) for _ in range(6)]
    elif BG == "gn":
        x = torch.randn(8, 4096, 320, generator=g).to(DEV); ga = torch.ones(320, device=DEV); be = torch.zeros(320, device=DEV)
        fn = lambda: [ops.groupnorm(x, 8, 32, ga, be, 1e-5, silu=True) for _ in range(60)]
    elif BG == "copy":
        x = torch.randn(64 << 20, generator=g).to(DEV)
        fn = lambda: [x * 1.0001 for _ in range(40)]
    gr, out, ws = capture(fn)
    KEEP.append((out, ws))
    return lambda: gr.replay()
KEEP = []
gr, outs, ws = capture(fg)
gr.replay(); torch.cuda.synchronize()
ref = [(y.clone(), s.clone()) for y, s in outs]
for i in range(1, INNER):  # same launch -> same result, already in the quiet run
    for j in range(3):
        assert torch.equal(ref[3 * i + j][0], ref[j][0]) and torch.equal(ref[3 * i + j][1], ref[j][1]), "quiet run is not reproducible"
bg = make_bg()
sa, sb = torch.cuda.Stream(), ops.side_stream(DEV)
names = ["gemm 8192x640x640", "gemm 32768x320x320", "conv 8x32x32 640->640"]
bad = [0, 0, 0]; bady = [0, 0, 0]; shown = 0
for rep in range(NREP):
    for y, s in outs:
        s.zero_()
    torch.cuda.synchronize()
    if bg is not None:
        with torch.cuda.stream(sb):
            bg(); bg()
    with torch.cuda.stream(sa):
        gr.replay()
    torch.cuda.synchronize()
    for k, ((y, s), (yr, sr)) in enumerate(zip(outs, ref)):
        if not torch.equal(y, yr):
            bady[k % 3] += 1
        if not torch.equal(s, sr):
            bad[k % 3] += 1
            if shown < 6:
                shown += 1
                dd = (s != sr).flatten().nonzero().flatten()
                nbk = s.shape[1]
                print(f"rep {rep} launch {k} ({names[k % 3]}): {len(dd)} statistics entries differ (row block, bucket, s/q):",
                      [(int(j) // (2 * nbk), (int(j) // 2) % nbk, int(j) % 2) for j in dd[:9]], "got", [round(v, 3) for v in s.flatten()[dd[:6]].tolist()],
                      "want", [round(v, 3) for v in sr.flatten()[dd[:6]].tolist()], flush=True)
print(f"{DTYPE} BG={BG}: launches with differing statistics {dict(zip(names, bad))}, differing outputs {dict(zip(names, bady))} over {NREP} x {INNER}")
sys.exit(1 if sum(bad) + sum(bady) else 0)

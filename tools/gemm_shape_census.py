#!/usr/bin/env python
"""Which gemm_nt / conv3x3 shapes carry the time of one dual-UNet loop iteration (SDR UNet at CFG batch 8 with the shared prefix + GM UNet
at batch 4, 64x64 latent, bf16): every launch of one eager single-stream iteration bracketed by HIP events, folded by (kind, M, N, K,
epilogue).  Feeds the launch-weighted PMC traffic record (tools/collect_pmc_traffic_r4.sh).  Prints JSON lines, largest total first."""
import collections, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gm-diffusion_amd")):
    sys.path.insert(0, p)
import torch
from gm_diffusion import hip_ops as ops
from gm_diffusion.components import UNet2DConditionModel

dev = torch.device("cuda", 0)
rec = collections.OrderedDict()
orig_gemm, orig_conv = ops.gemm_nt, ops.conv3x3
ACTIVE = [False]


def timed(key, fn):
    if not ACTIVE[0]:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = fn(); e1.record()
    rec.setdefault(key, []).append((e0, e1))
    return out


def gemm_nt(a, w, bias=None, residual=None, act=ops.ACT_NONE, **kw):
    batch = a.shape[0] if a.dim() == 3 else (w.shape[0] if w.dim() == 3 else 1)
    mode = "geglu" if act == ops.ACT_GEGLU else ("res" if residual is not None else ("bias" if bias is not None else "plain"))
    key = ("gemm_nt", a.shape[-2], w.shape[-2], a.shape[-1], batch, mode)
    return timed(key, lambda: orig_gemm(a, w, bias=bias, residual=residual, act=act, **kw))


def conv3x3(x, w, B, H, W, **kw):
    key = ("conv3x3", B, H, W, x.shape[-1], w.shape[0], "s2" if kw.get("stride", 1) == 2 else ("up" if kw.get("upsample") else ""))
    return timed(key, lambda: orig_conv(x, w, B, H, W, **kw))


orig_ff = ops.ff_geglu_fused


def ff_geglu_fused(x, w1i, b1i, w2, b2, residual):
    return timed(("gemm_nt", x.shape[0], "ff_geglu_fused", x.shape[1], 1, "fused ff1+ff2"), lambda: orig_ff(x, w1i, b1i, w2, b2, residual))


ops.gemm_nt, ops.conv3x3, ops.ff_geglu_fused = gemm_nt, conv3x3, ff_geglu_fused
ops.USE_FUSED_FF = getattr(ops, "USE_FUSED_FF", True)
unet = UNet2DConditionModel(in_channels=4).init_random(1234, device=dev).to(dev, torch.bfloat16)
gm = UNet2DConditionModel(in_channels=8).init_random(1238, device=dev).to(dev, torch.bfloat16)
g = torch.Generator("cpu").manual_seed(1)
ctx8, ctx4 = unet.prepare_context(torch.randn(8, 77, 768, generator=g).to(dev)), gm.prepare_context(torch.randn(4, 77, 768, generator=g).to(dev))
lat = torch.randn(4, 4, 64, 64, generator=g).to(dev)
for m in (unet, gm):
    m.set_timestep(501)
x8 = unet.pack_input(lat, dup=1)
x4 = gm.pack_input((lat, lat), dup=1)
from gm_diffusion._native import lib
lib().gmd_gemm_plan_family(int(os.environ.get("GMD_ONE_FAMILY", "1")))  # the co-running plan family: what the shipped two-stream pipeline launches
for rep in range(3):
    ACTIVE[0] = rep == 2
    torch.cuda._sleep(int(2e8)) if rep == 2 else None  # host head start: the event pairs must not include waits for the host
    unet.forward_packed(x8, 8, 64, 64, ctx8, cfg_shared=True)
    gm.forward_packed(x4, 4, 64, 64, ctx4)
    torch.cuda.synchronize()
rows = []
for key, evs in rec.items():
    us = [a.elapsed_time(b) * 1e3 for a, b in evs]
    rows.append(dict(kind=key[0], key=list(key[1:]), launches=len(us), total_us=round(sum(us), 1), avg_us=round(sum(us) / len(us), 2)))
tot = {k: sum(r["total_us"] for r in rows if r["kind"] == k) for k in ("gemm_nt", "conv3x3")}
for r in sorted(rows, key=lambda r: -r["total_us"]):
    r["share_of_kind"] = round(r["total_us"] / tot[r["kind"]], 4)
    print(json.dumps(r))
print(json.dumps(dict(totals_us=tot, note="one loop iteration: SDR forward (batch 8, CFG shared prefix) + GM forward (batch 4), eager, single stream, co-running plan family")))

#!/usr/bin/env python
"""Wall-time share of parts of the shipped pipeline by the SKIP method: a selected set of launches is replaced by
an uninitialised output of the right shape, the whole batch (HIP graphs, two streams, BASELINE config 2) is timed, and the
difference to the unmodified run is what removing -- or perfectly fusing away -- those launches could win at most.  Numbers
produced by a skipping run are wrong by construction; timing only.

Selections: --by level  : every conv / GEMM / attention / norm launch of one UNet resolution level (64x64, 32x32, 16x16, 8x8)
            --by kind   : per level and kind (conv, linear, attention, norm)
            --by part   : named parts (feed-forward, cross-attention chain, self-attention, second convolution of the resnets ...)
Usage: wall_share.py [--by level|kind|part] [--steps 3] [--dtype bf16]"""
import argparse, os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gm-diffusion_amd")):
    sys.path.insert(0, p)
import torch

from gm_diffusion import hdr, hip_ops as ops
from gm_diffusion.components import AutoencoderKL, PNDMScheduler, UNet2DConditionModel
from gm_diffusion.components import unet_2d_condition as U
from gm_diffusion.pipelines import StableDiffusionDualUNetPipeline

ap = argparse.ArgumentParser()
ap.add_argument("--by", default="level", choices=["level", "kind", "part", "linear"])
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--batch", type=int, default=4)
a = ap.parse_args()
dev = torch.device("cuda")
dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[a.dtype]
RES, STEPS_INF = 512, 50
LEVEL_TOKENS = {4096: "64x64", 1024: "32x32", 256: "16x16", 64: "8x8"}

unet = UNet2DConditionModel(in_channels=4).init_random(1234, device=dev).to(dev, dtype)
gm_unet = UNet2DConditionModel(in_channels=8).init_random(1238, device=dev).to(dev, dtype)
vae = AutoencoderKL().init_random(1334, device=dev).to(dev, dtype)
pipe = StableDiffusionDualUNetPipeline(vae=vae, text_encoder=None, tokenizer=None, unet=unet, gm_unet=gm_unet,
                                       scheduler=PNDMScheduler(num_train_timesteps=1000, skip_prk_steps=True, set_alpha_to_one=False, beta_start=0.00085,
                                                               beta_end=0.012, beta_schedule="scaled_linear", steps_offset=1),
                                       safety_checker=None, feature_extractor=None, requires_safety_checker=False)
pipe.set_progress_bar_config(disable=True)
ge = torch.Generator("cpu").manual_seed(1)
B, h = a.batch, RES // 8
pos, neg = torch.randn(B, 77, 768, generator=ge).to(dev), torch.randn(B, 77, 768, generator=ge).to(dev)
lat = torch.randn(B, 4, h, h, generator=torch.Generator("cpu").manual_seed(42)).to(dev)

# ---- the skip switch: SKIP(kind, tokens_per_sample, tag) -> bool, consulted by the wrappers below; `tag` names the call site ----
SKIP = lambda kind, tokens, tag: False
TAG = [None]  # set by the patched UNet methods around their launches
real = {n: getattr(ops, n) for n in ("conv3x3", "gemm_nt", "attention", "groupnorm", "layernorm", "ff_geglu_fused")}


def _tokens_of_rows(rows, batch_hint):
    for t in LEVEL_TOKENS:
        if rows % t == 0 and rows // t in (a.batch, 2 * a.batch):  # the GM UNet / CFG prefix run at the batch, the SDR UNet at twice it
            return t
    return None


def conv3x3(x, w, Bn, H, W, **kw):
    up, st = kw.get("upsample", False), kw.get("stride", 1)
    ho, wo = (2 * H, 2 * W) if up else ((H + 2 - 3) // st + 1, (W + 2 - 3) // st + 1)
    if x.shape[-1] >= 320 or w.shape[0] >= 320:  # UNet convolutions (the VAE's are not touched: H*W > 4096 or narrow)
        if ho * wo in LEVEL_TOKENS and SKIP("conv", ho * wo, TAG[0]):
            return torch.empty((Bn, ho * wo, w.shape[0]), dtype=kw.get("out_dtype") or x.dtype, device=x.device), ho, wo
    return real["conv3x3"](x, w, Bn, H, W, **kw)


def gemm_nt(a_, w, **kw):
    if a_.dim() == 2 and w.dim() == 2 and a_.shape[0] >= 64:
        tok = _tokens_of_rows(a_.shape[0], None)
        n_, k_ = w.shape[0], a_.shape[1]
        if kw.get("act") == ops.ACT_GEGLU:
            tag = "ff1"
        elif k_ == 4 * n_:
            tag = "ff2"
        elif n_ == 2 * k_:
            tag = "qk"
        elif n_ == k_:
            tag = "cc_res" if kw.get("residual") is not None else "cc"
        else:
            tag = "other"
        if tok and SKIP("linear", tok, tag):
            n = w.shape[0] // 2 if kw.get("act") == ops.ACT_GEGLU else w.shape[0]
            out = kw.get("out")
            return out if out is not None else torch.empty((a_.shape[0], n), dtype=kw.get("out_dtype") or a_.dtype, device=a_.device)
    elif a_.dim() == 2 and w.dim() == 3 and w.shape[1] in LEVEL_TOKENS:  # V^T = W_v x^T
        if SKIP("linear", w.shape[1], "vt"):
            return torch.empty((w.shape[0], a_.shape[0], kw.get("ldc") or w.shape[1]), dtype=a_.dtype, device=a_.device)
    return real["gemm_nt"](a_, w, **kw)


def attention(q, k, vt, heads, nk, scale, **kw):
    if q.shape[1] in LEVEL_TOKENS and SKIP("attention", q.shape[1], "self" if nk == q.shape[1] else "cross"):
        return torch.empty((q.shape[0], q.shape[1], vt.shape[1]), dtype=q.dtype, device=q.device)
    return real["attention"](q, k, vt, heads, nk, scale, **kw)


def groupnorm(x, Bn, *args, **kw):
    hw = x.numel() // (Bn * x.shape[-1])
    if hw in LEVEL_TOKENS and x.shape[-1] >= 320 and SKIP("norm", hw, "gn"):
        return torch.empty_like(x)
    return real["groupnorm"](x, Bn, *args, **kw)


def layernorm(x, *args, **kw):
    tok = _tokens_of_rows(x.numel() // x.shape[-1], None)
    if tok and SKIP("norm", tok, "ln"):
        return torch.empty_like(x)
    return real["layernorm"](x, *args, **kw)


def ff_geglu_fused(x, *args, **kw):
    tok = _tokens_of_rows(x.shape[0], None)
    if tok and SKIP("linear", tok, "ff"):
        return torch.empty_like(x)
    return real["ff_geglu_fused"](x, *args, **kw)


for n, f in (("conv3x3", conv3x3), ("gemm_nt", gemm_nt), ("attention", attention), ("groupnorm", groupnorm), ("layernorm", layernorm),
             ("ff_geglu_fused", ff_geglu_fused)):
    setattr(ops, n, f)

# call-site tags: which part of a block a launch belongs to (for --by part)
_orig_tr, _orig_res = U.UNet2DConditionModel._transformer, U.UNet2DConditionModel._resnet


def run(label):
    for m in (unet, gm_unet):
        m._graphs = {}
    torch.cuda.synchronize()

    def step():  # the denoising loop only: the decode tail (two VAE decodes + HDR kernel, 28 ms) is constant and not touched
        return pipe(prompt_embeds=pos, negative_prompt_embeds=neg, latents=lat, height=RES, width=RES, num_inference_steps=STEPS_INF,
                    guidance_scale=7.5, output_type="latent")

    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.steps * 1e3
    return ms


base = run("nothing")
print(f"{'removed':64s} {'ms/batch':>9s} {'share of the wall':>18s}")
print(f"{'nothing':64s} {base:9.1f}")
exps = []
if a.by == "level":
    for tok, name in LEVEL_TOKENS.items():
        exps.append((f"every conv / linear / attention / norm launch of level {name}", lambda k, t, tag, tok=tok: t == tok))
elif a.by == "kind":
    for tok, name in LEVEL_TOKENS.items():
        for kind in ("conv", "linear", "attention", "norm"):
            exps.append((f"{name}: {kind}", lambda k, t, tag, tok=tok, kind=kind: t == tok and k == kind))
elif a.by == "linear":
    exps = [("C->C projections with residual (o1, o2, proj_out)", lambda k, t, tag: k == "linear" and tag == "cc_res"),
            ("C->C projections without (q2, proj_in)", lambda k, t, tag: k == "linear" and tag == "cc"),
            ("q|k projection (N = 2C)", lambda k, t, tag: k == "linear" and tag == "qk"),
            ("V^T projection", lambda k, t, tag: k == "linear" and tag == "vt"),
            ("feed-forward (ff1 GEGLU, ff2, fused)", lambda k, t, tag: k == "linear" and tag in ("ff1", "ff2", "ff")),
            ("all LayerNorms", lambda k, t, tag: k == "norm" and tag == "ln"),
            ("64x64: C->C projections, both kinds", lambda k, t, tag: k == "linear" and tag in ("cc", "cc_res") and t == 4096),
            ("32x32: C->C projections, both kinds", lambda k, t, tag: k == "linear" and tag in ("cc", "cc_res") and t == 1024),
            ("16x16: C->C projections, both kinds", lambda k, t, tag: k == "linear" and tag in ("cc", "cc_res") and t == 256),
            ("64x64: everything row-local after self-attention (o1..proj_out, LN2/3, cross-attention, FF)",
             lambda k, t, tag: t == 4096 and ((k == "linear" and tag in ("cc", "cc_res", "ff1", "ff2", "ff")) or (k == "attention" and tag == "cross"))),
            ]
else:
    exps = [("self-attention kernels (all levels)", lambda k, t, tag: k == "attention" and tag == "self"),
            ("cross-attention kernels (all levels)", lambda k, t, tag: k == "attention" and tag == "cross"),
            ("all LayerNorms", lambda k, t, tag: k == "norm" and tag == "ln"),
            ("all GroupNorms", lambda k, t, tag: k == "norm" and tag == "gn"),
            ("all convolutions", lambda k, t, tag: k == "conv"),
            ("all linear layers (projections, feed-forward, V^T)", lambda k, t, tag: k == "linear")]
for name, pred in exps:
    SKIP = pred
    ms = run(name)
    print(f"{name:64s} {ms:9.1f} {100 * (base - ms) / base:17.1f}%", flush=True)
SKIP = lambda kind, tokens, tag: False
again = run("nothing")
print(f"{'nothing (again)':64s} {again:9.1f}")

#!/usr/bin/env python
"""Precision experiment for BASELINE.json configs[4] ("fp8 MFMA attention"): what rounding the attention operands to OCP fp8
(e4m3fn, gfx950's MFMA fp8 type) costs against bfloat16 / float16 operands, on the host in float64 arithmetic (only the operand
rounding is emulated, products and sums are exact -- the matrix core accumulates in float32 either way).  Shapes: one SDXL-like
head (d = 64), N = 1024 / 4096 keys, Gaussian q / k / v; per-tensor power-of-two scales bring each fp8 operand's maximum to 240
(e4m3 tops out at 448).  CPU only, no GPU needed.  Usage: fp8_attention_precision.py"""
import torch

torch.manual_seed(0)


def rnd(x, kind):
    if kind == "f64":
        return x
    if kind == "bf16":
        return x.to(torch.bfloat16).double()
    if kind == "f16":
        return x.to(torch.float16).double()
    s = 2.0 ** torch.floor(torch.log2(240.0 / x.abs().max()))  # per-tensor power-of-two scale
    return (x * s).float().to(torch.float8_e4m3fn).double() / s


def attn(q, k, v, kind, p_kind=None):
    s = rnd(q, kind) @ rnd(k, kind).t() / q.shape[1] ** 0.5
    p = torch.exp(s - s.max(-1, keepdim=True).values)  # unnormalised probabilities in (0, 1], as the flash kernels carry them
    pq = rnd(p, p_kind or kind)
    return (pq @ rnd(v, kind)) / pq.sum(-1, keepdim=True)


rel = lambda a, b: float((a - b).norm() / b.norm())
print(f"{'N':>6s} {'logit scale':>11s} | relative error of the attention output against float64 operands")
for N in (1024, 4096):
    for temp in (1.0, 3.0):  # temp 3: sharper rows (a few keys dominate)
        q, k, v = torch.randn(256, 64, dtype=torch.float64) * temp, torch.randn(N, 64, dtype=torch.float64), torch.randn(N, 64, dtype=torch.float64)
        ref = attn(q, k, v, "f64")
        row = []
        for kind, pk in (("f16", None), ("bf16", None), ("fp8", None), ("fp8", "bf16"), ("bf16", "fp8")):
            row.append(f"{kind}{'/P ' + pk if pk else ''}: {rel(attn(q, k, v, kind, pk), ref):.2e}")
        print(f"{N:6d} {temp:11.1f} | " + "   ".join(row))

#!/usr/bin/env python
"""Where does the bf16 drift of the UNet come from?  CPU experiment on the oracle UNet (SD-1.5 widths) with the rounding
points of the HIP bf16 path emulated by hooks:
  operands : inputs and weights of every conv / linear rounded to bf16 (what the MFMA consumes), attention P rounded
  stores   : every activation written to HBM rounded to bf16 (norm outputs, projections, attention outputs, GEGLU)
  residual : the residual stream itself -- ResnetBlock2D `x + h`, the three transformer adds, `proj_out(h) + x`, the
             conv_shortcut output -- rounded to bf16 after every add (mode "bf16") or kept in float32 (mode "f32res")
Prints the relative error of eps against the float32 oracle for one evaluation and the latent RMS drift of a short
single-UNet PNDM loop.  Test / design infrastructure only (imports oracle/)."""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn as nn
import torch.nn.functional as F
from oracle import fixtures, unet as OU, pipelines as OP, schedulers as OS

HALF = torch.float16 if os.environ.get("DRIFT_HALF") == "f16" else torch.bfloat16
r16 = lambda t: t.to(HALF).float()


def emulate(model, residual_f32):
    m = fixtures.build_unet("sd15", 4)
    m.load_state_dict(model.state_dict())
    rr = (lambda t: t) if residual_f32 else r16
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, (nn.Conv2d, nn.Linear)):
                mod.weight.copy_(r16(mod.weight))
    feeds_residual = set()
    for mod in m.modules():
        if isinstance(mod, OU.ResnetBlock2D):
            feeds_residual.add(mod.conv2)
            if residual_f32 and mod.conv_shortcut is not None:
                feeds_residual.add(mod.conv_shortcut)  # written as float32 residual
        if isinstance(mod, OU.Attention):
            feeds_residual.add(mod.to_out[0])
        if isinstance(mod, OU.FeedForward):
            feeds_residual.add(mod.net[2])
        if isinstance(mod, OU.Transformer2DModel):
            feeds_residual.add(mod.proj_out)
            if residual_f32:
                feeds_residual.add(mod.proj_in)  # starts the transformer's float32 stream
    feeds_residual.add(m.conv_out)  # eps leaves from the float32 accumulators
    for mod in m.modules():
        if isinstance(mod, (nn.Conv2d, nn.Linear)):
            mod.register_forward_pre_hook(lambda mo, inp: tuple(r16(i) for i in inp))
            if mod not in feeds_residual:
                mod.register_forward_hook(lambda mo, inp, out: r16(out))
        elif isinstance(mod, (nn.GroupNorm, nn.LayerNorm)):
            mod.register_forward_hook(lambda mo, inp, out: r16(out))

    def res_fwd(self, x, temb=None):
        h = self.conv1(r16(F.silu(self.norm1(x))))
        # (conv1 + time embedding: one epilogue, one rounding -- conv1's hook rounded before the add; close enough)
        if self.time_emb_proj is not None:
            h = r16(h + self.time_emb_proj(F.silu(temb))[:, :, None, None])
        h = self.conv2(r16(F.silu(self.norm2(h))))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return rr(x + h)

    def attn_fwd(self, x, context=None):
        ctx = x if context is None else context
        B, N, C = x.shape
        H = self.heads
        q = self.to_q(x).view(B, N, H, C // H).transpose(1, 2)
        k = self.to_k(ctx).view(B, ctx.shape[1], H, C // H).transpose(1, 2)
        v = self.to_v(ctx).view(B, ctx.shape[1], H, C // H).transpose(1, 2)
        s = torch.matmul(q, k.transpose(-1, -2)) * (C // H) ** -0.5
        p = torch.softmax(s, dim=-1)
        o = r16(torch.matmul(r16(p), v) )
        return self.to_out[0](o.transpose(1, 2).reshape(B, N, C))

    def geglu_fwd(self, x):
        h, gate = F.linear(r16(x), self.proj.weight, self.proj.bias).chunk(2, dim=-1)  # fp32 accumulators into the gate
        return r16(h * F.gelu(gate))

    def blk_fwd(self, x, context):
        x = rr(x + self.attn1(self.norm1(x)))
        x = rr(x + self.attn2(self.norm2(x), context))
        return rr(x + self.ff(self.norm3(x)))

    def t2d_fwd(self, x, context):
        B, C, H, W = x.shape
        h = self.proj_in(self.norm(x))
        h = rr(h).permute(0, 2, 3, 1).reshape(B, H * W, C)
        for blk in self.transformer_blocks:
            h = blk(h, context)
        h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
        return rr(self.proj_out(h) + x)

    for mod in m.modules():
        if isinstance(mod, OU.ResnetBlock2D): mod.forward = types.MethodType(res_fwd, mod)
        elif isinstance(mod, OU.Attention): mod.forward = types.MethodType(attn_fwd, mod)
        elif isinstance(mod, OU.GEGLU): mod.forward = types.MethodType(geglu_fwd, mod)
        elif isinstance(mod, OU.BasicTransformerBlock): mod.forward = types.MethodType(blk_fwd, mod)
        elif isinstance(mod, OU.Transformer2DModel): mod.forward = types.MethodType(t2d_fwd, mod)
    return m


def main():
    hw = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    torch.set_num_threads(8)
    ref = fixtures.build_unet("sd15", 4)
    models = {"bf16 (residual stream rounded)": emulate(ref, False), "bf16 operands, float32 residual stream": emulate(ref, True)}
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 4, hw, hw, generator=g)
    ctx = torch.randn(2, 77, 768, generator=g)
    rel = lambda a, b: float((a - b).norm() / b.norm())
    with torch.no_grad():
        for t in (981, 501, 21):
            e0 = ref(x, torch.tensor(t), encoder_hidden_states=ctx)[0]
            print(f"t={t:4d} " + "  ".join(f"{k}: rel err {rel(m(x, torch.tensor(t), encoder_hidden_states=ctx)[0], e0):.3e}" for k, m in models.items()))
        if steps:
            pe, ne, lat = fixtures.make_inputs(1, hw, hw)

            class One:  # single 4-channel UNet through the GM loop shape: sdr_latent is ignored by slicing
                def __init__(s, m): s.m = m
                def __call__(s, xx, t, encoder_hidden_states=None, return_dict=False): return s.m(xx[:, 4:], t, encoder_hidden_states=encoder_hidden_states)
            z = torch.zeros(1, 4, hw, hw)
            rec0 = []
            OP.gm_loop(One(ref), OS.PNDMScheduler(), z, pe, ne, lat, steps, guidance_scale=7.5, record=rec0)
            rms = lambda a, b: float(((a - b) ** 2).mean().sqrt())
            for k, m in models.items():
                rec = []
                OP.gm_loop(One(m), OS.PNDMScheduler(), z, pe, ne, lat, steps, guidance_scale=7.5, record=rec)
                print(f"{steps} PNDM steps, {k}: latent RMS drift per step " + " ".join(f"{rms(a, b):.1e}" for a, b in zip(rec, rec0)))


if __name__ == "__main__":
    main()

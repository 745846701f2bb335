"""
Oracle: pure-torch fp32 restatement of diffusers' ``UNet2DConditionModel`` for the
SD-1.5 layout the reference hard-codes (scripts/inference/generate_hdr.py:116-135).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

diffusers (>=0.33, README.md:54) is an un-vendored dependency that is not
installed here: this file restates its published architecture from knowledge of
diffusers 0.33.  PARITY UNPINNED against real
diffusers; pinned by known-answer tests against torch primitives only.
Parameter names follow diffusers so a real checkpoint's state-dict loads.

Call sites in the reference: stable_diffusion_gm.py:1051-1059,
stable_diffusion_dual_unet.py:1052-1060, 1083-1092:
    unet(sample, t, encoder_hidden_states=..., return_dict=False)[0]
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

SD15_UNET_CONFIG = dict(
    act_fn="silu",
    attention_head_dim=8,  # legacy name: this is the number of heads
    block_out_channels=[320, 640, 1280, 1280],
    center_input_sample=False,
    cross_attention_dim=768,
    down_block_types=["CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"],
    downsample_padding=1,
    flip_sin_to_cos=True,
    freq_shift=0,
    in_channels=4,
    layers_per_block=2,
    mid_block_scale_factor=1,
    norm_eps=1e-05,
    norm_num_groups=32,
    out_channels=4,
    sample_size=64,
    up_block_types=["UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"],
    time_cond_proj_dim=None,
    # SDXL-style options (BASELINE config 5; no reference behaviour exists for them -- SURVEY.md 7 "SDXL / fp8"): all off for SD-1.5
    transformer_layers_per_block=1,   # int, or one depth per down block (the up blocks use the reversed list, the mid block the last)
    use_linear_projection=False,      # Transformer2DModel proj_in / proj_out as nn.Linear on tokens instead of conv1x1
    addition_embed_type=None,         # "text_time": micro-conditioning (time_ids) + pooled text embedding added to the time embedding
    addition_time_embed_dim=None,
    projection_class_embeddings_input_dim=None,
)

# diffusers' config of stabilityai/stable-diffusion-xl-base-1.0 (unet/config.json), restated from knowledge of diffusers 0.33
# [3P-memory]; pinned here only by its published parameter count, 2,567,463,684 (tests/test_oracle_models.py)
SDXL_UNET_CONFIG = dict(
    block_out_channels=[320, 640, 1280], down_block_types=["DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"],
    up_block_types=["CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"], transformer_layers_per_block=[1, 2, 10],
    attention_head_dim=[5, 10, 20], cross_attention_dim=2048, use_linear_projection=True, addition_embed_type="text_time",
    addition_time_embed_dim=256, projection_class_embeddings_input_dim=2816, sample_size=128,
)


def timestep_embedding(timesteps, dim, flip_sin_to_cos=True, freq_shift=0.0, max_period=10000):
    """diffusers ``get_timestep_embedding``: exp(-ln(max_period) * i / (half - shift)),
    [sin, cos] then flipped to [cos, sin] when flip_sin_to_cos."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32, device=timesteps.device)
    exponent = exponent / (half - freq_shift)
    emb = torch.exp(exponent)
    emb = timesteps[:, None].float() * emb[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


class TimestepEmbedding(nn.Module):
    def __init__(self, in_dim, dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_dim, dim)
        self.linear_2 = nn.Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb_dim, groups=32, eps=1e-5):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_dim, cout) if temb_dim else None
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x, temb=None):
        h = self.conv1(F.silu(self.norm1(x)))
        if self.time_emb_proj is not None:
            h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h  # output_scale_factor == 1


class Attention(nn.Module):
    """diffusers ``Attention`` with the default processor: q/k/v without bias,
    out-proj with bias, ``heads`` heads of ``dim // heads``, scale d^-0.5."""

    def __init__(self, dim, heads, cross_dim=None, qkv_bias=False):
        super().__init__()
        self.heads = heads
        self.to_q = nn.Linear(dim, dim, bias=qkv_bias)
        self.to_k = nn.Linear(cross_dim or dim, dim, bias=qkv_bias)
        self.to_v = nn.Linear(cross_dim or dim, dim, bias=qkv_bias)
        self.to_out = nn.ModuleList([nn.Linear(dim, dim)])

    def forward(self, x, context=None):
        ctx = x if context is None else context
        B, N, C = x.shape
        H = self.heads
        q = self.to_q(x).view(B, N, H, C // H).transpose(1, 2)
        k = self.to_k(ctx).view(B, ctx.shape[1], H, C // H).transpose(1, 2)
        v = self.to_v(ctx).view(B, ctx.shape[1], H, C // H).transpose(1, 2)
        s = torch.matmul(q, k.transpose(-1, -2)) * (C // H) ** -0.5
        o = torch.matmul(torch.softmax(s, dim=-1), v)
        return self.to_out[0](o.transpose(1, 2).reshape(B, N, C))


class GEGLU(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, inner * 2)

    def forward(self, x):
        h, gate = self.proj(x).chunk(2, dim=-1)
        return h * F.gelu(gate)  # erf gelu


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * 4), nn.Identity(), nn.Linear(dim * 4, dim)])

    def forward(self, x):
        return self.net[2](self.net[0](x))


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, cross_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, heads)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, heads, cross_dim)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    def forward(self, x, context):
        x = x + self.attn1(self.norm1(x))
        x = x + self.attn2(self.norm2(x), context)
        return x + self.ff(self.norm3(x))


class Transformer2DModel(nn.Module):
    """SD-1.5 flavour: conv1x1 projections (use_linear_projection=False), one block, GN eps 1e-6.  SDXL flavour: nn.Linear
    projections applied on tokens, ``depth`` BasicTransformerBlocks."""

    def __init__(self, dim, heads, cross_dim, groups=32, depth=1, linear=False):
        super().__init__()
        self.linear = linear
        self.norm = nn.GroupNorm(groups, dim, eps=1e-6)
        self.proj_in = nn.Linear(dim, dim) if linear else nn.Conv2d(dim, dim, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(dim, heads, cross_dim) for _ in range(depth)])
        self.proj_out = nn.Linear(dim, dim) if linear else nn.Conv2d(dim, dim, 1)

    def forward(self, x, context):
        B, C, H, W = x.shape
        h = self.norm(x)
        if self.linear:
            h = self.proj_in(h.permute(0, 2, 3, 1).reshape(B, H * W, C))
        else:
            h = self.proj_in(h).permute(0, 2, 3, 1).reshape(B, H * W, C)
        for blk in self.transformer_blocks:
            h = blk(h, context)
        if self.linear:
            h = self.proj_out(h).reshape(B, H, W, C).permute(0, 3, 1, 2)
        else:
            h = self.proj_out(h.reshape(B, H, W, C).permute(0, 3, 1, 2))
        return h + x


class Downsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=1)

    def forward(self, x):
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class DownBlock(nn.Module):
    def __init__(self, cin, cout, temb, n, heads, cross_dim, attn, down, groups, eps, depth=1, linear=False):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, temb, groups, eps) for i in range(n)])
        if attn:
            self.attentions = nn.ModuleList([Transformer2DModel(cout, heads, cross_dim, groups, depth, linear) for _ in range(n)])
        self.has_attn = attn
        if down:
            self.downsamplers = nn.ModuleList([Downsample2D(cout)])
        self.has_down = down

    def forward(self, x, temb, ctx):
        outs = []
        for i, r in enumerate(self.resnets):
            x = r(x, temb)
            if self.has_attn:
                x = self.attentions[i](x, ctx)
            outs.append(x)
        if self.has_down:
            x = self.downsamplers[0](x)
            outs.append(x)
        return x, outs


class MidBlock(nn.Module):
    def __init__(self, c, temb, heads, cross_dim, groups, eps, depth=1, linear=False):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, temb, groups, eps), ResnetBlock2D(c, c, temb, groups, eps)])
        self.attentions = nn.ModuleList([Transformer2DModel(c, heads, cross_dim, groups, depth, linear)])

    def forward(self, x, temb, ctx):
        x = self.resnets[0](x, temb)
        x = self.attentions[0](x, ctx)
        return self.resnets[1](x, temb)


class UpBlock(nn.Module):
    def __init__(self, cin, cout, cprev, temb, n, heads, cross_dim, attn, up, groups, eps, depth=1, linear=False):
        super().__init__()
        res = []
        for i in range(n):
            skip = cin if i == n - 1 else cout
            rin = cprev if i == 0 else cout
            res.append(ResnetBlock2D(rin + skip, cout, temb, groups, eps))
        self.resnets = nn.ModuleList(res)
        if attn:
            self.attentions = nn.ModuleList([Transformer2DModel(cout, heads, cross_dim, groups, depth, linear) for _ in range(n)])
        self.has_attn = attn
        if up:
            self.upsamplers = nn.ModuleList([Upsample2D(cout)])
        self.has_up = up

    def forward(self, x, skips, temb, ctx):
        for i, r in enumerate(self.resnets):
            x = r(torch.cat([x, skips.pop()], dim=1), temb)
            if self.has_attn:
                x = self.attentions[i](x, ctx)
        if self.has_up:
            x = self.upsamplers[0](x)
        return x


class UNet2DConditionModel(nn.Module):
    def __init__(self, **overrides):
        super().__init__()
        cfg = dict(SD15_UNET_CONFIG)
        cfg.update(overrides)
        self.config = SimpleNamespace(**cfg)
        ch = cfg["block_out_channels"]
        L = len(ch)
        as_list = lambda v: list(v) if isinstance(v, (list, tuple)) else [v] * L
        heads = as_list(cfg["attention_head_dim"])           # legacy name: heads per level
        depth = as_list(cfg["transformer_layers_per_block"])
        linear = bool(cfg["use_linear_projection"])
        cross = cfg["cross_attention_dim"]
        groups, eps, n = cfg["norm_num_groups"], cfg["norm_eps"], cfg["layers_per_block"]
        temb = ch[0] * 4
        self.conv_in = nn.Conv2d(cfg["in_channels"], ch[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(ch[0], temb)
        if cfg["addition_embed_type"] == "text_time":
            self.add_embedding = TimestepEmbedding(cfg["projection_class_embeddings_input_dim"], temb)
        elif cfg["addition_embed_type"] is not None:
            raise NotImplementedError(cfg["addition_embed_type"])
        downs, cout = [], ch[0]
        for i, t in enumerate(cfg["down_block_types"]):
            cin, cout = cout, ch[i]
            downs.append(DownBlock(cin, cout, temb, n, heads[i], cross, t.startswith("CrossAttn"), i != L - 1, groups, eps, depth[i], linear))
        self.down_blocks = nn.ModuleList(downs)
        self.mid_block = MidBlock(ch[-1], temb, heads[-1], cross, groups, eps, depth[-1], linear)
        rev, rheads, rdepth = list(reversed(ch)), list(reversed(heads)), list(reversed(depth))
        ups, cout = [], rev[0]
        for i, t in enumerate(cfg["up_block_types"]):
            cprev, cout = cout, rev[i]
            cin = rev[min(i + 1, L - 1)]
            ups.append(UpBlock(cin, cout, cprev, temb, n + 1, rheads[i], cross, t.startswith("CrossAttn"), i != L - 1, groups, eps, rdepth[i], linear))
        self.up_blocks = nn.ModuleList(ups)
        self.conv_norm_out = nn.GroupNorm(groups, ch[0], eps=eps)
        self.conv_out = nn.Conv2d(ch[0], cfg["out_channels"], 3, padding=1)

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    @property
    def device(self):
        return self.conv_in.weight.device

    def forward(self, sample, timestep, encoder_hidden_states=None, timestep_cond=None,
                cross_attention_kwargs=None, added_cond_kwargs=None, return_dict=False):
        if not torch.is_tensor(timestep):
            timestep = torch.tensor([timestep], dtype=torch.float32, device=sample.device)
        t = timestep.reshape(-1).to(sample.device).expand(sample.shape[0])
        temb = timestep_embedding(t, self.config.block_out_channels[0], self.config.flip_sin_to_cos, self.config.freq_shift)
        temb = self.time_embedding(temb.to(sample.dtype))
        if self.config.addition_embed_type == "text_time":
            # diffusers get_aug_embed: sinusoid of every micro-conditioning scalar (time_ids [B, 6] -> [B, 6 * dim]) concatenated
            # BEHIND the pooled text embedding, through add_embedding, added to the time embedding
            text_embeds, time_ids = added_cond_kwargs["text_embeds"], added_cond_kwargs["time_ids"]
            te = timestep_embedding(time_ids.flatten(), self.config.addition_time_embed_dim, self.config.flip_sin_to_cos, self.config.freq_shift)
            te = te.reshape(text_embeds.shape[0], -1)
            temb = temb + self.add_embedding(torch.cat([text_embeds, te], dim=-1).to(sample.dtype))
        x = self.conv_in(sample)
        skips = [x]
        for blk in self.down_blocks:
            x, outs = blk(x, temb, encoder_hidden_states)
            skips.extend(outs)
        x = self.mid_block(x, temb, encoder_hidden_states)
        for blk in self.up_blocks:
            x = blk(x, skips, temb, encoder_hidden_states)
        x = self.conv_out(F.silu(self.conv_norm_out(x)))
        return (x,)


def tiny_unet_config(in_channels=4):
    """A structurally complete but small UNet (same block types, 2 levels fewer
    channels) used for pipeline parity tests that must finish in seconds on CPU."""
    return dict(in_channels=in_channels, block_out_channels=[64, 128, 128, 128], cross_attention_dim=64,
                attention_head_dim=2, norm_num_groups=8, sample_size=8)


def tiny_sdxl_unet_config(in_channels=4):
    """A small UNet with every SDXL-style feature on (three levels, no attention at the first, transformer depths 1 / 2 / 3,
    head dim 64, linear projections, text_time conditioning)."""
    return dict(in_channels=in_channels, block_out_channels=[64, 128, 256], down_block_types=["DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"],
                up_block_types=["CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"], transformer_layers_per_block=[1, 2, 3],
                attention_head_dim=[1, 2, 4], cross_attention_dim=128, use_linear_projection=True, addition_embed_type="text_time",
                addition_time_embed_dim=32, projection_class_embeddings_input_dim=80 + 6 * 32, norm_num_groups=8, sample_size=16)

"""
Oracle: pure-torch fp32 restatement of diffusers' ``AutoencoderKL`` (SD-1.5 VAE).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED against real
diffusers (not installed); architecture from SURVEY.md Appendix A.1.

Reference call sites: ``vae.decode(latents / vae.config.scaling_factor,
return_dict=False)[0]`` -- stable_diffusion_gm.py:1093-1096,
scripts/inference/generate_hdr.py:225-233; ``vae.encode(x).latent_dist.sample()``
-- generate_hdr.py:208.
"""
from __future__ import annotations

from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

from .unet import ResnetBlock2D, Upsample2D

SD15_VAE_CONFIG = dict(
    in_channels=3, out_channels=3, latent_channels=4, block_out_channels=[128, 256, 512, 512],
    layers_per_block=2, norm_num_groups=32, scaling_factor=0.18215, sample_size=512,
)


class VaeAttention(nn.Module):
    """Single-head attention over h*w tokens with GroupNorm, biased projections and a
    residual connection (diffusers ``Attention(..., residual_connection=True)``)."""

    def __init__(self, c, groups):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, c, eps=1e-6)
        self.to_q = nn.Linear(c, c)
        self.to_k = nn.Linear(c, c)
        self.to_v = nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c)])

    def forward(self, x):
        B, C, H, W = x.shape
        h = self.group_norm(x).view(B, C, H * W).transpose(1, 2)
        q, k, v = self.to_q(h), self.to_k(h), self.to_v(h)
        s = torch.matmul(q, k.transpose(-1, -2)) * C ** -0.5
        o = self.to_out[0](torch.matmul(torch.softmax(s, dim=-1), v))
        return o.transpose(1, 2).reshape(B, C, H, W) + x


class VaeMidBlock(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, None, groups, 1e-6), ResnetBlock2D(c, c, None, groups, 1e-6)])
        self.attentions = nn.ModuleList([VaeAttention(c, groups)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class UpDecoderBlock2D(nn.Module):
    def __init__(self, cin, cout, n, up, groups):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, None, groups, 1e-6) for i in range(n)])
        self.has_up = up
        if up:
            self.upsamplers = nn.ModuleList([Upsample2D(cout)])

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        return self.upsamplers[0](x) if self.has_up else x


class Decoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        ch, g = cfg["block_out_channels"], cfg["norm_num_groups"]
        rev = list(reversed(ch))
        self.conv_in = nn.Conv2d(cfg["latent_channels"], rev[0], 3, padding=1)
        self.mid_block = VaeMidBlock(rev[0], g)
        ups, cout = [], rev[0]
        for i in range(len(rev)):
            cin, cout = cout, rev[i]
            ups.append(UpDecoderBlock2D(cin, cout, cfg["layers_per_block"] + 1, i != len(rev) - 1, g))
        self.up_blocks = nn.ModuleList(ups)
        self.conv_norm_out = nn.GroupNorm(g, ch[0], eps=1e-6)
        self.conv_out = nn.Conv2d(ch[0], cfg["out_channels"], 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class DownEncoderBlock2D(nn.Module):
    def __init__(self, cin, cout, n, down, groups):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, None, groups, 1e-6) for i in range(n)])
        self.has_down = down
        if down:
            # diffusers Downsample2D(padding=0): F.pad (0,1,0,1) then conv stride 2 pad 0
            self.downsamplers = nn.ModuleList([nn.Module()])
            self.downsamplers[0].conv = nn.Conv2d(cout, cout, 3, stride=2, padding=0)

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.has_down:
            x = self.downsamplers[0].conv(F.pad(x, (0, 1, 0, 1)))
        return x


class Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        ch, g = cfg["block_out_channels"], cfg["norm_num_groups"]
        self.conv_in = nn.Conv2d(cfg["in_channels"], ch[0], 3, padding=1)
        downs, cout = [], ch[0]
        for i in range(len(ch)):
            cin, cout = cout, ch[i]
            downs.append(DownEncoderBlock2D(cin, cout, cfg["layers_per_block"], i != len(ch) - 1, g))
        self.down_blocks = nn.ModuleList(downs)
        self.mid_block = VaeMidBlock(ch[-1], g)
        self.conv_norm_out = nn.GroupNorm(g, ch[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(ch[-1], 2 * cfg["latent_channels"], 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(self.mid_block(x))))


class DiagonalGaussian:
    """diffusers ``DiagonalGaussianDistribution``: logvar clamped to [-30, 20]."""

    def __init__(self, params):
        self.mean, logvar = params.chunk(2, dim=1)
        self.logvar = logvar.clamp(-30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator=None):
        noise = torch.randn(self.mean.shape, generator=generator, dtype=self.mean.dtype, device=self.mean.device)
        return self.mean + self.std * noise

    def mode(self):
        return self.mean


class AutoencoderKL(nn.Module):
    def __init__(self, with_encoder=False, **overrides):
        super().__init__()
        cfg = dict(SD15_VAE_CONFIG)
        cfg.update(overrides)
        self.config = SimpleNamespace(**cfg)
        lc = cfg["latent_channels"]
        self.decoder = Decoder(cfg)
        self.post_quant_conv = nn.Conv2d(lc, lc, 1)
        if with_encoder:
            self.encoder = Encoder(cfg)
            self.quant_conv = nn.Conv2d(2 * lc, 2 * lc, 1)

    @property
    def dtype(self):
        return self.post_quant_conv.weight.dtype

    def decode(self, z, return_dict=False, generator=None):
        return (self.decoder(self.post_quant_conv(z)),)

    def encode(self, x):
        return SimpleNamespace(latent_dist=DiagonalGaussian(self.quant_conv(self.encoder(x))))


def tiny_vae_config():
    return dict(block_out_channels=[64, 64, 128, 128], norm_num_groups=8)

"""
``AutoencoderKL`` (SD-1.5 VAE) for the MI355X build.  Reference call sites:
``vae.decode(latents / vae.config.scaling_factor, return_dict=False)[0]`` --
gm_diffusion/pipelines/stable_diffusion_gm.py:1093-1096, scripts/inference/generate_hdr.py:225-233;
``vae.encode(x).latent_dist.sample()`` -- generate_hdr.py:208.  In the reference this is
``diffusers.AutoencoderKL`` (un-vendored); here the decoder and encoder run on the HIP kernels
of libgmd_hip.so over channels-last activations (conv3x3 implicit GEMM, GroupNorm+SiLU, the
single-head 512-d mid-block attention as GEMM -> row softmax -> GEMM).  No CPU path.

``decode`` returns float32 NCHW like diffusers; ``decode_nhwc`` additionally exposes the
channels-last [B, H*W, 4] float32 image the fused HDR tail kernel consumes without a transpose.
"""
from __future__ import annotations

import os
from types import SimpleNamespace

import torch

from .. import hip_ops as ops
from .._native import HipExtensionError
from .configuration import read_state_dict
from .unet_2d_condition import _HipModule, _in_own_f32_mode, _pad_to, composed_attention

SD15_VAE_DEFAULTS = dict(
    in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
    norm_num_groups=32, scaling_factor=0.18215, sample_size=512, act_fn="silu",
    down_block_types=("DownEncoderBlock2D",) * 4, up_block_types=("UpDecoderBlock2D",) * 4,
)


class DiagonalGaussianDistribution:
    """diffusers ``DiagonalGaussianDistribution`` over float32 NCHW parameters (logvar clamped to [-30, 20]).
    ``sample`` draws the noise with torch's generator (plumbing) and combines it on the host side of the
    boundary exactly like diffusers: mean + std * noise."""

    def __init__(self, mean, logvar):
        self.mean = mean
        self.logvar = logvar.clamp(-30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator=None):
        from .image_processor import randn_tensor

        noise = randn_tensor(self.mean.shape, generator=generator, device=self.mean.device, dtype=self.mean.dtype)
        return self.mean + self.std * noise

    def mode(self):
        return self.mean


class AutoencoderKL(_HipModule):
    config_name = "config.json"
    _defaults = SD15_VAE_DEFAULTS

    def __init__(self, **config):
        cfg = dict(SD15_VAE_DEFAULTS)
        cfg.update({k: v for k, v in config.items() if k in cfg})
        self.register_to_config(**cfg)
        self._init_module()
        self.with_encoder = True

    # ---- structure ---------------------------------------------------------------------------
    def expected_keys(self, encoder=None):
        c = self.config
        ch = list(c.block_out_channels)
        lc = c.latent_channels
        keys = {}

        def conv(k, co, ci, ks):
            keys[k + ".weight"], keys[k + ".bias"] = (co, ci, ks, ks), (co,)

        def norm(k, n):
            keys[k + ".weight"], keys[k + ".bias"] = (n,), (n,)

        def resnet(k, ci, co):
            norm(k + ".norm1", ci); conv(k + ".conv1", co, ci, 3); norm(k + ".norm2", co); conv(k + ".conv2", co, co, 3)
            if ci != co:
                conv(k + ".conv_shortcut", co, ci, 1)

        def mid(k, cm):
            resnet(k + ".resnets.0", cm, cm); resnet(k + ".resnets.1", cm, cm)
            a = k + ".attentions.0"
            norm(a + ".group_norm", cm)
            for nm in ("to_q", "to_k", "to_v", "to_out.0"):
                keys[f"{a}.{nm}.weight"], keys[f"{a}.{nm}.bias"] = (cm, cm), (cm,)

        rev = list(reversed(ch))
        conv("post_quant_conv", lc, lc, 1)
        conv("decoder.conv_in", rev[0], lc, 3)
        mid("decoder.mid_block", rev[0])
        cout = rev[0]
        for i in range(len(rev)):
            cin, cout = cout, rev[i]
            for j in range(c.layers_per_block + 1):
                resnet(f"decoder.up_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout)
            if i != len(rev) - 1:
                conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", cout, cout, 3)
        norm("decoder.conv_norm_out", ch[0]); conv("decoder.conv_out", c.out_channels, ch[0], 3)
        if self.with_encoder if encoder is None else encoder:
            conv("quant_conv", 2 * lc, 2 * lc, 1)
            conv("encoder.conv_in", ch[0], c.in_channels, 3)
            cout = ch[0]
            for i in range(len(ch)):
                cin, cout = cout, ch[i]
                for j in range(c.layers_per_block):
                    resnet(f"encoder.down_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout)
                if i != len(ch) - 1:
                    conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", cout, cout, 3)
            mid("encoder.mid_block", ch[-1])
            norm("encoder.conv_norm_out", ch[-1]); conv("encoder.conv_out", 2 * lc, ch[-1], 3)
        return keys

    def load_state_dict(self, sd, strict=True):
        self.with_encoder = any(k.startswith("encoder.") for k in sd)
        return super().load_state_dict(sd, strict)

    @classmethod
    def from_pretrained(cls, path, subfolder=None, torch_dtype=None, **overrides):
        d = os.path.join(path, subfolder) if subfolder else path
        cfg = cls.load_config(d)
        cfg.update(overrides)
        m = cls(**cfg)
        m.load_state_dict(read_state_dict(d))
        if torch_dtype is not None:
            m.to(torch_dtype)
        return m

    def init_random(self, seed=1334, with_encoder=False, device=None):
        self.with_encoder = with_encoder
        g = torch.Generator(device or "cpu").manual_seed(seed)
        keys = self.expected_keys()
        sd = {}
        for k, shp in keys.items():
            if "norm" in k.rsplit(".", 2)[-2]:
                sd[k] = torch.ones(shp) if k.endswith("weight") else torch.zeros(shp)
                continue
            wshape = keys[k.rsplit(".", 1)[0] + ".weight"]
            fan_in = 1
            for s_ in wshape[1:]:
                fan_in *= s_
            sd[k] = (torch.rand(shp, generator=g, device=g.device) * 2 - 1) * fan_in ** -0.5
        return self.load_state_dict(sd)

    # ---- weights -----------------------------------------------------------------------------
    def _prepare(self):
        c = self.config
        kmul = self._kmul()
        w = {}
        self._lc_pad = _pad_to(c.latent_channels, kmul)

        def resnet(k):
            r = dict(n1=self._norm(k + ".norm1"), c1=self._conv3(k + ".conv1"), n2=self._norm(k + ".norm2"), c2=self._conv3(k + ".conv2"))
            if k + ".conv_shortcut.weight" in self._raw:
                r["sc"] = self._lin(k + ".conv_shortcut")
            return r

        def mid(k):
            a = k + ".attentions.0"
            wq, bq = self._raw[a + ".to_q.weight"], self._raw[a + ".to_q.bias"]
            wk, bk = self._raw[a + ".to_k.weight"], self._raw[a + ".to_k.bias"]
            wv, bv = self._raw[a + ".to_v.weight"], self._raw[a + ".to_v.bias"]
            wo, bo = self._raw[a + ".to_out.0.weight"], self._raw[a + ".to_out.0.bias"]
            # softmax rows sum to one, so the V bias passes through the attention unchanged:
            # to_out(P (V + 1 b_v^T)) = W_o P V + (W_o b_v + b_o)
            bo_f = (wo.double() @ bv.double() + bo.double()).float()
            return dict(r0=resnet(k + ".resnets.0"), r1=resnet(k + ".resnets.1"), gn=self._norm(a + ".group_norm"),
                        qk=self._wt(torch.cat([wq, wk], 0)), qkb=self._f32(torch.cat([bq, bk], 0)),
                        v=self._wa(wv), o=self._wt(wo), ob=self._f32(bo_f))

        rev = list(reversed(list(c.block_out_channels)))
        # post_quant_conv (1x1, 4->4) folded into conv_in's padded input: done as its own tiny GEMM on padded channels
        pq_w = torch.zeros(self._lc_pad, self._lc_pad)
        pq_w[: c.latent_channels, : c.latent_channels] = self._raw["post_quant_conv.weight"].reshape(c.latent_channels, -1)
        pq_b = torch.zeros(self._lc_pad)
        pq_b[: c.latent_channels] = self._raw["post_quant_conv.bias"]
        w["pq"] = (self._wt(pq_w), self._f32(pq_b))
        w["d_in"] = self._conv3("decoder.conv_in", self._lc_pad)
        w["d_mid"] = mid("decoder.mid_block")
        w["d_up"] = []
        for i in range(len(rev)):
            e = dict(res=[resnet(f"decoder.up_blocks.{i}.resnets.{j}") for j in range(c.layers_per_block + 1)])
            if i != len(rev) - 1:
                e["us"] = self._conv3(f"decoder.up_blocks.{i}.upsamplers.0.conv")
            w["d_up"].append(e)
        w["d_norm"] = self._norm("decoder.conv_norm_out")
        # conv_out: 3 output channels padded to 4 so the image is [B, HW, 4] (16-byte pixels in float32)
        wo = self._raw["decoder.conv_out.weight"]
        t = torch.zeros(4, 3, 3, wo.shape[1])
        t[: wo.shape[0]] = wo.permute(0, 2, 3, 1)
        bo = torch.zeros(4)
        bo[: wo.shape[0]] = self._raw["decoder.conv_out.bias"]
        w["d_out"] = (self._wt(t.reshape(4, -1)), self._f32(bo))
        if self.with_encoder:
            ch = list(c.block_out_channels)
            self._img_pad = _pad_to(c.in_channels, kmul)
            w["e_in"] = self._conv3("encoder.conv_in", self._img_pad)
            w["e_down"] = []
            for i in range(len(ch)):
                e = dict(res=[resnet(f"encoder.down_blocks.{i}.resnets.{j}") for j in range(c.layers_per_block)])
                if i != len(ch) - 1:
                    e["ds"] = self._conv3(f"encoder.down_blocks.{i}.downsamplers.0.conv")
                w["e_down"].append(e)
            w["e_mid"] = mid("encoder.mid_block")
            w["e_norm"] = self._norm("encoder.conv_norm_out")
            w["e_out"] = self._conv3("encoder.conv_out")
            w["q"] = self._lin("quant_conv")
        return w

    # ---- blocks ------------------------------------------------------------------------------
    def _resnet(self, r, x, B, H, W):
        G = self.config.norm_num_groups
        h = ops.groupnorm(x, B, G, r["n1"][0], r["n1"][1], 1e-6, silu=True)
        h, _, _ = ops.conv3x3(h, r["c1"][0], B, H, W, bias=r["c1"][1])
        h = ops.groupnorm(h, B, G, r["n2"][0], r["n2"][1], 1e-6, silu=True)
        if "sc" in r:
            x = ops.gemm_nt(x.view(-1, x.shape[-1]), r["sc"][0], bias=r["sc"][1]).view(B, H * W, -1)
        y, _, _ = ops.conv3x3(h, r["c2"][0], B, H, W, bias=r["c2"][1], residual=x)
        return y

    def _mid(self, m, x, B, H, W):
        x = self._resnet(m["r0"], x, B, H, W)
        C, N = x.shape[-1], H * W
        h = ops.groupnorm(x, B, self.config.norm_num_groups, m["gn"][0], m["gn"][1], 1e-6, silu=False)
        qk = ops.gemm_nt(h.view(B * N, C), m["qk"], bias=m["qkb"])  # [B*N, 2C]
        npad = _pad_to(N, 64 if ops.is_half(self._dtype) else 4)
        vt = torch.zeros((B, C, npad), dtype=self._dtype, device=x.device) if npad != N else None
        vt = ops.gemm_nt(m["v"], h.view(B, N, C), out=vt, ldc=npad)  # V^T (bias folded into to_out)
        o = composed_attention(qk, 0, 2 * C, qk, C, 2 * C, vt, B, 1, C, N, N, C ** -0.5, self._dtype)
        x = ops.gemm_nt(o.view(B * N, C), m["o"], bias=m["ob"], residual=x.view(B * N, C)).view(B, N, C)
        return self._resnet(m["r1"], x, B, H, W)

    # ---- public ------------------------------------------------------------------------------
    max_tensor_bytes = (1 << 32) - 1  # addressable through one raw buffer descriptor

    @_in_own_f32_mode
    def decode_nhwc(self, z):
        """z: float32 NCHW latents (already divided by scaling_factor).  Returns ([B, H*W, 4] float32
        channels-last image, channel 3 is padding), H, W."""
        self._ensure()
        w = self._w
        c = self.config
        if z.dtype != torch.float32:
            z = ops.cast(z.contiguous(), torch.float32)
        B, _, H, W = z.shape
        # The conv / GEMM kernels address a tensor with 32-bit byte offsets (raw buffer descriptors): keep the largest
        # activation of the decoder -- the widest full-resolution tensor of the last up block -- under 4 GiB by decoding the
        # batch in slices (16 images at 1024x1024 would need 8.6 GB for [B, 1024*1024, 256] in bf16).
        up = 2 ** (len(c.block_out_channels) - 1)
        widest = max(c.block_out_channels[0], c.block_out_channels[min(1, len(c.block_out_channels) - 1)])
        per_image = H * up * W * up * widest * (2 if ops.is_half(self._dtype) else 4)
        max_b = max(1, self.max_tensor_bytes // per_image)
        if B > max_b:
            parts = [self.decode_nhwc(z[i:i + max_b]) for i in range(0, B, max_b)]
            return torch.cat([p[0] for p in parts]), parts[0][1], parts[0][2]
        x = ops.pack_unet_input(z.contiguous(), None, 1, self._lc_pad, self._dtype)
        x = ops.gemm_nt(x.view(B * H * W, self._lc_pad), w["pq"][0], bias=w["pq"][1]).view(B, H * W, self._lc_pad)
        x, _, _ = ops.conv3x3(x, w["d_in"][0], B, H, W, bias=w["d_in"][1])
        x = self._mid(w["d_mid"], x, B, H, W)
        for blk in w["d_up"]:
            for r in blk["res"]:
                x = self._resnet(r, x, B, H, W)
            if "us" in blk:
                x, H, W = ops.conv3x3(x, blk["us"][0], B, H, W, bias=blk["us"][1], upsample=True)
        x = ops.groupnorm(x, B, c.norm_num_groups, w["d_norm"][0], w["d_norm"][1], 1e-6, silu=True)
        y, _, _ = ops.conv3x3(x, w["d_out"][0], B, H, W, bias=w["d_out"][1], out_dtype=torch.float32)
        return y, H, W

    def decode(self, z, return_dict=True, generator=None):
        y, H, W = self.decode_nhwc(z)
        img = ops.unpack_nchw(y, z.shape[0], self.config.out_channels, H, W)
        return (img,) if not return_dict else SimpleNamespace(sample=img)

    @_in_own_f32_mode
    def encode(self, x, return_dict=True):
        """x: float32 NCHW image in [-1,1] -> latent_dist (generate_hdr.py:208)."""
        self._ensure()
        if not self.with_encoder:
            raise RuntimeError("this AutoencoderKL was loaded without encoder weights")
        w = self._w
        c = self.config
        if x.dtype != torch.float32:
            x = ops.cast(x.contiguous(), torch.float32)
        B, _, H, W = x.shape
        h = ops.pack_unet_input(x.contiguous(), None, 1, self._img_pad, self._dtype)
        h, _, _ = ops.conv3x3(h, w["e_in"][0], B, H, W, bias=w["e_in"][1])
        for blk in w["e_down"]:
            for r in blk["res"]:
                h = self._resnet(r, h, B, H, W)
            if "ds" in blk:
                h, H, W = ops.conv3x3(h, blk["ds"][0], B, H, W, bias=blk["ds"][1], stride=2, pad_mode=1)
        h = self._mid(w["e_mid"], h, B, H, W)
        h = ops.groupnorm(h, B, c.norm_num_groups, w["e_norm"][0], w["e_norm"][1], 1e-6, silu=True)
        h, _, _ = ops.conv3x3(h, w["e_out"][0], B, H, W, bias=w["e_out"][1])
        m = ops.gemm_nt(h.view(B * H * W, -1), w["q"][0], bias=w["q"][1], out_dtype=torch.float32)
        moments = ops.unpack_nchw(m.view(B, H * W, -1), B, 2 * c.latent_channels, H, W)
        dist = DiagonalGaussianDistribution(moments[:, : c.latent_channels].contiguous(), moments[:, c.latent_channels:].contiguous())
        return SimpleNamespace(latent_dist=dist) if return_dict else (dist,)

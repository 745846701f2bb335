"""
``UNet2DConditionModel`` for the MI355X build -- the denoiser the reference pipelines call at
gm_diffusion/pipelines/stable_diffusion_gm.py:1051-1059 and
stable_diffusion_dual_unet.py:1052-1060, 1083-1092:

    unet(sample, t, encoder_hidden_states=..., return_dict=False)[0]

In the reference this is ``diffusers.UNet2DConditionModel`` (un-vendored dependency) with the
SD-1.5 hyper-parameters hard-coded at scripts/inference/generate_hdr.py:116-135 and
``in_channels`` 4 (SDR UNet) or 8 (GM UNet).  Here the same network runs entirely on the
hand-written HIP kernels of libgmd_hip.so over channels-last activations:

  * conv3x3 (+bias +time-embedding +residual)      -> gmd_conv3x3 (implicit GEMM, MFMA)
  * Linear / conv1x1 (+bias +residual +SiLU)       -> gmd_gemm_nt
  * GroupNorm(+SiLU), LayerNorm, GEGLU              -> gmd_groupnorm_*, gmd_layernorm, gmd_geglu
  * self / cross attention                          -> gmd_attention (bf16) or GEMM+softmax+GEMM (f32)
  * 8-channel concat / CFG duplicate / layout+cast  -> gmd_pack_unet_input, gmd_unpack_nchw

State-dict keys follow diffusers so a real SD-1.5 checkpoint loads.
There is no CPU path: ``forward`` raises ``HipExtensionError`` off-device.

dtype policy: weights/activations are ``self.dtype`` (bfloat16 or float16 = MFMA path -- float16 is what the
reference's own half-precision scripts use, scripts/stage2/experiments/batch_size_sweep.py -- float32 = parity
path); the input latents are float32 NCHW and are cast while being packed; the output eps is
float32 NCHW taken from the fp32 accumulators of ``conv_out`` (never rounded to bf16).
"""
from __future__ import annotations

import os

import torch

from .. import hip_ops as ops
from .._native import HipExtensionError
from .configuration import ConfigMixin, read_state_dict

SD15_UNET_DEFAULTS = dict(
    sample_size=64, in_channels=4, out_channels=4, center_input_sample=False, flip_sin_to_cos=True, freq_shift=0,
    down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
    up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
    block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, downsample_padding=1, mid_block_scale_factor=1,
    act_fn="silu", norm_num_groups=32, norm_eps=1e-5, cross_attention_dim=768, attention_head_dim=8,
    time_cond_proj_dim=None,
    # SDXL-style options (BASELINE.json configs[4]; the reference has no such path -- SURVEY.md §7 "SDXL / fp8" -- so these
    # follow diffusers' UNet2DConditionModel config names and are checked against the oracle's restatement only): all off for SD-1.5
    transformer_layers_per_block=1, use_linear_projection=False, addition_embed_type=None, addition_time_embed_dim=None,
    projection_class_embeddings_input_dim=None,
)

# stabilityai/stable-diffusion-xl-base-1.0 unet/config.json as restated in oracle/unet.py (2,567,463,684 parameters)
SDXL_UNET_CONFIG = dict(
    block_out_channels=(320, 640, 1280), down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
    up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"), transformer_layers_per_block=(1, 2, 10),
    attention_head_dim=(5, 10, 20), cross_attention_dim=2048, use_linear_projection=True, addition_embed_type="text_time",
    addition_time_embed_dim=256, projection_class_embeddings_input_dim=2816, sample_size=128,
)


def _pad_to(n, m):
    return (n + m - 1) // m * m


def _in_own_f32_mode(fn):
    """Run a module's compute method under the float32 contraction mode its weights were laid out for.  ``hip_ops`` reads the mode at
    every call, while a module's weights are pre-split / power-of-two scaled / channel-padded for ONE mode when they are placed on
    the device: a model prepared under "split" and run after ``set_f32_mode("exact")`` (bench.py holds both kinds alive) would
    otherwise run the split kernels' weights through the exact kernels or the other way round.  The override is scoped to the calling
    THREAD (``hip_ops.f32_mode_scope``): two host threads driving a "split" and an "exact" module at the same time keep their own
    modes.  16-bit modules do not depend on the mode."""
    import functools

    @functools.wraps(fn)
    def run(self, *args, **kwargs):
        self._ensure()
        if self._dtype != torch.float32 or self._f32_mode == ops.f32_mode():
            return fn(self, *args, **kwargs)
        with ops.f32_mode_scope(self._f32_mode):
            return fn(self, *args, **kwargs)

    return run


class _HipModule(ConfigMixin):
    """Shared weight handling: raw CPU float32 state-dict -> device tensors in kernel layouts."""

    def _init_module(self):
        self._raw = None       # diffusers-keyed CPU float32 tensors
        self._w = None         # prepared device tensors
        self._dtype = torch.float32
        self._device = torch.device("cpu")
        self._f32_mode = None  # float32 contraction mode the prepared weights are laid out for (set by _ensure)

    @property
    def dtype(self):
        return self._dtype

    @property
    def device(self):
        return self._device

    def state_dict(self):
        return dict(self._raw or {})

    def load_state_dict(self, sd, strict=True):
        want = self.expected_keys()
        missing = [k for k in want if k not in sd]
        unexpected = [k for k in sd if k not in want]
        if strict and (missing or unexpected):
            raise KeyError(f"{type(self).__name__}.load_state_dict: missing={missing[:5]} (+{max(len(missing) - 5, 0)}) "
                           f"unexpected={unexpected[:5]} (+{max(len(unexpected) - 5, 0)})")
        bad = [k for k in want if k in sd and tuple(sd[k].shape) != tuple(want[k])]
        if bad:
            raise ValueError(f"shape mismatch for {bad[:5]}: e.g. {tuple(sd[bad[0]].shape)} vs {tuple(want[bad[0]])}")
        self._raw = {k: sd[k].detach().to("cpu", torch.float32) for k in want if k in sd}
        self._w = None
        return self

    def to(self, *args, **kwargs):
        device, dtype = kwargs.get("device"), kwargs.get("dtype")
        for a in args:
            if isinstance(a, torch.dtype):
                dtype = a
            elif a is not None:
                device = a
        if dtype is not None:
            self._dtype = dtype
        if device is not None:
            self._device = torch.device(device)
        self._w = None
        return self

    def cuda(self):
        return self.to("cuda")

    def eval(self):
        return self

    def requires_grad_(self, flag=False):
        return self

    def _ensure(self):
        if self._w is not None:
            return
        if self._raw is None:
            raise RuntimeError(f"{type(self).__name__} has no weights: call load_state_dict()/from_pretrained() first")
        if self._device.type != "cuda":
            raise HipExtensionError(f"{type(self).__name__} runs on hand-written HIP kernels only; call .to('cuda') "
                                    "(there is no CPU fallback in the MI355X build)")
        ops.dtype_code(self._dtype)
        self._f32_mode = ops.f32_mode()  # _prepare and every later forward of this module use THIS mode (_in_own_f32_mode)
        self._w = self._prepare()

    # -- layout helpers --------------------------------------------------------------------------
    def _kmul(self):
        """Channel multiple the convolution kernels need: 64 (16-bit), 32 (float32 on the matrix cores), 16 (exact float32)."""
        return 64 if ops.is_half(self._dtype) else (32 if self._split_mode() else 16)

    def _act(self, t):
        return t.to(self._device, self._dtype).contiguous()

    def _f32(self, t):
        return t.to(self._device, torch.float32).contiguous()

    def _split_mode(self):
        """float32 module laid out for the matrix-core ("split") kernels: from the record taken when the weights were prepared."""
        return self._dtype == torch.float32 and (self._f32_mode or ops.f32_mode()) == "split"

    def _wt(self, t):
        """A weight that is the W operand of gemm_nt / conv3x3.  float32 on the matrix cores: scaled by a power of two and
        split into float16 hi / lo once, here (hip_ops.split_weights)."""
        w = self._act(t)
        if self._split_mode() and w.dim() == 2 and w.shape[1] % 32 == 0:
            return ops.split_weights(w)
        return w

    def _wa(self, t):
        """A weight that is the A operand (V^T[b] = W_v x_b^T): plain float32, power-of-two scaled in split mode."""
        w = self._act(t)
        return ops.scale_weight(w) if self._split_mode() else w

    def _conv3(self, key, cin_pad=None):
        w = self._raw[key + ".weight"]  # [Cout, Cin, 3, 3]
        cout, cin = w.shape[0], w.shape[1]
        cin_pad = cin_pad or cin
        t = torch.zeros(cout, 3, 3, cin_pad)
        t[..., :cin] = w.permute(0, 2, 3, 1)
        return self._wt(t.reshape(cout, 9 * cin_pad)), self._f32(self._raw[key + ".bias"])

    def _lin(self, key, bias=True):
        w = self._raw[key + ".weight"]
        w = w.reshape(w.shape[0], -1)  # Linear [out,in] or conv1x1 [out,in,1,1]
        return self._wt(w), (self._f32(self._raw[key + ".bias"]) if bias else None)

    def _norm(self, key):
        return self._f32(self._raw[key + ".weight"]), self._f32(self._raw[key + ".bias"])


def composed_attention(q, q_col, ldq, k, k_col, ldk, vt, B, H, d, nq, nk, scale, dtype, causal=False):
    """Attention as GEMM -> row softmax -> GEMM (float32 parity path and the d=512 VAE block).
    q/k: buffers with rows of ldq/ldk elements, head h at columns q_col + h*d; vt: [B, H*d, ldvt] with
    zero-filled columns >= nk.  Returns [B, nq, H*d]."""
    es = q.element_size()
    mul = 64 if ops.is_half(dtype) else 4
    nkp = _pad_to(nk, mul)
    ldvt = vt.shape[2]
    if ldvt < nkp:
        raise HipExtensionError("composed_attention: vt row stride too small")
    out = torch.empty((B, nq, H * d), dtype=dtype, device=q.device)
    s = torch.empty((H, nq, nkp), dtype=torch.float32, device=q.device)
    for b in range(B):
        ops.gemm_raw(q.data_ptr() + (b * nq * ldq + q_col) * es, k.data_ptr() + (b * nk * ldk + k_col) * es, s.data_ptr(),
                     dtype, torch.float32, nq, nk, d, ldq, ldk, nkp, batch=H, sA=d, sW=d, sC=nq * nkp)
        p = ops.softmax_rows(s, nk, scale, dtype, ldp=nkp, causal_nq=nq if causal else 0)
        # float32: the probabilities (~1/nk) would lose their float16 lo half to the subnormal range in a split product, so
        # this one stays on the exact kernel (small: the VAE mid block and the 77-token text encoder only)
        ops.gemm_raw(p.data_ptr(), vt.data_ptr() + b * H * d * ldvt * es, out.data_ptr() + b * nq * H * d * es,
                     dtype, dtype, nq, d, nkp, nkp, ldvt, H * d, batch=H, sA=nq * nkp, sW=d * ldvt, sC=d, exact=True)
    return out


class _GraphedForward:
    """A captured UNet forward: static input ``x``, static output ``out``, ``replay()`` on the current stream."""

    x = None
    out = None
    graph = None
    ws = None

    def replay(self):
        self.graph.replay()
        return self.out


class UNet2DConditionModel(_HipModule):
    config_name = "config.json"
    max_cached_graphs = 8  # captured forwards kept per model (one per batch / latent size / token count)
    _defaults = SD15_UNET_DEFAULTS

    def __init__(self, **config):
        cfg = dict(SD15_UNET_DEFAULTS)
        unknown = [k for k in config if k not in cfg and not k.startswith("_")]
        cfg.update({k: v for k, v in config.items() if k in cfg})
        # generate_hdr.py:103-105 rewrites `num_attention_heads` to the legacy `attention_head_dim` name
        if "num_attention_heads" in config and config["num_attention_heads"] is not None:
            cfg["attention_head_dim"] = config["num_attention_heads"]
        if isinstance(cfg["attention_head_dim"], (list, tuple)) and len(set(cfg["attention_head_dim"])) == 1:
            cfg["attention_head_dim"] = cfg["attention_head_dim"][0]
        if cfg["act_fn"] != "silu" or cfg["time_cond_proj_dim"] is not None or cfg["center_input_sample"]:
            raise NotImplementedError("only silu / no time_cond_proj / no centering UNets are implemented (SD-1.5, SDXL)")
        if cfg["addition_embed_type"] not in (None, "text_time"):
            raise NotImplementedError(f"addition_embed_type={cfg['addition_embed_type']!r}")
        self._unknown_config = unknown
        self.register_to_config(**cfg)
        self._init_module()
        self._kv_cache = {}
        self._t_dev = None
        self._capturing = False
        self._transformers = []
        self._graphs = {}
        self._aug = {}  # text_time conditioning: persistent [B, temb] buffers, rewritten in place by set_added_cond

    # ------------------------------------------------------------------------------------------
    # structure
    # ------------------------------------------------------------------------------------------
    def _layout(self):
        """Yield the block structure shared by expected_keys / _prepare / forward."""
        c = self.config
        ch = list(c.block_out_channels)
        n = c.layers_per_block
        heads, depth = self._per_level(c.attention_head_dim), self._per_level(c.transformer_layers_per_block)
        downs, cout = [], ch[0]
        for i, t in enumerate(c.down_block_types):
            cin, cout = cout, ch[i]
            downs.append(dict(res=[(cin if j == 0 else cout, cout) for j in range(n)], attn=t.startswith("CrossAttn"),
                              down=i != len(ch) - 1, c=cout, heads=heads[i], depth=depth[i]))
        rev, rheads, rdepth = list(reversed(ch)), list(reversed(heads)), list(reversed(depth))
        ups, cout = [], rev[0]
        for i, t in enumerate(c.up_block_types):
            cprev, cout = cout, rev[i]
            cin = rev[min(i + 1, len(ch) - 1)]
            res = []
            for j in range(n + 1):
                skip = cin if j == n else cout
                rin = cprev if j == 0 else cout
                res.append((rin + skip, cout))
            ups.append(dict(res=res, attn=t.startswith("CrossAttn"), up=i != len(ch) - 1, c=cout, heads=rheads[i], depth=rdepth[i]))
        return downs, ups

    def _per_level(self, v):
        """An int or one value per resolution level (``attention_head_dim`` = heads, ``transformer_layers_per_block``)."""
        L = len(self.config.block_out_channels)
        v = list(v) if isinstance(v, (list, tuple)) else [v] * L
        if len(v) != L:
            raise ValueError(f"expected one value per level ({L}), got {v}")
        return v

    def expected_keys(self):
        c = self.config
        ch = list(c.block_out_channels)
        temb, cross = ch[0] * 4, c.cross_attention_dim
        keys = {}

        def conv(k, co, ci, ks):
            keys[k + ".weight"], keys[k + ".bias"] = (co, ci, ks, ks), (co,)

        def lin(k, co, ci, bias=True):
            keys[k + ".weight"] = (co, ci)
            if bias:
                keys[k + ".bias"] = (co,)

        def norm(k, n):
            keys[k + ".weight"], keys[k + ".bias"] = (n,), (n,)

        def resnet(k, ci, co):
            norm(k + ".norm1", ci); conv(k + ".conv1", co, ci, 3); lin(k + ".time_emb_proj", co, temb)
            norm(k + ".norm2", co); conv(k + ".conv2", co, co, 3)
            if ci != co:
                conv(k + ".conv_shortcut", co, ci, 1)

        def transformer(k, d, depth):
            norm(k + ".norm", d)
            if c.use_linear_projection:
                lin(k + ".proj_in", d, d); lin(k + ".proj_out", d, d)
            else:
                conv(k + ".proj_in", d, d, 1); conv(k + ".proj_out", d, d, 1)
            for n_ in range(depth):
                b = f"{k}.transformer_blocks.{n_}"
                for nm in ("norm1", "norm2", "norm3"):
                    norm(f"{b}.{nm}", d)
                for a, kd in (("attn1", d), ("attn2", cross)):
                    lin(f"{b}.{a}.to_q", d, d, False); lin(f"{b}.{a}.to_k", d, kd, False); lin(f"{b}.{a}.to_v", d, kd, False)
                    lin(f"{b}.{a}.to_out.0", d, d)
                lin(f"{b}.ff.net.0.proj", 8 * d, d); lin(f"{b}.ff.net.2", d, 4 * d)

        conv("conv_in", ch[0], c.in_channels, 3)
        lin("time_embedding.linear_1", temb, ch[0]); lin("time_embedding.linear_2", temb, temb)
        if c.addition_embed_type == "text_time":
            lin("add_embedding.linear_1", temb, c.projection_class_embeddings_input_dim); lin("add_embedding.linear_2", temb, temb)
        downs, ups = self._layout()
        for i, blk in enumerate(downs):
            for j, (ci, co) in enumerate(blk["res"]):
                resnet(f"down_blocks.{i}.resnets.{j}", ci, co)
                if blk["attn"]:
                    transformer(f"down_blocks.{i}.attentions.{j}", co, blk["depth"])
            if blk["down"]:
                conv(f"down_blocks.{i}.downsamplers.0.conv", blk["c"], blk["c"], 3)
        resnet("mid_block.resnets.0", ch[-1], ch[-1]); transformer("mid_block.attentions.0", ch[-1], downs[-1]["depth"])
        resnet("mid_block.resnets.1", ch[-1], ch[-1])
        for i, blk in enumerate(ups):
            for j, (ci, co) in enumerate(blk["res"]):
                resnet(f"up_blocks.{i}.resnets.{j}", ci, co)
                if blk["attn"]:
                    transformer(f"up_blocks.{i}.attentions.{j}", co, blk["depth"])
            if blk["up"]:
                conv(f"up_blocks.{i}.upsamplers.0.conv", blk["c"], blk["c"], 3)
        norm("conv_norm_out", ch[0]); conv("conv_out", c.out_channels, ch[0], 3)
        return keys

    # ------------------------------------------------------------------------------------------
    # loading
    # ------------------------------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, path, subfolder=None, torch_dtype=None, **config_overrides):
        """Read a diffusers-layout directory (``config.json`` + ``diffusion_pytorch_model.safetensors``).
        Extra kwargs override config entries, as scripts/inference/generate_hdr.py:138-142 does
        (``in_channels=8, **config``)."""
        d = os.path.join(path, subfolder) if subfolder else path
        cfg = cls.load_config(d)
        cfg.update(config_overrides)
        m = cls(**cfg)
        m.load_state_dict(read_state_dict(d))
        if torch_dtype is not None:
            m.to(torch_dtype)
        return m

    def save_pretrained(self, directory):
        import json
        from safetensors.torch import save_file

        os.makedirs(directory, exist_ok=True)
        with open(os.path.join(directory, "config.json"), "w") as f:
            json.dump({"_class_name": "UNet2DConditionModel", **{k: (list(v) if isinstance(v, tuple) else v) for k, v in self.config.items()}}, f, indent=2)
        save_file({k: v.contiguous() for k, v in self._raw.items()}, os.path.join(directory, "diffusion_pytorch_model.safetensors"))

    def replace_conv_in(self, in_channels=8):
        """The reference's ``_replace_unet_conv_in`` (scripts/inference/generate_hdr.py:75-94,
        scripts/stage2/train_gm_unet.py:658-677): widen ``conv_in`` to ``in_channels`` by repeating the weight along
        the input-channel axis and halving it; the bias is kept.  Returns self."""
        w, b = self._raw["conv_in.weight"], self._raw["conv_in.bias"]
        rep = in_channels // w.shape[1]
        if rep * w.shape[1] != in_channels:
            raise ValueError(f"in_channels={in_channels} is not a multiple of {w.shape[1]}")
        sd = dict(self._raw)
        sd["conv_in.weight"] = w.repeat(1, rep, 1, 1) * 0.5
        sd["conv_in.bias"] = b.clone()
        cfg = dict(self.config)
        cfg["in_channels"] = in_channels
        self._internal_dict = type(self._internal_dict)(cfg)
        return self.load_state_dict(sd)

    def init_random(self, seed=1234, device=None):
        """Deterministic synthetic weights (uniform +-1/sqrt(fan_in), unit norms) for benchmarks; no checkpoint
        exists offline.  ``device``: draw them with that device's generator instead of the host's (the
        multi-GPU bench: 8 ranks each drawing 1.8 G host randoms would be the longest phase of the run); the values then
        differ from the host draw but are the same on every rank."""
        g = torch.Generator(device or "cpu").manual_seed(seed)
        keys = self.expected_keys()
        sd = {}
        for k, shp in keys.items():
            if ".norm" in k or k.startswith("conv_norm_out") or k.endswith("norm.weight") or k.endswith("norm.bias"):
                sd[k] = torch.ones(shp) if k.endswith("weight") else torch.zeros(shp)
                continue
            wshape = keys[k.rsplit(".", 1)[0] + ".weight"]
            fan_in = 1
            for s_ in wshape[1:]:
                fan_in *= s_
            bound = fan_in ** -0.5
            sd[k] = (torch.rand(shp, generator=g, device=g.device) * 2 - 1) * bound
        return self.load_state_dict(sd)

    # ------------------------------------------------------------------------------------------
    # weight preparation
    # ------------------------------------------------------------------------------------------
    def _prepare(self):
        c = self.config
        w = {}
        self._cin_pad = _pad_to(c.in_channels, self._kmul())
        w["conv_in"] = self._conv3("conv_in", self._cin_pad)
        w["te1"] = self._lin("time_embedding.linear_1")
        w["te2"] = self._lin("time_embedding.linear_2")

        te_w, te_b = [], []
        self._transformers = []

        def resnet(k):
            r = dict(n1=self._norm(k + ".norm1"), c1=self._conv3(k + ".conv1"), n2=self._norm(k + ".norm2"), c2=self._conv3(k + ".conv2"))
            r["te_off"] = sum(t.shape[0] for t in te_w)  # column offset into the fused time-embedding projection
            te_w.append(self._raw[k + ".time_emb_proj.weight"])
            te_b.append(self._raw[k + ".time_emb_proj.bias"])
            if k + ".conv_shortcut.weight" in self._raw:
                r["sc"] = self._lin(k + ".conv_shortcut")
            return r

        def transformer(k, heads, depth):
            """GroupNorm + proj_in / proj_out around ``depth`` BasicTransformerBlocks (1 for SD-1.5; 1 / 2 / 10 per level for SDXL)."""
            t = dict(norm=self._norm(k + ".norm"), pin=self._lin(k + ".proj_in"), pout=self._lin(k + ".proj_out"), heads=heads, blocks=[])
            for n_ in range(depth):
                b = f"{k}.transformer_blocks.{n_}"
                blk = dict(heads=heads)
                for nm in ("norm1", "norm2", "norm3"):
                    blk[nm] = self._norm(f"{b}.{nm}")
                q1, k1 = self._raw[f"{b}.attn1.to_q.weight"], self._raw[f"{b}.attn1.to_k.weight"]
                blk["v1"] = self._wa(self._raw[f"{b}.attn1.to_v.weight"])
                # q | k | v stacked: ONE launch writes Q|K row-major and V transposed (hip_ops.gemm_qkv_vt; 16-bit and float32 split modes)
                blk["qkv1"] = self._wt(torch.cat([q1, k1, self._raw[f"{b}.attn1.to_v.weight"]], 0))
                # fused [2C, C] projection of the fallback path.  16-bit modes: the first 2C rows of the stacked matrix (a contiguous
                # view, no third copy of q | k: ~62 MB per SD-1.5 UNet); the float32 split mode keeps its own copy, because a pre-split
                # weight carries ONE power-of-two scale for the whole matrix and q | k alone would get another
                blk["qk1"] = blk["qkv1"][: q1.shape[0] + k1.shape[0]] if ops.is_half(self._dtype) else self._wt(torch.cat([q1, k1], 0))
                blk["o1"] = self._lin(f"{b}.attn1.to_out.0")
                blk["q2"] = self._lin(f"{b}.attn2.to_q", False)[0]
                blk["k2"] = self._lin(f"{b}.attn2.to_k", False)[0]
                blk["v2"] = self._wa(self._raw[f"{b}.attn2.to_v.weight"])
                blk["o2"] = self._lin(f"{b}.attn2.to_out.0")
                if ops.is_half(self._dtype) or self._split_mode():
                    # fused GEGLU epilogue: interleave value / gate rows in groups of 16 so both land in the same MFMA lane
                    wf, bf = self._raw[f"{b}.ff.net.0.proj.weight"], self._raw[f"{b}.ff.net.0.proj.bias"]
                    half = wf.shape[0] // 2
                    wi = torch.stack([wf[:half].reshape(half // 16, 16, -1), wf[half:].reshape(half // 16, 16, -1)], 1).reshape(2 * half, -1)
                    bi = torch.stack([bf[:half].reshape(half // 16, 16), bf[half:].reshape(half // 16, 16)], 1).reshape(2 * half)
                    blk["ff1"] = (self._wt(wi), self._f32(bi))
                    blk["ff1_fused"] = True
                else:
                    blk["ff1"] = self._lin(f"{b}.ff.net.0.proj")
                    blk["ff1_fused"] = False
                blk["ff2"] = self._lin(f"{b}.ff.net.2")
                blk["key"] = b
                self._transformers.append(blk)  # every block has its own cross-attention K / V^T of the text tokens
                t["blocks"].append(blk)
            return t

        downs, ups = self._layout()
        if c.addition_embed_type == "text_time":
            wa = self._raw["add_embedding.linear_1.weight"]  # K = pooled text width + 6 sinusoids: zero-padded to the kernels' K multiple
            self._add_k = _pad_to(wa.shape[1], self._kmul())
            wp = torch.zeros(wa.shape[0], self._add_k)
            wp[:, : wa.shape[1]] = wa
            w["add1"] = (self._wt(wp), self._f32(self._raw["add_embedding.linear_1.bias"]))
            w["add2"] = self._lin("add_embedding.linear_2")
        w["down"] = []
        for i, blk in enumerate(downs):
            e = dict(res=[resnet(f"down_blocks.{i}.resnets.{j}") for j in range(len(blk["res"]))])
            if blk["attn"]:
                e["attn"] = [transformer(f"down_blocks.{i}.attentions.{j}", blk["heads"], blk["depth"]) for j in range(len(blk["res"]))]
            if blk["down"]:
                e["ds"] = self._conv3(f"down_blocks.{i}.downsamplers.0.conv")
            w["down"].append(e)
        w["mid"] = dict(r0=resnet("mid_block.resnets.0"), a=transformer("mid_block.attentions.0", downs[-1]["heads"], downs[-1]["depth"]),
                        r1=resnet("mid_block.resnets.1"))
        w["up"] = []
        for i, blk in enumerate(ups):
            e = dict(res=[resnet(f"up_blocks.{i}.resnets.{j}") for j in range(len(blk["res"]))])
            if blk["attn"]:
                e["attn"] = [transformer(f"up_blocks.{i}.attentions.{j}", blk["heads"], blk["depth"]) for j in range(len(blk["res"]))]
            if blk["up"]:
                e["us"] = self._conv3(f"up_blocks.{i}.upsamplers.0.conv")
            w["up"].append(e)
        w["norm_out"] = self._norm("conv_norm_out")
        w["conv_out"] = self._conv3("conv_out")
        # all ResnetBlock2D.time_emb_proj layers as ONE [sum(Cout), 1280] GEMM per forward
        w["te_all"] = (self._wt(torch.cat(te_w, 0)), self._f32(torch.cat(te_b, 0)))
        self._kv_cache = {}
        self._graphs = {}
        self._aug = {}
        self._t_dev = torch.zeros(1, dtype=torch.float32, device=self._device)
        return w

    # ------------------------------------------------------------------------------------------
    # forward
    # ------------------------------------------------------------------------------------------
    @staticmethod
    def _sa(*ws):
        """Store the activation that feeds these weights PRE-SPLIT (hip_ops: GMD_F32SA)?  Only when every consumer weight is itself
        pre-split, i.e. the float32 matrix-core mode; the producers then write [hi | lo] and the contractions skip their conversions."""
        return ops.USE_F32SA and all(getattr(w, "_split", False) for w in ws)

    def _resnet(self, r, x, B, H, W, temb, eps):
        G = self.config.norm_num_groups
        cs = self._wants_colstats(H, W)  # the GroupNorm that reads a conv output takes its statistics from the conv epilogue
        h = ops.groupnorm(x, B, G, r["n1"][0], r["n1"][1], eps, silu=True, split_out=self._sa(r["c1"][0]))
        # conv1 -> + time embedding -> norm2 -> SiLU: one GroupNorm launch over the split-K slabs on the 16x16 / 8x8 levels
        _, h = ops.conv3x3_groupnorm(h, r["c1"][0], B, H, W, G, r["n2"][0], r["n2"][1], eps, silu=True, bias=r["c1"][1],
                                     rowbias=(temb, r["te_off"]), colstats=cs, split_out=self._sa(r["c2"][0]))
        if "sc" in r:
            cin = x.shape[-1]
            x = ops.gemm_nt(x.view(-1, cin), r["sc"][0], bias=r["sc"][1]).view(B, H * W, -1)
        y, _, _ = ops.conv3x3(h, r["c2"][0], B, H, W, bias=r["c2"][1], residual=x, colstats=cs)
        return y

    @staticmethod
    def _wants_colstats(H, W):
        """Levels whose GroupNorms take the two-launch split path (hip_ops.groupnorm): there the producer's epilogue
        statistics replace the statistics launch.  Smaller levels use the single-launch GroupNorm, which needs none."""
        return H * W >= 1024 and (H * W) % 64 == 0

    def _self_attention(self, t, n1, B, N, C):
        heads = t["heads"]
        d = C // heads
        scale = d ** -0.5
        if "qkv1" in t and (ops.is_half(self._dtype) or ops.split_attention_ok(self._dtype, d)):
            r = ops.gemm_qkv_vt(n1, t["qkv1"], 2 * C, N)
            if r is not None:
                qk = r[0].view(B, N, 2 * C)
                return ops.attention(qk, qk, r[1], heads, N, scale, k_col=C, split_out=self._sa(t["o1"][0]))
        qk = ops.gemm_nt(n1, t["qk1"]).view(B, N, 2 * C)
        if ops.is_half(self._dtype) or ops.split_attention_ok(self._dtype, d):
            nv = n1.view(B, N, C)
            if ops.is_asplit(n1):  # a pre-split activation as the W operand of V^T = Wv n1^T: exactly the pre-split weight layout
                nv._split, nv._alpha = True, 1.0
            vt = ops.gemm_nt(t["v1"], nv, ldc=_pad_to(N, 8 if ops.is_half(self._dtype) else 4))  # V^T [B, C, N]
            return ops.attention(qk, qk, vt, heads, N, scale, k_col=C, split_out=self._sa(t["o1"][0]))
        npad = _pad_to(N, 4)
        if npad != N:
            vt = torch.zeros((B, C, npad), dtype=self._dtype, device=n1.device)
            ops.gemm_nt(t["v1"], n1.view(B, N, C), out=vt, ldc=npad)
        else:
            vt = ops.gemm_nt(t["v1"], n1.view(B, N, C), ldc=npad)
        return composed_attention(qk, 0, 2 * C, qk, C, 2 * C, vt, B, heads, d, N, N, scale, self._dtype)

    def _cross_kv(self, t, ehs):
        """K and V^T of the text tokens: constant over the whole denoising loop, so computed once per embedding tensor.
        The buffers are persistent per (layer, shape) and rewritten in place for new embeddings, so a captured HIP
        graph of the forward keeps reading valid addresses."""
        key = (ehs.data_ptr(), ehs._version, tuple(ehs.shape))
        B, L, E = ehs.shape
        # one persistent buffer pair per (layer, batch, tokens): a graph captured for one batch size must keep finding ITS
        # buffers after the model has served another batch size in between
        slot = (t["key"], B, L, self._dtype)
        ent = self._kv_cache.get(slot)
        if ent is not None and ent["key"] == key:
            return ent["kc"], ent["vt"]
        if self._capturing:
            raise HipExtensionError("cross-attention K/V must be prepared (update_context) before graph capture")
        C = t["k2"].shape[0]
        lpad = _pad_to(L, 8 if ops.is_half(self._dtype) else 4)
        if ent is None:
            ent = dict(kc=torch.empty((B, L, C), dtype=self._dtype, device=ehs.device),
                       vt=torch.zeros((B, C, lpad), dtype=self._dtype, device=ehs.device))
            self._kv_cache[slot] = ent
        ops.gemm_nt(ehs.view(B * L, E), t["k2"], out=ent["kc"].view(B * L, C))
        ops.gemm_nt(t["v2"], ehs, out=ent["vt"], ldc=lpad)
        ent["key"] = key
        ent["ehs"] = ehs  # holding `ehs` keeps its address from being recycled while the key is valid
        return ent["kc"], ent["vt"]

    @_in_own_f32_mode
    def update_context(self, ehs):
        """Prepare the cross-attention K / V^T of every transformer block for these text hidden states (eager,
        in place).  Called once per pipeline call; required before replaying a captured forward."""
        self._ensure()
        for t in self._transformers:
            self._cross_kv(t, ehs)

    @staticmethod
    def _dup_batch(t):
        """[B, ...] -> [2B, ...]: both halves equal to ``t`` (gmd_dup_batch: one read, two writes)."""
        out = ops.dup_batch(t.contiguous())
        st = getattr(t, "_colstats", None)
        if st is not None and not isinstance(st, list):  # producer statistics are per 64-row block, sample-major: duplicate alike
            out._colstats = (torch.cat([st[0], st[0]], 0), st[1])
        return out

    def _transformer(self, t, x, B, H, W, ehs, cfg_dup=False):
        """``cfg_dup``: x holds the B UNIQUE samples of a classifier-free-guidance pair whose two halves differ only in the
        text conditioning; everything up to the first cross-attention query is computed once, then duplicated (returns 2B rows)."""
        C = x.shape[-1]
        N = H * W
        h = ops.groupnorm(x, B, self.config.norm_num_groups, t["norm"][0], t["norm"][1], 1e-6, silu=False, split_out=self._sa(t["pin"][0]))
        h = ops.gemm_nt(h.view(B * N, C), t["pin"][0], bias=t["pin"][1], a_split=ops.is_asplit(h))  # conv1x1 and nn.Linear proj_in are the same GEMM on tokens
        for blk in t["blocks"]:
            h, x, B = self._transformer_block(blk, h, x, B, N, C, ehs, cfg_dup)
            cfg_dup = False  # the first cross-attention has duplicated the batch
        y = ops.gemm_nt(h, t["pout"][0], bias=t["pout"][1], residual=x.view(B * N, C), colstats=self._wants_colstats(H, W))
        return ops.carry_colstats(y.view(B, N, C), y)

    def _transformer_block(self, t, h, x, B, N, C, ehs, cfg_dup):
        """One BasicTransformerBlock on tokens ``h`` [B*N, C]; returns (h, x, B) -- x / B change when ``cfg_dup`` duplicates."""
        heads = t["heads"]
        d = C // heads
        # self-attention
        # (the V^T product takes n1 as its W operand: pre-split only where the matrix-core attention path is taken)
        n1 = ops.layernorm(h, *t["norm1"], split_out=self._sa(t["qk1"]) and ops.split_attention_ok(self._dtype, d))
        o = self._self_attention(t, n1, B, N, C)
        h = ops.gemm_nt(o.view(B * N, C), t["o1"][0], bias=t["o1"][1], residual=h, a_split=ops.is_asplit(o))
        # cross-attention over the text tokens
        n2 = ops.layernorm(h, *t["norm2"], split_out=self._sa(t["q2"]))
        q = ops.gemm_nt(n2, t["q2"]).view(B, N, C)
        if cfg_dup:  # first use of the text conditioning: from here on the two halves differ
            q, h, x = self._dup_batch(q), self._dup_batch(h.view(B, N, C)).view(2 * B * N, C), self._dup_batch(x)
            B *= 2
        kc, vtc = self._cross_kv(t, ehs)
        L = ehs.shape[1]
        if ops.is_half(self._dtype) or ops.split_attention_ok(self._dtype, d):
            o = ops.attention(q, kc, vtc, heads, L, d ** -0.5, split_out=self._sa(t["o2"][0]))
        else:
            o = composed_attention(q, 0, C, kc, 0, C, vtc, B, heads, d, N, L, d ** -0.5, self._dtype)
        h = ops.gemm_nt(o.view(B * N, C), t["o2"][0], bias=t["o2"][1], residual=h, a_split=ops.is_asplit(o))
        # GEGLU feed-forward
        n3 = ops.layernorm(h, *t["norm3"], split_out=self._sa(t["ff1"][0]))
        if t["ff1_fused"] and ops.ff_fused_ok(n3, C):  # the whole feed-forward in one launch: [tokens, 4C] never reaches HBM
            h = ops.ff_geglu_fused(n3, t["ff1"][0], t["ff1"][1], t["ff2"][0], t["ff2"][1], h)
        else:
            if t["ff1_fused"]:  # h * gelu(g) formed in the GEMM epilogue (float32 matrix-core mode: stored pre-split for ff2)
                f = ops.gemm_nt(n3, t["ff1"][0], bias=t["ff1"][1], act=ops.ACT_GEGLU, split_out=self._sa(t["ff2"][0]))
            else:
                f = ops.geglu(ops.gemm_nt(n3, t["ff1"][0], bias=t["ff1"][1]))
            h = ops.gemm_nt(f, t["ff2"][0], bias=t["ff2"][1], residual=h)
        return h, x, B

    def set_timestep(self, timestep):
        """Write the timestep into the device scalar read by the embedding kernel (kept outside any
        captured graph so the same graph can be replayed for every step)."""
        self._ensure()
        t = float(timestep)
        self._t_dev.copy_(torch.tensor([t], dtype=torch.float32), non_blocking=False)

    def set_timestep_from(self, ts_dev, i):
        """Device-to-device variant: copy element ``i`` of a float32 device vector of timesteps (no host round trip,
        so the host can run ahead of the GPU)."""
        self._ensure()
        self._t_dev.copy_(ts_dev[i:i + 1], non_blocking=True)
        if getattr(self, "_stamp_buf", None) is not None:  # measurement hook: this iteration's row of the stamp table
            self._stamp_row.fill_(i)

    def supports_cfg_shared(self):
        """True when the first down block has a transformer (SD-1.5): the prefix shared by a CFG pair ends at its
        cross-attention."""
        return self.config.down_block_types[0].startswith("CrossAttn") and self.config.addition_embed_type is None

    @_in_own_f32_mode
    def set_added_cond(self, added_cond_kwargs, B):
        """SDXL micro-conditioning (diffusers ``get_aug_embed`` for addition_embed_type "text_time"): sinusoid of every scalar of
        ``time_ids`` [B, 6] concatenated behind the pooled text embedding ``text_embeds`` [B, P], through add_embedding.linear_1 ->
        SiLU -> linear_2; the result is ADDED to the time embedding in every forward.  Constant over the denoising loop: computed
        once per call here (eager, HIP kernels) into a persistent [B, temb] buffer that captured graphs keep reading."""
        self._ensure()
        c = self.config
        if c.addition_embed_type != "text_time":
            if added_cond_kwargs:
                raise NotImplementedError("this UNet has no addition embedding (added_cond_kwargs given)")
            return None
        if not added_cond_kwargs or "text_embeds" not in added_cond_kwargs or "time_ids" not in added_cond_kwargs:
            raise ValueError("this UNet needs added_cond_kwargs = {'text_embeds': [B, P], 'time_ids': [B, 6]} (addition_embed_type 'text_time')")
        te, ids = added_cond_kwargs["text_embeds"], added_cond_kwargs["time_ids"]
        if te.shape[0] != B or ids.shape[0] != B or not te.is_cuda:
            raise ValueError(f"added_cond_kwargs must hold {B} device rows (got {tuple(te.shape)}, {tuple(ids.shape)})")
        d = c.addition_time_embed_dim
        P = te.shape[1]
        if P + ids.shape[1] * d != c.projection_class_embeddings_input_dim:
            raise ValueError("text_embeds / time_ids widths do not match projection_class_embeddings_input_dim")
        key = (te.data_ptr(), te._version, ids.data_ptr(), ids._version, B)
        ent = self._aug.get(B)
        if ent is not None and ent["key"] == key:
            return ent["aug"]
        if self._capturing:
            raise HipExtensionError("the added conditioning must be prepared (set_added_cond) before graph capture")
        cat = torch.zeros((B, self._add_k), dtype=self._dtype, device=self._device)
        tex = te.contiguous()
        if tex.dtype != self._dtype:
            tex = ops.cast(tex if tex.dtype in (torch.float32, torch.bfloat16, torch.float16) else tex.float(), self._dtype)
        cat[:, :P].copy_(tex)
        flat = ids.to(self._device, torch.float32).contiguous().view(-1)  # (a [B, 6] table of image sizes / crops given by the caller)
        for i in range(flat.numel()):  # one [1, d] sinusoid per scalar: the timestep-embedding kernel, reading the scalar on the device
            b, j = divmod(i, ids.shape[1])
            cat[b, P + j * d:P + (j + 1) * d].copy_(ops.timestep_embedding(flat[i:i + 1], 1, d, self._dtype, c.flip_sin_to_cos, c.freq_shift)[0])
        w = self._w
        a1 = ops.gemm_nt(cat, w["add1"][0], bias=w["add1"][1], act=ops.ACT_SILU)
        aug = ops.gemm_nt(a1, w["add2"][0], bias=w["add2"][1])
        if ent is None:
            ent = self._aug[B] = dict(aug=aug)
        else:
            ent["aug"].copy_(aug)  # in place: a captured graph keeps reading this buffer
        ent["key"] = key
        ent["src"] = (te, ids)  # keep the sources alive while the key is valid
        return ent["aug"]

    @_in_own_f32_mode
    def forward_packed(self, x, B, H, W, encoder_hidden_states, cfg_shared=False):
        """x: packed channels-last input [B, H*W, cin_pad]; the timestep must already be in ``_t_dev``.
        Returns float32 eps [B, out_channels, H, W].  Stream-ordered, allocation via torch only.

        ``cfg_shared``: classifier-free guidance evaluates the SAME latents twice and only the text conditioning differs
        (stable_diffusion_dual_unet.py:1045-1047 ``torch.cat([latents] * 2)``).  Nothing before the first cross-attention
        reads the conditioning, so conv_in, the first ResnetBlock2D and the first transformer's GroupNorm / proj_in / whole
        self-attention / cross-attention query are computed ONCE on the B/2 unique samples (x then has B/2 rows, B is still
        the row count of ``encoder_hidden_states`` and of the result) and duplicated there: the full-resolution
        self-attention -- the longest kernel of the forward -- runs at half the batch."""
        w = self._w
        c = self.config
        eps = c.norm_eps
        ehs = encoder_hidden_states
        # measurement hook (tools/timeline.py): device-time stamps at the block boundaries of this forward, into
        # self._stamp_buf[self._stamp_row[0], k]; None in every normal run
        sb = getattr(self, "_stamp_buf", None)
        sk = [0]

        def mark():
            if sb is not None:
                ops.stamp(sb, sk[0], self._stamp_row)
                sk[0] += 1

        mark()
        if ehs.dtype != self._dtype:
            raise HipExtensionError("encoder_hidden_states must already be in the UNet dtype (use prepare_context)")
        if cfg_shared and (B % 2 or not self.supports_cfg_shared()):
            raise HipExtensionError("cfg_shared needs an even batch and a transformer in the first down block")
        te = ops.timestep_embedding(self._t_dev, B, c.block_out_channels[0], self._dtype, c.flip_sin_to_cos, c.freq_shift)
        te = ops.gemm_nt(te, w["te1"][0], bias=w["te1"][1], act=ops.ACT_SILU)
        aug = None
        if c.addition_embed_type == "text_time":  # emb = time_embedding(t) + add_embedding(...): the sum goes through the SiLU below
            ent = self._aug.get(B)
            if ent is None:
                raise HipExtensionError("this UNet needs its added conditioning first: set_added_cond(added_cond_kwargs, batch)")
            aug = ent["aug"]
        temb = ops.gemm_nt(te, w["te2"][0], bias=w["te2"][1], residual=aug, act=ops.ACT_SILU)  # = silu(temb): the only use of temb
        temb = ops.gemm_nt(temb, w["te_all"][0], bias=w["te_all"][1], out_dtype=torch.float32)  # [B, sum(Cout)] f32
        Bc = B // 2 if cfg_shared else B  # rows currently carried (the unique half until the first cross-attention)
        x, _, _ = ops.conv3x3(x, w["conv_in"][0], Bc, H, W, bias=w["conv_in"][1], colstats=self._wants_colstats(H, W))
        skips = [(x, H, W)]
        for blk in w["down"]:
            for j, r in enumerate(blk["res"]):
                x = self._resnet(r, x, Bc, H, W, temb[:Bc], eps)  # (both halves share the timestep: temb rows are identical)
                if "attn" in blk:
                    dup = Bc != B
                    x = self._transformer(blk["attn"][j], x, Bc, H, W, ehs, cfg_dup=dup)
                    if dup:
                        skips = [(self._dup_batch(s_), h_, w_) for s_, h_, w_ in skips]
                        Bc = B
                skips.append((x, H, W))
            if "ds" in blk:
                x, H, W = ops.conv3x3(x, blk["ds"][0], B, H, W, bias=blk["ds"][1], stride=2,
                                      colstats=self._wants_colstats(H // 2, W // 2))
                skips.append((x, H, W))
            mark()  # end of a down block
        x = self._resnet(w["mid"]["r0"], x, B, H, W, temb, eps)
        x = self._transformer(w["mid"]["a"], x, B, H, W, ehs)
        x = self._resnet(w["mid"]["r1"], x, B, H, W, temb, eps)
        mark()  # end of the mid block
        for blk in w["up"]:
            for j, r in enumerate(blk["res"]):
                s, _, _ = skips.pop()
                x = self._resnet(r, ops.concat_channels(x, s), B, H, W, temb, eps)
                if "attn" in blk:
                    x = self._transformer(blk["attn"][j], x, B, H, W, ehs)
            if "us" in blk:
                x, H, W = ops.conv3x3(x, blk["us"][0], B, H, W, bias=blk["us"][1], upsample=True,
                                      colstats=self._wants_colstats(2 * H, 2 * W))
            mark()  # end of an up block
        x = ops.groupnorm(x, B, c.norm_num_groups, w["norm_out"][0], w["norm_out"][1], eps, silu=True, split_out=self._sa(w["conv_out"][0]))
        y, _, _ = ops.conv3x3(x, w["conv_out"][0], B, H, W, bias=w["conv_out"][1], out_dtype=torch.float32)
        out = ops.unpack_nchw(y, B, c.out_channels, H, W)
        mark()  # end of the forward
        return out

    @_in_own_f32_mode
    def graphed_forward(self, B, H, W, ehs, cfg_shared=False, co_run=False):
        """Capture ``forward_packed`` for this (batch, latent size) into a HIP graph (one per shape, cached).
        ``co_run``: the replays share the chip with another stream's forward (the dual-UNet pipeline's two streams): the launches are
        captured under the co-running plan family (``hip_ops.plan_family``, gmd_gemm_plan_family) -- a separate cache entry.
        Returns an object with ``.x`` (static packed-input buffer: fill it with ``pack_input(..., out=g.x)``) and
        ``.replay()`` -> float32 eps [B, out_channels, H, W] (static output buffer).  The timestep is read from the
        device scalar written by ``set_timestep``; the text conditioning from the in-place K / V^T buffers written by
        ``update_context`` -- both outside the graph, so one capture serves every step and every prompt."""
        self._ensure()
        self.update_context(ehs)
        if self.config.addition_embed_type is not None and B not in self._aug:
            raise HipExtensionError("set_added_cond(added_cond_kwargs, batch) must precede graphed_forward for this UNet")
        key = (B, H, W, tuple(ehs.shape), bool(cfg_shared), bool(co_run))
        g = self._graphs.get(key)
        if g is not None:
            return g
        from .. import profiling

        if profiling.active() is not None:
            raise HipExtensionError("graph capture with an active KernelTimer is not supported")
        g = _GraphedForward()
        g.x = torch.empty((B // 2 if cfg_shared else B, H * W, self._cin_pad), dtype=self._dtype, device=self._device)
        g.x.zero_()
        cur = torch.cuda.current_stream(self._device)
        side = torch.cuda.Stream(device=self._device)
        side.wait_stream(cur)
        with torch.cuda.stream(side), ops.plan_family(co_run):  # warm-up on a side stream: one-time attribute calls, workspaces, allocator pools
            for _ in range(2):
                self.forward_packed(g.x, B, H, W, ehs, cfg_shared=cfg_shared)
        cur.wait_stream(side)
        torch.cuda.synchronize(self._device)
        g.graph = torch.cuda.CUDAGraph()
        g.ws = ops.new_workspace(self._device)  # this graph's own split-K scratch (see hip_ops.workspace_scope)
        self._capturing = True
        try:
            with ops.capture_in_flight(), ops.workspace_scope(g.ws), ops.plan_family(co_run), torch.cuda.graph(g.graph):
                g.out = self.forward_packed(g.x, B, H, W, ehs, cfg_shared=cfg_shared)
        finally:
            self._capturing = False
        while len(self._graphs) >= self.max_cached_graphs:  # oldest first: a graph pins its activations pool and 96 MB of scratch
            self._graphs.pop(next(iter(self._graphs)))
        self._graphs[key] = g
        return g

    @_in_own_f32_mode
    def prepare_context(self, encoder_hidden_states):
        """Cast text-encoder hidden states to the UNet dtype once (HIP cast kernel)."""
        self._ensure()
        ehs = encoder_hidden_states
        if not ehs.is_cuda:
            raise HipExtensionError("encoder_hidden_states must be on the HIP device")
        ehs = ehs.contiguous()
        if ehs.dtype == self._dtype:
            return ehs
        if ehs.dtype not in (torch.float32, torch.bfloat16, torch.float16):
            ehs = ehs.float()
        return ops.cast(ehs, self._dtype)

    def pack_input(self, sample, dup=1, out=None):
        """sample: float32 NCHW tensor, or a tuple ``(cond, x)`` concatenated on channels (conditioning first:
        stable_diffusion_gm.py:1045, dual_unet.py:1080); dup=2 duplicates the batch for CFG (gm.py:1047)."""
        self._ensure()
        a, b = (sample if isinstance(sample, (tuple, list)) else (sample, None))
        if a.dtype != torch.float32:
            a = ops.cast(a.contiguous(), torch.float32)
        if b is not None and b.dtype != torch.float32:
            b = ops.cast(b.contiguous(), torch.float32)
        nch = a.shape[1] + (0 if b is None else b.shape[1])
        if nch != self.config.in_channels:
            raise ValueError(f"UNet expects {self.config.in_channels} input channels, got {nch}")
        return ops.pack_unet_input(a.contiguous(), None if b is None else b.contiguous(), dup, self._cin_pad, self._dtype, out=out)

    @_in_own_f32_mode
    def __call__(self, sample, timestep, encoder_hidden_states=None, timestep_cond=None, cross_attention_kwargs=None,
                 added_cond_kwargs=None, return_dict=True, **kwargs):
        if timestep_cond is not None:
            raise NotImplementedError("timestep_cond is not part of the GM-Diffusion path")
        self._ensure()
        first = sample[0] if isinstance(sample, (tuple, list)) else sample
        B, _, H, W = first.shape
        if H % (2 ** (len(self.config.block_out_channels) - 1)) or W % (2 ** (len(self.config.block_out_channels) - 1)):
            raise ValueError(f"latent size {H}x{W} must be divisible by {2 ** (len(self.config.block_out_channels) - 1)}")
        x = self.pack_input(sample)
        ehs = self.prepare_context(encoder_hidden_states)
        if ehs.shape[0] != B:
            raise ValueError(f"encoder_hidden_states batch {ehs.shape[0]} != sample batch {B}")
        self.set_timestep(timestep)
        self.set_added_cond(added_cond_kwargs, B)
        out = self.forward_packed(x, B, H, W, ehs)
        if not return_dict:
            return (out,)
        from .image_processor import UNetOutput

        return UNetOutput(sample=out)

    forward = __call__

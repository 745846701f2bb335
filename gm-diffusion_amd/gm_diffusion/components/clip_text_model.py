"""
CLIP text encoder on the HIP kernels -- the ``text_encoder`` component of the reference pipelines
(``self.text_encoder(text_input_ids, attention_mask=...)``, stable_diffusion_gm.py:398-439).  In the
reference this is ``transformers.CLIPTextModel``; here token + position embedding, the pre-LN transformer layers under the
causal mask (fused QKV projection, causal flash attention, quick-GELU MLP in the GEMM epilogue) and the final LayerNorm run
on libgmd_hip.so.  No CPU path.

Protocol kept from transformers (what ``encode_prompt`` uses): ``model(input_ids, attention_mask=None,
output_hidden_states=False)`` returns an object indexable like ``BaseModelOutputWithPooling`` ([0] last_hidden_state,
[1] pooler_output, [-1] hidden_states when requested), ``model.config``, ``model.dtype`` and
``model.text_model.final_layer_norm(x)`` for the ``clip_skip`` branch (stable_diffusion_gm.py:419-428).  Checkpoints: the
``text_encoder/`` folder of an SD-1.5 directory (``config.json`` + ``model.safetensors``, keys with the ``text_model.``
prefix of transformers 4.x; un-prefixed 5.x keys are accepted).
"""
from __future__ import annotations

import os
from types import SimpleNamespace

import torch

from .. import hip_ops as ops
from .._native import HipExtensionError
from .configuration import read_state_dict
from .unet_2d_condition import _HipModule, _in_own_f32_mode, composed_attention

CLIP_L_TEXT_DEFAULTS = dict(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                            num_attention_heads=12, max_position_embeddings=77, hidden_act="quick_gelu", layer_norm_eps=1e-5,
                            eos_token_id=2, bos_token_id=49406, pad_token_id=1)


class CLIPTextOutput(tuple):
    """(last_hidden_state, pooler_output[, hidden_states]) with the attribute names of ``BaseModelOutputWithPooling``."""

    last_hidden_state = property(lambda self: self[0])
    pooler_output = property(lambda self: self[1])
    hidden_states = property(lambda self: self[2] if len(self) > 2 else None)


class _FinalLayerNorm:
    def __init__(self, owner):
        self._owner = owner

    def __call__(self, x):
        o = self._owner
        o._ensure()
        g, b = o._w["final_ln"]
        return ops.layernorm(x.contiguous(), g, b, o.config.layer_norm_eps)


class CLIPTextModel(_HipModule):
    config_name = "config.json"

    def __init__(self, **config):
        cfg = dict(CLIP_L_TEXT_DEFAULTS)
        cfg.update({k: v for k, v in config.items() if k in cfg})
        if cfg["hidden_act"] not in ("quick_gelu",):
            raise NotImplementedError(f"hidden_act={cfg['hidden_act']!r}: the HIP text encoder implements quick_gelu (SD-1.5)")
        if cfg["hidden_size"] % cfg["num_attention_heads"]:
            raise ValueError("hidden_size must be a multiple of num_attention_heads")
        self.register_to_config(**cfg)
        self._init_module()
        self.text_model = SimpleNamespace(final_layer_norm=_FinalLayerNorm(self))

    # ---- weights -----------------------------------------------------------------------------------
    def expected_keys(self):
        c = self.config
        d, i = c.hidden_size, c.intermediate_size
        keys = {"text_model.embeddings.token_embedding.weight": (c.vocab_size, d),
                "text_model.embeddings.position_embedding.weight": (c.max_position_embeddings, d),
                "text_model.final_layer_norm.weight": (d,), "text_model.final_layer_norm.bias": (d,)}
        for n in range(c.num_hidden_layers):
            p = f"text_model.encoder.layers.{n}."
            for proj in ("q_proj", "k_proj", "v_proj", "out_proj"):
                keys[p + f"self_attn.{proj}.weight"] = (d, d)
                keys[p + f"self_attn.{proj}.bias"] = (d,)
            for ln in ("layer_norm1", "layer_norm2"):
                keys[p + ln + ".weight"] = (d,)
                keys[p + ln + ".bias"] = (d,)
            keys[p + "mlp.fc1.weight"] = (i, d)
            keys[p + "mlp.fc1.bias"] = (i,)
            keys[p + "mlp.fc2.weight"] = (d, i)
            keys[p + "mlp.fc2.bias"] = (d,)
        return keys

    def load_state_dict(self, sd, strict=True):
        sd = {(k if k.startswith("text_model.") else "text_model." + k): v for k, v in sd.items() if "position_ids" not in k}
        return super().load_state_dict(sd, strict=strict)

    @classmethod
    def from_pretrained(cls, path, subfolder=None, torch_dtype=None, **config_overrides):
        d = os.path.join(path, subfolder) if subfolder else path
        cfg = cls.load_config(d)
        cfg.update(config_overrides)
        m = cls(**cfg)
        m.load_state_dict(read_state_dict(d, basename="model"))
        if torch_dtype is not None:
            m.to(torch_dtype)
        return m

    def init_random(self, seed=77):
        """Synthetic weights with the initialiser scales of transformers' CLIP (normal 0.02 embeddings, scaled projections)."""
        g = torch.Generator().manual_seed(seed)
        sd = {}
        for k, shape in self.expected_keys().items():
            if k.endswith("layer_norm1.weight") or k.endswith("layer_norm2.weight") or k.endswith("final_layer_norm.weight"):
                sd[k] = torch.ones(shape)
            elif k.endswith(".bias"):
                sd[k] = torch.randn(shape, generator=g) * 0.02
            elif "embedding" in k:
                sd[k] = torch.randn(shape, generator=g) * 0.02
            else:
                sd[k] = torch.randn(shape, generator=g) * shape[1] ** -0.5
        return self.load_state_dict(sd)

    def _prepare(self):
        c, r = self.config, self._raw
        w = {"tok": self._act(r["text_model.embeddings.token_embedding.weight"]),
             "pos": self._act(r["text_model.embeddings.position_embedding.weight"]),
             "final_ln": self._norm("text_model.final_layer_norm"), "layers": []}
        for n in range(c.num_hidden_layers):
            p = f"text_model.encoder.layers.{n}."
            qkv_w = torch.cat([r[p + f"self_attn.{x}.weight"] for x in ("q_proj", "k_proj")])
            qkv_b = torch.cat([r[p + f"self_attn.{x}.bias"] for x in ("q_proj", "k_proj")])
            w["layers"].append(dict(
                ln1=self._norm(p + "layer_norm1"), ln2=self._norm(p + "layer_norm2"),
                qk=(self._wt(qkv_w), self._f32(qkv_b)),                       # one GEMM -> [tokens, 2C] = [q | k]
                v=self._wa(r[p + "self_attn.v_proj.weight"]),
                # softmax rows sum to one, so the V bias passes through the attention unchanged: fold it into out_proj
                out=(self._wt(r[p + "self_attn.out_proj.weight"]),
                     self._f32(r[p + "self_attn.out_proj.bias"] + r[p + "self_attn.out_proj.weight"] @ r[p + "self_attn.v_proj.bias"])),
                fc1=self._lin(p + "mlp.fc1"), fc2=self._lin(p + "mlp.fc2")))
        return w

    # ---- forward -----------------------------------------------------------------------------------
    def _attention(self, qk, vt, B, T):
        c = self.config
        C, H = c.hidden_size, c.num_attention_heads
        d = C // H
        scale = d ** -0.5
        if ops.is_half(self._dtype):
            return ops.attention(qk, qk, vt, H, T, scale, k_col=C, causal=True)
        return composed_attention(qk, 0, 2 * C, qk, C, 2 * C, vt, B, H, d, T, T, scale, self._dtype, causal=True)

    @_in_own_f32_mode
    def __call__(self, input_ids, attention_mask=None, position_ids=None, output_hidden_states=False, return_dict=True):
        self._ensure()
        if attention_mask is not None and not bool(torch.all(attention_mask != 0)):
            raise NotImplementedError("padding attention_mask: SD-1.5's text encoder config has no use_attention_mask "
                                      "(stable_diffusion_gm.py:409-412 passes None)")
        if position_ids is not None:
            raise NotImplementedError("explicit position_ids")
        if input_ids.device.type != "cuda":
            raise HipExtensionError("CLIPTextModel runs on hand-written HIP kernels only: input_ids must be on the GPU")
        c, w = self.config, self._w
        B, T = input_ids.shape
        if T > c.max_position_embeddings:
            raise ValueError(f"sequence length {T} exceeds max_position_embeddings={c.max_position_embeddings}")
        lo, hi = int(input_ids.min()), int(input_ids.max())
        if lo < 0 or hi >= c.vocab_size:
            raise IndexError(f"token id out of range [0, {c.vocab_size}): min={lo} max={hi}")
        C = c.hidden_size
        h = ops.embedding_lookup(input_ids, w["tok"], w["pos"]).view(B * T, C)
        states = [h.view(B, T, C)]
        mul = 64 if ops.is_half(self._dtype) else 4
        ldvt = (T + mul - 1) // mul * mul
        for L in w["layers"]:
            x = ops.layernorm(h, *L["ln1"], c.layer_norm_eps)
            qk = ops.gemm_nt(x, L["qk"][0], bias=L["qk"][1]).view(B, T, 2 * C)
            # V^T[b] = W_v x_b^T: an operand swap of the same GEMM, keys contiguous for the second attention product
            vt = torch.zeros((B, C, ldvt), dtype=self._dtype, device=h.device)
            ops.gemm_nt(L["v"], x.view(B, T, C), out=vt, ldc=ldvt)
            a = self._attention(qk, vt, B, T).view(B * T, C)
            h = ops.gemm_nt(a, L["out"][0], bias=L["out"][1], residual=h)
            x = ops.layernorm(h, *L["ln2"], c.layer_norm_eps)
            x = ops.gemm_nt(x, L["fc1"][0], bias=L["fc1"][1], act=ops.ACT_QUICK_GELU)
            h = ops.gemm_nt(x, L["fc2"][0], bias=L["fc2"][1], residual=h)
            states.append(h.view(B, T, C))
        last = ops.layernorm(h, *w["final_ln"], c.layer_norm_eps).view(B, T, C)
        ids = input_ids.to(torch.int)
        pos = ids.argmax(-1) if c.eos_token_id == 2 else (ids == c.eos_token_id).int().argmax(-1)
        pooled = last[torch.arange(B, device=last.device), pos]
        out = (last, pooled) + ((tuple(states),) if output_hidden_states else ())
        return CLIPTextOutput(out)

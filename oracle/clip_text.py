"""
Oracle: CLIP text encoder (the ``text_encoder`` of the reference pipelines, stable_diffusion_gm.py:398-439), restated in
plain torch.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference takes ``transformers.CLIPTextModel`` from its environment.  ``transformers`` IS installed in this image
(weights are not), so unlike the diffusers restatements this one is PINNED: tests/test_oracle_models.py builds the real
``transformers.CLIPTextModel`` from a config with random weights, copies its state dict in here and compares
``last_hidden_state``, ``pooler_output`` and every ``hidden_states`` entry.

State-dict keys follow the checkpoint layout of SD-1.5's ``text_encoder/`` folder (transformers 4.x: ``text_model.``
prefix); keys without the prefix (transformers 5.x modules) are accepted too.
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F


def clip_l_config():
    """openai/clip-vit-large-patch14 text tower = SD-1.5 ``text_encoder/config.json``."""
    return dict(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                max_position_embeddings=77, hidden_act="quick_gelu", layer_norm_eps=1e-5, eos_token_id=2)


def tiny_clip_config():
    return dict(vocab_size=1000, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2,
                max_position_embeddings=77, hidden_act="quick_gelu", layer_norm_eps=1e-5, eos_token_id=2)


def _act(name):
    if name == "quick_gelu":
        return lambda x: x * torch.sigmoid(1.702 * x)
    if name == "gelu":
        return F.gelu
    raise NotImplementedError(name)


class _Layer(nn.Module):
    def __init__(self, c):
        super().__init__()
        d, i = c["hidden_size"], c["intermediate_size"]
        self.self_attn = nn.Module()
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            setattr(self.self_attn, n, nn.Linear(d, d))
        self.layer_norm1 = nn.LayerNorm(d, eps=c["layer_norm_eps"])
        self.mlp = nn.Module()
        self.mlp.fc1 = nn.Linear(d, i)
        self.mlp.fc2 = nn.Linear(i, d)
        self.layer_norm2 = nn.LayerNorm(d, eps=c["layer_norm_eps"])
        self.heads = c["num_attention_heads"]
        self.act = _act(c["hidden_act"])

    def forward(self, h, mask):
        B, T, C = h.shape
        hd = C // self.heads
        x = self.layer_norm1(h)
        q = self.self_attn.q_proj(x).view(B, T, self.heads, hd).transpose(1, 2)
        k = self.self_attn.k_proj(x).view(B, T, self.heads, hd).transpose(1, 2)
        v = self.self_attn.v_proj(x).view(B, T, self.heads, hd).transpose(1, 2)
        s = q @ k.transpose(-1, -2) * hd ** -0.5 + mask
        o = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, T, C)
        h = h + self.self_attn.out_proj(o)
        return h + self.mlp.fc2(self.act(self.mlp.fc1(self.layer_norm2(h))))


class CLIPTextModel(nn.Module):
    """token + position embeddings -> N pre-LN transformer layers under a causal mask -> final LayerNorm; pooled output =
    the final hidden state at the EOS position (``eos_token_id == 2``: position of the largest id, the legacy rule)."""

    def __init__(self, **cfg):
        super().__init__()
        c = dict(clip_l_config(), **cfg)
        self.config = SimpleNamespace(**c)
        self.text_model = nn.Module()
        tm = self.text_model
        tm.embeddings = nn.Module()
        tm.embeddings.token_embedding = nn.Embedding(c["vocab_size"], c["hidden_size"])
        tm.embeddings.position_embedding = nn.Embedding(c["max_position_embeddings"], c["hidden_size"])
        tm.encoder = nn.Module()
        tm.encoder.layers = nn.ModuleList([_Layer(c) for _ in range(c["num_hidden_layers"])])
        tm.final_layer_norm = nn.LayerNorm(c["hidden_size"], eps=c["layer_norm_eps"])

    @property
    def dtype(self):
        return self.text_model.final_layer_norm.weight.dtype

    def load_state_dict(self, sd, strict=True):
        sd = {(k if k.startswith("text_model.") else "text_model." + k): v for k, v in sd.items() if "position_ids" not in k}
        return super().load_state_dict(sd, strict=strict)

    @torch.no_grad()
    def forward(self, input_ids, attention_mask=None, output_hidden_states=False):
        tm = self.text_model
        B, T = input_ids.shape
        h = tm.embeddings.token_embedding(input_ids) + tm.embeddings.position_embedding(torch.arange(T, device=input_ids.device))[None]
        mask = torch.full((T, T), float("-inf"), dtype=h.dtype, device=h.device).triu(1)[None, None]
        if attention_mask is not None:  # padding mask: 1 = attend
            mask = mask + (1.0 - attention_mask[:, None, None, :].to(h.dtype)) * torch.finfo(h.dtype).min
        states = [h]
        for layer in tm.encoder.layers:
            h = layer(h, mask)
            states.append(h)
        last = tm.final_layer_norm(h)
        ids = input_ids.to(torch.int)
        pos = ids.argmax(-1) if self.config.eos_token_id == 2 else (ids == self.config.eos_token_id).int().argmax(-1)
        pooled = last[torch.arange(B, device=last.device), pos]
        out = (last, pooled) + ((tuple(states),) if output_hidden_states else ())
        return out

"""
Tone-mapping operators and the gain-map recomposition (Eq. 1), MI355X build.  Same names,
signatures and semantics as the reference's gm_diffusion/stage1/tone_mapping.py:14-101; every
function launches a hand-written HIP kernel (csrc/hdr_tail.hip) on device tensors.  There is no
CPU path: host tensors raise ``HipExtensionError``.
"""
from __future__ import annotations

import random

import torch

from .. import hip_ops as ops


def _as_f32(x):
    if not torch.is_tensor(x):
        raise TypeError(f"expected a torch.Tensor, got {type(x)}")
    return x.contiguous() if x.dtype == torch.float32 else ops.cast(x.contiguous(), torch.float32)


def linear_scale_tmo(img: torch.Tensor, qmax: float) -> torch.Tensor:
    """``img / (qmax + 1)`` (tone_mapping.py:14-18)."""
    return ops.tmo(_as_f32(img), 0, qmax=qmax)


def hard_clip_tmo(hdr_img: torch.Tensor, qmax: float) -> torch.Tensor:
    """``clamp(hdr_img, 0, 1)``; qmax ignored (tone_mapping.py:21-26)."""
    del qmax
    return ops.tmo(_as_f32(hdr_img), 1)


def fix_mulog_tmo(hdr_img: torch.Tensor, qmax: float) -> torch.Tensor:
    """``clamp(log1p(500 * hdr/(qmax+1)) / log1p(500), 0, 1)`` (tone_mapping.py:29-36)."""
    return ops.tmo(_as_f32(hdr_img), 2, qmax=qmax, mu=500.0)


def tmo_cuda(hdr_img: torch.Tensor) -> torch.Tensor:
    """``x = clamp(hdr/10, 0, 1); log1p(5000 x)/log1p(5000)`` (tone_mapping.py:39-47).  The reference's range
    check can only fire on NaN input; it is reproduced as such."""
    x = _as_f32(hdr_img)
    out = ops.tmo(x, 3)
    if bool(torch.isnan(out).any()):
        raise ValueError("HDR image values should be in the range [0, 1]")
    return out


def random_tmo_cuda(hdr_img: torch.Tensor, qmax: float) -> torch.Tensor:
    """mu ~ U(500, 5000) from python's ``random`` (tone_mapping.py:50-57)."""
    mu = random.uniform(500, 5_000)
    return ops.tmo(_as_f32(hdr_img), 2, qmax=qmax, mu=mu)


def apply_gm_to_sdr(gm: torch.Tensor, sdr: torch.Tensor, qmax: float = 9, eps: float = 1 / 64) -> torch.Tensor:
    """Eq. 1: ``clamp((clamp(sdr,0,1)**2.2 + eps) * (1 + gm*qmax) - eps, 0, qmax+1)`` (tone_mapping.py:60-71)."""
    return ops.apply_gm_to_sdr(_as_f32(gm), _as_f32(sdr), qmax=qmax, eps=eps, clamp=True)


def gamut_compress(tmo_hdr_img: torch.Tensor) -> torch.Tensor:
    """BT.2020 -> BT.709 per-pixel 3x3 matrix then clamp(0,1), (B,C,H,W) in/out (tone_mapping.py:74-90)."""
    return ops.gamut_compress(_as_f32(tmo_hdr_img))


__all__ = [
    "linear_scale_tmo",
    "hard_clip_tmo",
    "fix_mulog_tmo",
    "tmo_cuda",
    "random_tmo_cuda",
    "apply_gm_to_sdr",
    "gamut_compress",
]

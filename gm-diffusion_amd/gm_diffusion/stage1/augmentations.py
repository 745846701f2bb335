"""
Stage-1 augmentation pieces that sit on the Stage-3 path.  Only
``RandomExposureAdjust.discretize_to_uint16`` (reference gm_diffusion/stage1/augmentations.py:38-41)
is on it (the "uint16 gain-map quantisation" of the north star); the rest of the class is
training-time data augmentation and keeps the reference's torch formulation on whatever device
the caller uses (out of scope for the HIP path).
"""
from __future__ import annotations

import random
from typing import Dict, Tuple, Union

import torch

from .. import hip_ops as ops


class RandomExposureAdjust:
    def __init__(self, gamma: float = 2.2, prob: float = 1.0):
        self.gamma = gamma
        self.prob = prob
        self.exposure_levels = torch.tensor([0.1, 0.25, 0.5, 1.0, 4.0, 8.0, 16.0], dtype=torch.float32)

    def hdr_to_ldr(self, img: torch.Tensor, exposure: float) -> torch.Tensor:
        img = torch.clamp(img * exposure, 0.0, 1.0)
        return torch.pow(img, 1.0 / self.gamma)

    @staticmethod
    def sample_camera_curve() -> Tuple[float, float]:
        n = float(torch.clamp(torch.normal(mean=0.65, std=0.1, size=()), 0.4, 0.9))
        sigma = float(torch.clamp(torch.normal(mean=0.6, std=0.1, size=()), 0.4, 0.8))
        return n, sigma

    @staticmethod
    def apply_inv_sigmoid_curve(y: torch.Tensor, n: float, sigma: float) -> torch.Tensor:
        return torch.pow((sigma * y) / (1 + sigma - y + 1e-8), 1.0 / n)

    @staticmethod
    def discretize_to_uint16(img: torch.Tensor) -> torch.Tensor:
        """``clamp(img*65535, 0, 65535).round() / 65535`` (round-half-to-even), HIP kernel, bit-exact."""
        x = img.contiguous() if img.dtype == torch.float32 else ops.cast(img.contiguous(), torch.float32)
        return ops.discretize_u16(x)

    @staticmethod
    def uint16_codes(img: torch.Tensor) -> torch.Tensor:
        """The integer codes behind :meth:`discretize_to_uint16` (uint16 tensor)."""
        x = img.contiguous() if img.dtype == torch.float32 else ops.cast(img.contiguous(), torch.float32)
        return ops.discretize_u16(x, codes=True)[1]

    def __call__(self, imgs: torch.Tensor, *, return_metadata: bool = False
                 ) -> Union[torch.Tensor, Tuple[torch.Tensor, Dict[str, float]]]:
        if random.random() > self.prob:
            return (imgs, {"exposure": 1.0, "n": 1.0, "sigma": 0.0}) if return_metadata else imgs
        exposure = float(self.exposure_levels[torch.randint(len(self.exposure_levels), (1,))])
        n, sigma = self.sample_camera_curve()
        is_batched = imgs.dim() == 4
        if imgs.dim() == 3:
            imgs = imgs.unsqueeze(0)
        if imgs.dim() != 4:
            raise ValueError("RandomExposureAdjust expects a tensor with shape (C,H,W) or (N,C,H,W)")
        if imgs.dtype != torch.float32:
            raise TypeError(f"RandomExposureAdjust expects float32 tensors, received {imgs.dtype}")
        linear_img = self.apply_inv_sigmoid_curve(imgs, n, sigma)
        linear_img = self.discretize_to_uint16(linear_img)
        ldr_img = self.hdr_to_ldr(linear_img, exposure)
        if not is_batched:
            ldr_img = ldr_img.squeeze(0)
        if return_metadata:
            return ldr_img, {"exposure": exposure, "n": n, "sigma": sigma}
        return ldr_img

    def __repr__(self) -> str:  # pragma: no cover
        return f"{self.__class__.__name__}(gamma={self.gamma}, prob={self.prob}, exposure_levels={self.exposure_levels.tolist()})"

"""
GM Diffusion -- MI355X build
============================

MI355X-native (gfx950, PyTorch-ROCm + hand-written HIP kernels + RCCL) implementation of ONE hot
path of Guanys-dar/GM-Diffusion: the Stage-3 SDR+GM denoising loop, VAE decode and gain-map HDR
recomposition, behind the reference's own Python surface (same names as the reference's
gm_diffusion/__init__.py:16-34).  ``gm_diffusion.components`` supplies the model / scheduler
classes the reference imports from ``diffusers``; ``gm_diffusion.hdr`` the fused Stage-3 tail;
``gm_diffusion.distributed`` the one-process-per-GPU prompt sharding over RCCL.
"""

from .stage1.augmentations import RandomExposureAdjust
from .stage1.tone_mapping import (
    apply_gm_to_sdr,
    gamut_compress,
    hard_clip_tmo,
    linear_scale_tmo,
    random_tmo_cuda,
    tmo_cuda,
)

__all__ = [
    "RandomExposureAdjust",
    "apply_gm_to_sdr",
    "gamut_compress",
    "hard_clip_tmo",
    "linear_scale_tmo",
    "random_tmo_cuda",
    "tmo_cuda",
]

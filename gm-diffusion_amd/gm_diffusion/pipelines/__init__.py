"""
Diffusion pipelines used during Stage 2 fine-tuning and inference (export list identical to the
reference's gm_diffusion/pipelines/__init__.py:5-19).
"""

from .stable_diffusion_gm import StableDiffusionGMPipeline
from .stable_diffusion_dual_unet import (
    StableDiffusionDualUNetPipeline,
    rescale_noise_cfg,
    retrieve_timesteps,
)
from .stable_diffusion_dual_unet_improved import StableDiffusionDualUNetImprovedPipeline

__all__ = [
    "StableDiffusionGMPipeline",
    "StableDiffusionDualUNetPipeline",
    "StableDiffusionDualUNetImprovedPipeline",
    "rescale_noise_cfg",
    "retrieve_timesteps",
]

"""
Pipeline components for the MI355X build: the classes a reference user would import from
``diffusers`` (un-vendored, absent here) to construct the Stage-3 pipelines
(scripts/inference/generate_hdr.py:13, 152-176).  Models run on hand-written HIP kernels
(no CPU path); schedulers are host-side state machines.
"""
from .autoencoder_kl import AutoencoderKL
from .clip_text_model import CLIPTextModel
from .configuration import ConfigMixin, FrozenDict
from .image_processor import StableDiffusionPipelineOutput, VaeImageProcessor, randn_tensor
from .schedulers import DDPMScheduler, DPMSolverMultistepScheduler, PNDMScheduler
from .unet_2d_condition import UNet2DConditionModel

__all__ = [
    "AutoencoderKL", "UNet2DConditionModel", "CLIPTextModel", "PNDMScheduler", "DDPMScheduler", "DPMSolverMultistepScheduler", "VaeImageProcessor",
    "StableDiffusionPipelineOutput", "randn_tensor", "FrozenDict", "ConfigMixin",
]

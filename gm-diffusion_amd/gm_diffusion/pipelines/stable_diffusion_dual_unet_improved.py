"""
``StableDiffusionDualUNetImprovedPipeline``: in the reference
(gm_diffusion/pipelines/stable_diffusion_dual_unet_improved.py:156, 1079-1083) this class differs
from ``StableDiffusionDualUNetPipeline`` by its name and five comment lines only, so here it is
the same implementation under the second name.
"""
from .stable_diffusion_dual_unet import StableDiffusionDualUNetPipeline


class StableDiffusionDualUNetImprovedPipeline(StableDiffusionDualUNetPipeline):
    pass

import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gm-diffusion_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import test_pipeline_gpu as T
from oracle import fixtures, pipelines as OP, schedulers as OS
DEV = "cuda"
ou, og = fixtures.build_unet("tiny", 4), fixtures.build_unet("tiny", 8)
pe, ne, lat = fixtures.make_inputs(2, 16, 16, cross_dim=64)
for steps in (3, 4, 5, 10):
    want = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(5))
    rs, rg = OP.dual_loop(ou, og, OS.PNDMScheduler(), pe, ne, want, steps, guidance_scale=7.5)
    for gr, ov in ((0, 0), (1, 0), (1, 1)):
        pipe = T._dual_pipe(torch.float32); pipe.set_progress_bar_config(disable=True)
        pipe.use_hip_graphs, pipe.overlap_streams = bool(gr), bool(ov)
        a = pipe(prompt_embeds=pe.to(DEV), negative_prompt_embeds=ne.to(DEV), latents=want.to(DEV), height=128, width=128, num_inference_steps=steps, output_type="latent")
        print(steps, gr, ov, "sdr %.2e gm %.2e" % (T.rms(a[0], rs), T.rms(a[1], rg)))

"""Graph / eager / two-stream equivalence stress (no syncs between pipeline calls): every pair must print [0.0, 0.0]."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gm-diffusion_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import test_pipeline_gpu as T
DEV = "cuda"
for trial in range(6):
    pipe = T._dual_pipe(torch.bfloat16)
    pipe.set_progress_bar_config(disable=True)
    g = torch.Generator().manual_seed(5)
    pe, ne = torch.randn(2, 77, 64, generator=g).to(DEV), torch.randn(2, 77, 64, generator=g).to(DEV)
    lat = torch.randn(2, 4, 16, 16, generator=g).to(DEV)
    kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, latents=lat, height=128, width=128, num_inference_steps=6, output_type="latent")
    pe2 = torch.randn(2, 77, 64, generator=g).to(DEV)
    kw2 = dict(kw, prompt_embeds=pe2)
    res = {}
    for name, gr, ov, k in (("a", 0, 0, kw), ("b", 1, 0, kw), ("c", 1, 1, kw), ("d", 1, 1, kw2), ("d2", 1, 0, kw2), ("e", 0, 0, kw2), ("e2", 0, 0, kw2), ("f", 1, 1, kw2)):
        pipe.use_hip_graphs, pipe.overlap_streams = bool(gr), bool(ov)
        o = pipe(**k)
        res[name] = o
    for x, y in (("a", "b"), ("a", "c"), ("d", "e"), ("d2", "e"), ("e", "e2"), ("d", "f"), ("d", "d2")):
        print(trial, x, y, [float((res[x][i] - res[y][i]).abs().max()) for i in (0, 1)])

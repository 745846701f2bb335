#!/usr/bin/env python
"""Wall time of the decode + recomposition tail (two VAE decodes + HDR kernels) beside one denoising loop (diagnostic)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gm-diffusion_amd"))
import torch
from gm_diffusion import hdr
from gm_diffusion.components import AutoencoderKL

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
res = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dt = torch.bfloat16
vae = AutoencoderKL(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                    down_block_types=("DownEncoderBlock2D",) * 4, up_block_types=("UpDecoderBlock2D",) * 4, norm_num_groups=32,
                    scaling_factor=0.18215).init_random(1334).to("cuda", dt)
vae._ensure()
g = torch.Generator().manual_seed(0)
sdr = torch.randn(B, 4, res // 8, res // 8, generator=g).cuda()
gm = torch.randn(B, 4, res // 8, res // 8, generator=g).cuda()
f = lambda: hdr.decode_to_hdr(vae, sdr, gm, qmax=99.0, want=("sdr_u8", "gm_u8", "hdr", "hdr_u16"))
f(); torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); f(); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"tail B={B} {res}x{res}: {(t1 - t0) * 1e3:.2f} ms wall")
from gm_diffusion import profiling
tm = profiling.KernelTimer()
profiling.set_timer(tm)
f(); torch.cuda.synchronize()
profiling.set_timer(None)
for k, v in tm.summary().items():
    print(f"{k:10s} launches {v['launches']:4d}  {v['ms']:8.3f} ms  avg {v['avg_us']:8.1f} us  {v['tflops']:7.1f} TF/s  {v['gbps']:7.1f} GB/s")

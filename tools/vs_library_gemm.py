#!/usr/bin/env python
"""How far are the hand-written GEMM / conv kernels from the vendor library on the pipeline's shapes?  torch.nn.functional.linear /
conv2d (hipBLASLt / rocBLAS / MIOpen through PyTorch) against gmd_gemm_nt / gmd_conv3x3, bf16, device time per launch with
rotating operands.  A measurement tool: the product path never calls torch operators."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gm-diffusion_amd"))
import torch
import torch.nn.functional as F
from gm_diffusion import hip_ops as ops


def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    torch.cuda._sleep(int(1e7))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


g = torch.Generator().manual_seed(0)
print("GEMM  (y = x W^T + b [+ residual])")
for M, N, K in [(32768, 320, 320), (32768, 960, 320), (32768, 2560, 320), (32768, 320, 1280), (8192, 640, 640), (8192, 5120, 640), (8192, 640, 2560),
                (2048, 1280, 1280), (2048, 10240, 1280), (2048, 1280, 5120), (1024, 1280, 1280), (16384, 320, 320)]:
    xs = [torch.randn(M, K, generator=g).bfloat16().cuda() for _ in range(3)]
    w = (torch.randn(N, K, generator=g) * 0.02).bfloat16().cuda()
    b32 = torch.randn(N, generator=g).cuda()
    b16 = b32.bfloat16()
    i = [0]

    def mine():
        i[0] = (i[0] + 1) % 3
        return ops.gemm_nt(xs[i[0]], w, bias=b32)

    def lib():
        i[0] = (i[0] + 1) % 3
        return F.linear(xs[i[0]], w, b16)

    tm, tl = timeit(mine), timeit(lib)
    fl = 2.0 * M * N * K / 1e6
    print(f"  M={M:6d} N={N:5d} K={K:5d}: this repo {tm:7.1f} us ({fl / tm:5.0f} TF/s)   torch/hipBLASLt {tl:7.1f} us ({fl / tl:5.0f} TF/s)", flush=True)
print("conv3x3 (channels-last bf16)")
for B, H, ci, co in [(8, 64, 320, 320), (8, 64, 640, 320), (8, 32, 640, 640), (8, 32, 1280, 640), (8, 16, 1280, 1280), (8, 8, 1280, 1280), (4, 64, 320, 320)]:
    x = torch.randn(B, H * H, ci, generator=g).bfloat16().cuda()
    w = (torch.randn(co, 9 * ci, generator=g) * 0.02).bfloat16().cuda()
    b32 = torch.randn(co, generator=g).cuda()
    xt = x.view(B, H, H, ci).permute(0, 3, 1, 2)  # NCHW view of channels-last memory
    wt = w.view(co, 3, 3, ci).permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
    b16 = b32.bfloat16()
    tm = timeit(lambda: ops.conv3x3(x, w, B, H, H, bias=b32))
    tl = timeit(lambda: F.conv2d(xt, wt, b16, padding=1))
    fl = 2.0 * B * H * H * co * 9 * ci / 1e6
    print(f"  B={B} {H}x{H} {ci:4d}->{co:4d}: this repo {tm:7.1f} us ({fl / tm:5.0f} TF/s)   torch/MIOpen {tl:7.1f} us ({fl / tl:5.0f} TF/s)", flush=True)

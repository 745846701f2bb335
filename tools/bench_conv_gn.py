#!/usr/bin/env python
"""conv3x3 -> GroupNorm(+SiLU) on the split-K levels: reduce + GroupNorm launches against the GroupNorm-from-slabs kernel
(gmd_conv3x3_groupnorm), device time per pair of calls.  GMD_ONE_DTYPE=bf16|f16|f32."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gm-diffusion_amd"))
import torch
from gm_diffusion import hip_ops as ops

dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[os.environ.get("GMD_ONE_DTYPE", "bf16")]


def timeit(fn, reps=60):
    fn(); fn(); torch.cuda.synchronize()
    torch.cuda._sleep(int(2e7))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


g = torch.Generator().manual_seed(0)
for B, H, Cin, Cout in [(8, 16, 1280, 1280), (4, 16, 1280, 1280), (8, 16, 640, 1280), (8, 8, 1280, 1280), (4, 8, 1280, 1280), (8, 16, 2560, 1280)]:
    x = torch.randn(B, H * H, Cin, generator=g).to("cuda", dt)
    w = (torch.randn(Cout, 9 * Cin, generator=g) * 0.01).to("cuda", dt)
    if dt == torch.float32:
        w = ops.split_weights(w)
    bias, temb = torch.randn(Cout, generator=g).cuda(), torch.randn(B, Cout, generator=g).cuda()
    ga, be = torch.randn(Cout, generator=g).cuda(), torch.randn(Cout, generator=g).cuda()
    conv_only = timeit(lambda: ops.conv3x3(x, w, B, H, H, bias=bias, rowbias=temb))

    def two():
        y, _, _ = ops.conv3x3(x, w, B, H, H, bias=bias, rowbias=temb)
        return ops.groupnorm(y, B, 32, ga, be, 1e-5, True)

    ops.USE_CONV_GN_FUSION = False
    t_two = timeit(two)
    ops.USE_CONV_GN_FUSION = True
    t_f = timeit(lambda: ops.conv3x3_groupnorm(x, w, B, H, H, 32, ga, be, 1e-5, silu=True, bias=bias, rowbias=temb))
    print(f"B={B} {H}x{H} {Cin}->{Cout}: conv (with reduce) {conv_only:6.1f} us; conv + GroupNorm {t_two:6.1f} us; fused {t_f:6.1f} us")

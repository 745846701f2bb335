#!/usr/bin/env python
"""Run one conv3x3 / gemm shape a few times (target for rocprofv3 --pmc)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gm-diffusion_amd")):
    sys.path.insert(0, p)
import torch
from gm_diffusion import hip_ops as ops
from gm_diffusion._native import lib
lib().gmd_gemm_plan_family(int(os.environ.get("GMD_ONE_FAMILY", "0")))  # 1 = the co-running plan family of the dual pipeline
B, H, W, ci, co = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (8, 64, 64, 320, 320))]
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
g = torch.Generator().manual_seed(0)
DT = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[os.environ.get("GMD_ONE_DTYPE", "bf16")]  # f32 = split path, pre-split W
x = torch.randn(B, H * W, ci, generator=g).to(DT).cuda()
w = (torch.randn(co, 9 * ci, generator=g) * 0.02).to(DT).cuda()
if DT == torch.float32:
    w = ops.split_weights(w)
b = torch.randn(co, generator=g).cuda()
for _ in range(reps):
    ops.conv3x3(x, w, B, H, W, bias=b)
torch.cuda.synchronize()
print("done")

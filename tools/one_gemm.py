import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gm-diffusion_amd"))
import torch
from gm_diffusion import hip_ops as ops
from gm_diffusion._native import lib
lib().gmd_gemm_plan_family(int(os.environ.get("GMD_ONE_FAMILY", "0")))  # 1 = the co-running plan family of the dual pipeline
M, N, K = [int(v) for v in sys.argv[1:4]]
mode = sys.argv[4] if len(sys.argv) > 4 else "bias"
g = torch.Generator().manual_seed(0)
DT = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[os.environ.get("GMD_ONE_DTYPE", "bf16")]  # f32 = split path, pre-split W
a = torch.randn(M, K, generator=g).to(DT).cuda(); w = (torch.randn(N, K, generator=g) * 0.02).to(DT).cuda(); b = torch.randn(N, generator=g).cuda()
if DT == torch.float32: w = ops.split_weights(w)
kw = dict(bias=b) if mode == "bias" else {}
if mode == "f32": kw = dict(out_dtype=torch.float32)
if mode == "geglu": kw = dict(bias=b, act=ops.ACT_GEGLU)          # the ff1 projection (N = 8C, output [M, N/2])
if mode == "res": kw = dict(bias=b, residual=torch.randn(M, N, generator=g).to(DT).cuda())  # o1 / o2 / pout / ff2
for _ in range(3): ops.gemm_nt(a, w, **kw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.gemm_nt(a, w, **kw)
e1.record(); torch.cuda.synchronize()
print("GEMM", os.environ.get("GMD_LIB_OVERRIDE", "prod")[-8:], M, N, K, mode, "%.1f us" % (e0.elapsed_time(e1) / 20 * 1e3))

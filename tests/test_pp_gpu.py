"""The ping-pong GEMM kernel (csrc/gemm.hip: gemm_pp_kernel -- one workgroup of 8 consumer + 4 loader waves on a 256-row tile,
plan code 283) through the C ABI: the shapes the heuristic hands it, and -- through the debug plan override -- the corners of its
addressing (ragged M / N, stride 2, fused upsample, asymmetric pad, K slices, row bias across a sample seam, both 16-bit types,
GEGLU pairs, producer column statistics) against float64 references computed by torch on the host."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture
def force_plan():
    """gmd_gemm_plan_override is refused unless the process has GMD_TUNING=1 (include/gmd_hip.h)."""
    from gm_diffusion._native import lib

    prev = os.environ.get("GMD_TUNING")
    os.environ["GMD_TUNING"] = "1"

    def force(bm, bn, pf, ks):
        assert lib().gmd_gemm_plan_override(bm, bn, pf, ks) == 0

    yield force
    lib().gmd_gemm_plan_override(0, 0, 0, 0)
    lib().gmd_conv_patch_override(0)  # the default (per-tap implicit GEMM since the end of round 5)
    if prev is None:
        os.environ.pop("GMD_TUNING", None)
    else:
        os.environ["GMD_TUNING"] = prev


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def _conv_ref(x, w, B, H, W, bias=None, rowbias=None, residual=None, stride=1, upsample=False, pad_mode=0):
    ci, co = x.shape[-1], w.shape[0]
    xi = x.double().view(B, H, W, ci).permute(0, 3, 1, 2)
    wt = w.double().view(co, 3, 3, ci).permute(0, 3, 1, 2)
    if upsample:
        xi = F.interpolate(xi, scale_factor=2, mode="nearest")
    if pad_mode == 1:
        y = F.conv2d(F.pad(xi, (0, 1, 0, 1)), wt, stride=2)
    else:
        y = F.conv2d(xi, wt, stride=stride, padding=1)
    y = y.permute(0, 2, 3, 1).reshape(B, -1, co)
    if bias is not None:
        y = y + bias.double()
    if rowbias is not None:
        y = y + rowbias.double()[:, None, :]
    if residual is not None:
        y = y + residual.double()
    return y


@pytest.mark.parametrize("M,N,K,geglu,expect", [
    (32768, 320, 320, False, (256, 160, 283, 1)),    # level-0 projections at batch 8: 256 ping-pong tiles
    (32768, 320, 2880, False, (256, 160, 283, 1)),   # level-0 convolutions at batch 8
    (8192, 1280, 640, False, (256, 160, 283, 1)),    # level-1 fused qk
    (131072, 512, 4608, False, (256, 128, 283, 1)),  # VAE decoder 128x128 512->512 at batch 8
    (2048, 10240, 1280, True, (256, 128, 283, 1)),   # GEGLU projections from K = 1280 up
    (8192, 5120, 640, True, (256, 128, 283, 1)),
    (16384, 320, 2880, False, (128, 160, 244, 1)),   # level-0 convolutions at batch 4: 256 tiles of 128 x 160, one per CU
    (8192, 640, 5760, False, (128, 160, 244, 1)),    # level-1 convolutions at batch 8
    (4096, 640, 5760, False, (64, 160, 244, 1)),     # ... at batch 4: 64-row tiles
    (2048, 1280, 1280, False, (64, 160, 244, 1)),    # level-2 projections at batch 8
    (2048, 1280, 11520, False, (128, 160, 244, 2)),  # level-2 convolutions: deep K -> two slices of 128-row tiles beat 64-row tiles
    (1024, 1280, 5120, False, (128, 160, 244, 4)),   # feed-forward output at batch 4, level 2
    (512, 1280, 11520, False, (128, 160, 244, 7)),   # 8x8 convolutions at batch 8: 32 tiles x 7 slices
    (1024, 1280, 1280, False, (64, 128, 244, 1)),    # GM UNet 16x16 projections at batch 4: 160 tiles of 64 x 128 (tools/dbg/sweep_rows.py: 13.7 -> 11.2 us)
])
def test_default_policy_picks_the_fastest_plan_launch_by_launch(M, N, K, geglu, expect):
    """make_plan's default rules (csrc/gemm.hip): ping-pong tiles where >= 256 of them exist, the loader/consumer kernel at about
    one tile per CU (tools/sweep_pp.py, sweep_lc.py, check_ring.py)."""
    from gm_diffusion import hip_ops as ops

    assert ops.gemm_plan_info(torch.bfloat16, M, N, K, 1, geglu) == expect


@pytest.mark.parametrize("M,N,K,geglu", [(1024, 640, 640, False), (512, 1280, 1280, False), (16384, 2560, 320, True), (4096, 200, 640, False)])
def test_default_policy_keeps_the_ring_kernels_elsewhere(M, N, K, geglu):
    from gm_diffusion import hip_ops as ops

    assert ops.gemm_plan_info(torch.bfloat16, M, N, K, 1, geglu)[2] == 0


@pytest.mark.parametrize("bm,bn,ks,M,N,K", [(128, 160, 1, 8192, 640, 640), (64, 160, 1, 2048, 1280, 1280), (128, 128, 2, 1000, 384, 1280),
                                            (64, 128, 3, 520, 256, 1920), (128, 160, 4, 1024, 1280, 5120)])
def test_lc_kernel_vs_float64(bm, bn, ks, M, N, K, force_plan):
    """gemm_lc_kernel (plan code 244; selected by GMD_PP=i, the launch-by-launch rules): 4 consumer waves with double-buffered
    fragments + 4 loader waves, 4-stage ring.  Full and ragged tiles, K slices, residual."""
    from gm_diffusion import hip_ops as ops

    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).bfloat16().to(DEV)
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16().to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    r = torch.randn(M, N, generator=g).bfloat16().to(DEV)
    force_plan(bm, bn, 244, ks)
    y = ops.gemm_nt(a, w, bias=b, residual=r)
    assert _rel(y, a.double() @ w.double().T + b.double() + r.double()) < 4e-3


@pytest.mark.parametrize("kw", [dict(), dict(stride=2), dict(upsample=True)])
def test_lc_kernel_conv3x3_vs_float64(kw, force_plan):
    from gm_diffusion import hip_ops as ops

    g = torch.Generator().manual_seed(len(kw) + 21)
    B, H, W, ci, co = 3, 24, 20, 128, 320
    x = torch.randn(B, H * W, ci, generator=g).bfloat16().to(DEV)
    w = (torch.randn(co, 9 * ci, generator=g) * 0.03).bfloat16().to(DEV)
    b = torch.randn(co, generator=g).to(DEV)
    tb = torch.randn(B, co, generator=g).to(DEV)
    for bm in (128, 64):
        force_plan(bm, 160, 244, 1)
        y, _, _ = ops.conv3x3(x, w, B, H, W, bias=b, rowbias=tb, **kw)
        assert _rel(y, _conv_ref(x, w, B, H, W, bias=b, rowbias=tb, **kw)) < 4e-3


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 4e-3), (torch.float16, 5e-4)])
@pytest.mark.parametrize("M,N,K", [(32768, 320, 320), (8192, 1280, 640), (4096, 320, 1280)])
def test_pp_gemm_with_bias_and_residual_vs_float64(M, N, K, dtype, tol, force_plan):
    from gm_diffusion import hip_ops as ops

    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(dtype).to(DEV)
    w = (torch.randn(N, K, generator=g) * 0.05).to(dtype).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    r = torch.randn(M, N, generator=g).to(dtype).to(DEV)
    if ops.gemm_plan_info(dtype, M, N, K)[2] != 283:
        force_plan(256, 160, 283, 1)
    y = ops.gemm_nt(a, w, bias=b, residual=r)
    ref = a.double() @ w.double().T + b.double() + r.double()
    assert _rel(y, ref) < tol


@pytest.mark.parametrize("bn,ks", [(160, 1), (128, 1), (160, 2), (128, 3)])
def test_pp_ragged_edges_and_k_slices_vs_float64(bn, ks, force_plan):
    """M and N not multiples of the tile (register epilogue, zero-filled out-of-range rows), K slices with the fixed-order
    reduction, 5 K steps over 3 slices (the last slice has fewer steps)."""
    from gm_diffusion import hip_ops as ops

    g = torch.Generator().manual_seed(bn + ks)
    M, N, K = 1000, 328, 320
    a = torch.randn(M, K, generator=g).bfloat16().to(DEV)
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16().to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    force_plan(256, bn, 283, ks)
    y = ops.gemm_nt(a, w, bias=b, act=ops.ACT_SILU)
    ref = F.silu(a.double() @ w.double().T + b.double())
    assert y.shape == (M, N) and _rel(y, ref) < 4e-3
    yf = ops.gemm_nt(a, w, bias=b, out_dtype=torch.float32)  # float32 output straight from the accumulators
    assert _rel(yf, a.double() @ w.double().T + b.double()) < 1e-5 + (3e-3 if ks > 1 else 0)


@pytest.mark.parametrize("kw", [dict(), dict(stride=2), dict(upsample=True), dict(stride=2, pad_mode=1)])
@pytest.mark.parametrize("bn", [160, 128])
def test_pp_conv3x3_variants_vs_float64(kw, bn, force_plan):
    """Implicit-GEMM addressing of the loader waves: zero padding, stride 2, the VAE's (0,1,0,1) pad, the fused nearest-2x upsample,
    a per-sample row bias (time embedding) whose sample seam falls inside a 256-row tile, residual; odd feature-map sizes (the
    division path of the pixel decomposition)."""
    from gm_diffusion import hip_ops as ops

    g = torch.Generator().manual_seed(bn + len(kw))
    B, H, W, ci, co = 3, 24, 20, 128, 320
    x = torch.randn(B, H * W, ci, generator=g).bfloat16().to(DEV)
    w = (torch.randn(co, 9 * ci, generator=g) * 0.03).bfloat16().to(DEV)
    b = torch.randn(co, generator=g).to(DEV)
    tb = torch.randn(B, co, generator=g).to(DEV)
    force_plan(256, bn, 283, 1)
    y, ho, wo = ops.conv3x3(x, w, B, H, W, bias=b, rowbias=tb, **kw)
    ref = _conv_ref(x, w, B, H, W, bias=b, rowbias=tb, **kw)
    assert y.shape == ref.shape and _rel(y, ref) < 4e-3
    r = torch.randn(B, ho * wo, co, generator=g).bfloat16().to(DEV)
    y2, _, _ = ops.conv3x3(x, w, B, H, W, bias=b, residual=r, **kw)
    assert _rel(y2, _conv_ref(x, w, B, H, W, bias=b, residual=r, **kw)) < 4e-3


def test_pp_conv3x3_channel_block_order_and_two_k_slices_equal_unsplit(force_plan):
    """64x64, Cin = 640: the host picks a channel-block K order (cblk 320); two K slices must give the unsplit result up to the
    float32 summation order, and both the float64 convolution to bf16 rounding."""
    from gm_diffusion import hip_ops as ops

    g = torch.Generator().manual_seed(5)
    B, H, ci, co = 2, 64, 640, 320
    x = torch.randn(B, H * H, ci, generator=g).bfloat16().to(DEV)
    w = (torch.randn(co, 9 * ci, generator=g) * 0.02).bfloat16().to(DEV)
    b = torch.randn(co, generator=g).to(DEV)
    force_plan(256, 160, 283, 1)
    y1, _, _ = ops.conv3x3(x, w, B, H, H, bias=b)
    force_plan(256, 160, 283, 2)
    y2, _, _ = ops.conv3x3(x, w, B, H, H, bias=b)
    ref = _conv_ref(x, w, B, H, H, bias=b)
    assert _rel(y1, ref) < 4e-3 and _rel(y2, ref) < 4e-3
    assert float((y1.float() - y2.float()).abs().max()) <= 2 * float(ref.abs().max()) * 2 ** -8


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 6e-3), (torch.float16, 8e-4)])
def test_pp_geglu_epilogue_vs_float64(dtype, tol, force_plan):
    """value * gelu_erf(gate) on 16-row interleaved [value | gate] weight rows (GMD_ACT_GEGLU), written as [M, N/2]."""
    from gm_diffusion import hip_ops as ops

    g = torch.Generator().manual_seed(11)
    M, C = 2048, 320
    x = torch.randn(M, C, generator=g).to(dtype).to(DEV)
    w1 = (torch.randn(8 * C, C, generator=g) * 0.05).to(dtype)
    b1 = torch.randn(8 * C, generator=g) * 0.5
    half = 4 * C  # [16 value rows | 16 gate rows] groups: the layout of GMD_ACT_GEGLU
    wi = torch.stack([w1[:half].reshape(half // 16, 16, -1), w1[half:].reshape(half // 16, 16, -1)], 1).reshape(2 * half, -1).contiguous().to(DEV)
    bi = torch.stack([b1[:half].reshape(half // 16, 16), b1[half:].reshape(half // 16, 16)], 1).reshape(2 * half).contiguous().to(DEV)
    force_plan(256, 128, 283, 1)
    y = ops.gemm_nt(x, wi, bias=bi, act=ops.ACT_GEGLU)
    h = x.double().cpu() @ w1.double().T + b1.double()
    val, gate = h[:, : 4 * C], h[:, 4 * C:]
    ref = val * F.gelu(gate)
    assert y.shape == (M, 4 * C) and _rel(y.cpu(), ref) < tol


@pytest.mark.parametrize("M,C", [(1536, 640), (4096, 640), (1280, 1280), (8192, 640)])
def test_tile_groups_of_four_m_panels_geglu_vs_float64(M, C):
    """Round 5: the XCD-aware tile order walks W-heavy launches in groups of at most FOUR M-panels (pick_tile_group, csrc/gemm.hip:
    the group's A panels have to stay in the XCD's L2).  The GEGLU projections of the co-running plan family on row counts that give
    one whole group, several groups, and a short last group (6 and 5 M-panels), against float64: every tile exactly once."""
    from gm_diffusion import hip_ops as ops

    g = torch.Generator().manual_seed(M + C)
    x = torch.randn(M, C, generator=g).bfloat16().to(DEV)
    w1 = (torch.randn(8 * C, C, generator=g) * 0.04).bfloat16()
    b1 = torch.randn(8 * C, generator=g) * 0.5
    half = 4 * C
    wi = torch.stack([w1[:half].reshape(half // 16, 16, -1), w1[half:].reshape(half // 16, 16, -1)], 1).reshape(2 * half, -1).contiguous().to(DEV)
    bi = torch.stack([b1[:half].reshape(half // 16, 16), b1[half:].reshape(half // 16, 16)], 1).reshape(2 * half).contiguous().to(DEV)
    with ops.plan_family(1):
        assert ops.gemm_plan_info(torch.bfloat16, M, 8 * C, C, 1, True)[:3] == (256, 128, 283)
        y = ops.gemm_nt(x, wi, bias=bi, act=ops.ACT_GEGLU)
    h = x.double().cpu() @ w1.double().T + b1.double()
    ref = h[:, :half] * F.gelu(h[:, half:])
    assert y.shape == (M, half) and _rel(y.cpu(), ref) < 6e-3
    # ... and a plain W-heavy projection with residual through the same order (N > M: the rule that used to take all M-panels)
    w2 = (torch.randn(4 * C, C, generator=g) * 0.04).bfloat16().to(DEV)
    r = torch.randn(M, 4 * C, generator=g).bfloat16().to(DEV)
    with ops.plan_family(1):
        y2 = ops.gemm_nt(x, w2, residual=r)
    assert _rel(y2.cpu(), x.double().cpu() @ w2.double().cpu().T + r.double().cpu()) < 4e-3


def test_pp_column_statistics_feed_groupnorm(force_plan):
    """Producer statistics out of the ping-pong kernel's row epilogue (same strips and bucket layout as the ring kernels): sums of
    the STORED values per 64 rows x 10 channels, and GroupNorm from them == GroupNorm of the stored tensor."""
    from gm_diffusion import hip_ops as ops

    g = torch.Generator().manual_seed(3)
    B, H, ci, co = 8, 64, 320, 320
    x = torch.randn(B, H * H, ci, generator=g).bfloat16().to(DEV)
    w = (torch.randn(co, 9 * ci, generator=g) * 0.02).bfloat16().to(DEV)
    b = torch.randn(co, generator=g).to(DEV)
    assert ops.gemm_plan_info(torch.bfloat16, B * H * H, co, 9 * ci)[2] == 283
    y, _, _ = ops.conv3x3(x, w, B, H, H, bias=b, colstats=True)
    st, _ = y._colstats
    yd = y.double().view(B * H * H // 64, 64, co // ops.COLSTATS_BUCKET, ops.COLSTATS_BUCKET)
    assert float((st[..., 0].double().cpu() - yd.sum((1, 3)).cpu()).abs().max()) < 2e-3
    assert float(((st[..., 1].double().cpu() - (yd * yd).sum((1, 3)).cpu()).abs() / (yd * yd).sum((1, 3)).cpu()).max()) < 1e-5
    ga, be = torch.randn(co, generator=g).to(DEV), torch.randn(co, generator=g).to(DEV)
    a = ops.groupnorm(y, B, 32, ga, be, 1e-5, silu=True)               # consumes the producer statistics
    y_plain = y.clone()                                                  # no statistics attached: two-launch path
    bref = ops.groupnorm(y_plain, B, 32, ga, be, 1e-5, silu=True)
    assert float((a.float() - bref.float()).abs().max()) <= 2 ** -6


@pytest.mark.parametrize("B,H,W,ci,co,ks", [
    (2, 64, 64, 128, 320, 1),    # 4 image rows per tile, 6 x 66-pixel patch (NPP = 13), two 64-channel blocks (patch double buffer)
    (1, 64, 64, 64, 160, 1),     # a single block: no next patch
    (3, 32, 32, 192, 320, 1),    # 8 rows per tile, 10 x 34 patch (NPP = 11), three blocks
    (3, 16, 16, 320, 128, 1),    # one image per tile, 18 x 18 patch; 256 x 128 tile
    (6, 8, 8, 128, 320, 1),      # four images per tile (4 x 10 x 10 = 400 pixels); the second tile is half empty (M = 384)
    (8, 8, 8, 320, 160, 2),      # K slices in whole 64-channel blocks: 5 blocks over 2 slices (3 + 2)
    (2, 32, 32, 320, 320, 5),    # one block per slice
    (4, 64, 32, 128, 160, 1),    # H != W
])
@pytest.mark.parametrize("mode", [2, 1])
def test_conv_patch_kernel_vs_float64(B, H, W, ci, co, ks, mode, force_plan):
    """conv_patch_kernel: the input patch of a 256-pixel tile (+ halo) resident in LDS, nine taps read it at shifted rows.  Every
    geometry class of its tile -> patch mapping, with bias, per-sample row bias and residual, against the float64 convolution;
    and bit-identical to the per-tap implicit GEMM of gemm_pp_kernel (same tiles, same K order within a tap? no: the K order
    differs -- channel block outer -- so the comparison is to rounding)."""
    from gm_diffusion import hip_ops as ops

    g = torch.Generator().manual_seed(B * 1000 + H + ci)
    x = torch.randn(B, H * W, ci, generator=g).bfloat16().to(DEV)
    w = (torch.randn(co, 9 * ci, generator=g) * 0.03).bfloat16().to(DEV)
    b = torch.randn(co, generator=g).to(DEV)
    tb = torch.randn(B, co, generator=g).to(DEV)
    r = torch.randn(B, H * W, co, generator=g).bfloat16().to(DEV)
    from gm_diffusion._native import lib

    force_plan(256, 160 if co % 160 == 0 else 128, 283, ks)
    assert lib().gmd_conv_patch_override(mode) == 0  # 2: continuous consumers (the default until the end of round 5), 1: ping-pong consumers
    y, ho, wo = ops.conv3x3(x, w, B, H, W, bias=b, rowbias=tb, residual=r)
    ref = _conv_ref(x, w, B, H, W, bias=b, rowbias=tb, residual=r)
    assert (ho, wo) == (H, W) and _rel(y, ref) < 4e-3
    # the same launch through the per-tap implicit GEMM of gemm_pp_kernel: same tiles, another K order
    assert lib().gmd_conv_patch_override(0) == 0
    y2, _, _ = ops.conv3x3(x, w, B, H, W, bias=b, rowbias=tb, residual=r)
    assert float((y.float() - y2.float()).abs().max()) <= 4 * float(ref.abs().max()) * 2 ** -8


def test_conv_patch_kernel_f16_and_column_statistics(force_plan):
    from gm_diffusion import hip_ops as ops

    g = torch.Generator().manual_seed(77)
    B, H, ci, co = 4, 32, 128, 320
    x = torch.randn(B, H * H, ci, generator=g).half().to(DEV)
    w = (torch.randn(co, 9 * ci, generator=g) * 0.03).half().to(DEV)
    b = torch.randn(co, generator=g).to(DEV)
    from gm_diffusion._native import lib
    force_plan(256, 160, 283, 1)
    assert lib().gmd_conv_patch_override(2) == 0  # (the patch-resident kernel: not the default any more)
    y, _, _ = ops.conv3x3(x, w, B, H, H, bias=b)
    assert _rel(y, _conv_ref(x, w, B, H, H, bias=b)) < 5e-4
    lib().gmd_gemm_plan_override(0, 0, 0, 0)
    # heuristic path with producer statistics (M = 32768 at 64x64 x 8: ping-pong tiles, unsplit)
    B, H, ci, co = 8, 64, 64, 320
    x = torch.randn(B, H * H, ci, generator=g).bfloat16().to(DEV)
    w = (torch.randn(co, 9 * ci, generator=g) * 0.03).bfloat16().to(DEV)
    assert ops.gemm_plan_info(torch.bfloat16, B * H * H, co, 9 * ci) == (256, 160, 283, 1)
    y, _, _ = ops.conv3x3(x, w, B, H, H, bias=b, colstats=True)
    st, _ = y._colstats
    yd = y.double().view(B * H * H // 64, 64, co // ops.COLSTATS_BUCKET, ops.COLSTATS_BUCKET)
    assert float((st[..., 0].double().cpu() - yd.sum((1, 3)).cpu()).abs().max()) < 2e-3
    assert _rel(y, _conv_ref(x, w, B, H, H, bias=b)) < 4e-3


@pytest.mark.parametrize("M,N,K,geglu", [
    (8192, 5120, 640, True),     # weights (6.5 MB) beyond an XCD's L2: M-panels walked in groups of 4, one N tile at a time
    (2304, 5120, 640, False),    # 9 M-panels in groups of 2: the last group is short
    (2048, 10240, 1280, True),   # N > M: m fastest (each weight tile goes to one XCD)
    (600, 1280, 320, False),     # N > M with a ragged last M tile
])
def test_grouped_tile_orders_cover_every_tile_once(M, N, K, geglu, force_plan):
    """The L2-aware tile orders of the round-4 kernels (GemmParams::tile_group) are pure permutations of the tile grid: results must
    equal the float64 product whatever the order (a tile visited twice or never shows as a wrong or stale block)."""
    from gm_diffusion import hip_ops as ops

    g = torch.Generator().manual_seed(M + N)
    a = torch.randn(M, K, generator=g).bfloat16().to(DEV)
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16().to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    for plan in ((256, 128, 283, 1), (128, 128, 244, 1)):
        force_plan(*plan)
        if geglu:
            y = ops.gemm_nt(a, w, bias=b, act=ops.ACT_GEGLU).double().cpu()
            h = a.double().cpu() @ w.double().cpu().T + b.double().cpu()
            hv = h.view(M, N // 32, 2, 16)  # [16 value | 16 gate] column groups
            ref = (hv[:, :, 0, :] * F.gelu(hv[:, :, 1, :])).reshape(M, N // 2)
            assert y.shape == ref.shape and _rel(y, ref) < 8e-3
        else:
            y = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
            ops.gemm_nt(a, w, bias=b, out=y)
            assert torch.isfinite(y).all() and _rel(y, a.double() @ w.double().T + b.double()) < 4e-3


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,tokens,C", [(8, 4096, 320), (8, 1024, 640), (8, 256, 1280), (4, 4096, 320), (4, 1024, 640), (4, 256, 1280), (8, 64, 1280), (2, 64, 1280)])
def test_fused_qkv_projection_writes_v_transposed(dt, B, tokens, C):
    """gmd_gemm_qkv_vt (round 4): ONE launch gives Q|K row-major and V transposed ([B, C, tokens], what gmd_attention reads) -- against the
    two launches it replaces (fused q|k projection + batched V^T GEMM) and against float64.  Where the plan cannot write transposed
    tiles the op says so (None) and the caller keeps the two launches."""
    from gm_diffusion import hip_ops as o
    from gm_diffusion._native import lib

    g = torch.Generator().manual_seed(B + tokens + C)
    n = torch.randn(B * tokens, C, generator=g).to(DEV, dt)
    wq, wk, wv = ((torch.randn(C, C, generator=g) * C ** -0.5).to(DEV, dt) for _ in range(3))
    r = o.gemm_qkv_vt(n, torch.cat([wq, wk, wv], 0).contiguous(), 2 * C, tokens)
    qk_ref = o.gemm_nt(n, torch.cat([wq, wk], 0).contiguous())
    vt_ref = o.gemm_nt(wv, n.view(B, tokens, C), ldc=tokens)
    if r is None:
        assert not lib().gmd_gemm_qkv_vt_ok(o.dtype_code(dt), B * tokens, 3 * C, C, 2 * C, tokens, o.WORKSPACE_BYTES)
        return
    qk, vt = r
    assert qk.shape == qk_ref.shape and vt.shape == vt_ref.shape == (B, C, tokens)
    vd = (wv.double() @ n.double().view(B, tokens, C).transpose(1, 2))
    tol = 2e-2 if dt == torch.bfloat16 else 3e-3
    assert float((vt.double() - vd).abs().max()) <= tol * float(vd.abs().max())
    # the same products in the same K order as the launches it replaces: equal up to the rounding of one ulp where a tile shape differs
    assert float((qk.float() - qk_ref.float()).abs().max()) <= tol * float(qk_ref.float().abs().max())
    assert float((vt.float() - vt_ref.float()).abs().max()) <= tol * float(vt_ref.float().abs().max())
    print(f"qkv {dt} B={B} tokens={tokens} C={C}: bit-identical to the two launches: qk {bool(torch.equal(qk, qk_ref))} vt {bool(torch.equal(vt, vt_ref))}")


@pytest.mark.parametrize("B,tokens,C,presplit_a", [(8, 4096, 320, True), (8, 1024, 640, False), (8, 256, 1280, True), (4, 1024, 640, True), (8, 64, 1280, False)])
def test_fused_qkv_projection_float32_matrix_core_path(B, tokens, C, presplit_a):
    """The same launch on the float32 path (three float16 passes, pre-split stacked weight; optionally a pre-split activation operand):
    Q|K and V^T against float64 at float32 grade."""
    from gm_diffusion import hip_ops as o
    from gm_diffusion._native import lib

    prev = o.set_f32_mode("split")
    try:
        g = torch.Generator().manual_seed(B + tokens + C)
        n = torch.randn(B * tokens, C, generator=g).to(DEV)
        w = (torch.randn(3 * C, C, generator=g) * C ** -0.5).to(DEV)
        a = o.split_activation(n) if presplit_a else n
        r = o.gemm_qkv_vt(a, o.split_weights(w), 2 * C, tokens)
        if r is None:
            assert not lib().gmd_gemm_qkv_vt_ok(4, B * tokens, 3 * C, C, 2 * C, tokens, o.WORKSPACE_BYTES)
            return
        qk, vt = r
        ref = n.double() @ w.double().t()
        e1 = float((qk.double() - ref[:, :2 * C]).norm() / ref[:, :2 * C].norm())
        vd = ref[:, 2 * C:].view(B, tokens, C).transpose(1, 2)
        e2 = float((vt.double() - vd).norm() / vd.norm())
        assert vt.shape == (B, C, tokens) and e1 < 1.5e-6 and e2 < 1.5e-6, (e1, e2)
        # the plain and the pre-split activation operand give the same bits
        r2 = o.gemm_qkv_vt(n if presplit_a else o.split_activation(n), o.split_weights(w), 2 * C, tokens)
        assert torch.equal(r2[0], qk) and torch.equal(r2[1], vt)
    finally:
        o.set_f32_mode(prev)


# ------------------------------------------------------------------------------------------------------------------------------
# Round 5: in-kernel split-K reduction (csrc/gemm.hip: splitk_fixup) and the co-running plan family (gmd_gemm_plan_family)
# ------------------------------------------------------------------------------------------------------------------------------
@pytest.fixture
def fixup_knob():
    from gm_diffusion._native import lib

    prev = lib().gmd_splitk_fixup_max(-1)
    yield lib().gmd_splitk_fixup_max
    lib().gmd_splitk_fixup_max(prev)


def _tail_is_zero(ops):
    ws = ops._workspace(torch.device("cuda", torch.cuda.current_device()))
    return int(ws[-(ops.WS_TAIL_BYTES // 4):].view(torch.int32).abs().max()) == 0


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K,slices", [
    (2048, 1280, 1280, 2),   # SDR UNet 16x16 projections: 80 tiles of 256 x 128 -> 2 slices (the target is 128 workgroups: half the chip)
    (1024, 1280, 1280, 2),   # GM UNet 16x16 projections
    (4096, 640, 2560, 2),    # GM UNet 32x32 feed-forward output projection: 80 tiles
    (768, 1280, 5120, 4),    # 30 tiles -> 4 slices: the finisher adds three fragment sets in slice order
    (1024, 1280, 5120, 3),   # GM UNet 16x16 feed-forward output projection: 3 slices (the last one shorter)
    (2000, 1280, 1280, 2),   # ragged last row tile: the finisher leaves through the register epilogue
])
def test_splitk_fixup_is_bit_identical_to_the_slab_reduction(dt, M, N, K, slices, fixup_knob):
    """Split-K launches of the co-running plan family reduce inside the kernel (the last K slice of a tile waits for the others'
    accumulator fragments, adds them in slice order and runs the fused epilogue) instead of writing float32 slabs for a reduction
    launch.  Same partial sums, same order of additions: the outputs must be BIT-identical to the slab path, launch after launch
    (the arrival counters in the workspace tail clean themselves), and float32-grade against the float64 product."""
    from gm_diffusion import hip_ops as ops

    g = torch.Generator().manual_seed(M + K)
    a = torch.randn(M, K, generator=g).to(dt).to(DEV)
    w = (torch.randn(N, K, generator=g) * 0.03).to(dt).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    r = torch.randn(M, N, generator=g).to(dt).to(DEV)
    with ops.plan_family(1):
        plan = ops.gemm_plan_info(dt, M, N, K)
        assert plan[0] == 256 and plan[2] == 283 and 2 <= plan[3] <= min(slices, 4), plan  # (128-column tiles where one round suffices: fewer slices)
        fixup_knob(0)
        ref = ops.gemm_nt(a, w, bias=b, residual=r)
        fixup_knob(4)
        outs = [ops.gemm_nt(a, w, bias=b, residual=r) for _ in range(3)]
    torch.cuda.synchronize()
    for y in outs:
        assert torch.equal(y, ref)
    assert _tail_is_zero(ops)
    exact = a.double().cpu() @ w.double().cpu().T + b.double().cpu() + r.double().cpu()
    assert _rel(ref.cpu(), exact) < (6e-3 if dt == torch.bfloat16 else 8e-4)


@pytest.mark.parametrize("B,H,ci,co,slices", [(4, 32, 640, 640, 2), (4, 32, 1280, 640, 4), (8, 16, 1280, 1280, 4), (2, 64, 320, 320, 2)])
def test_splitk_fixup_conv3x3_bit_identical(B, H, ci, co, slices, fixup_knob):
    """The same for the convolutions (patch-resident kernel and the per-tap ping-pong kernel share the fix-up), with the two epilogues
    a ResnetBlock2D uses (conv1 = bias + time-embedding row bias, conv2 = bias + residual) and with all three (every epilogue adds
    them as (acc * alpha + bias) + (residual + row bias))."""
    from gm_diffusion import hip_ops as ops

    g = torch.Generator().manual_seed(B * H + ci)
    x = torch.randn(B, H * H, ci, generator=g).bfloat16().to(DEV)
    w = (torch.randn(co, 9 * ci, generator=g) * 0.02).bfloat16().to(DEV)
    b = torch.randn(co, generator=g).to(DEV)
    rb = torch.randn(B, co, generator=g).to(DEV)
    r = torch.randn(B, H * H, co, generator=g).bfloat16().to(DEV)
    with ops.plan_family(1):
        assert 2 <= ops.gemm_plan_info(torch.bfloat16, B * H * H, co, 9 * ci)[3] <= slices
        for kw, exact in ((dict(bias=b, rowbias=rb), True), (dict(bias=b, residual=r), True), (dict(bias=b, rowbias=rb, residual=r), True)):
            fixup_knob(0)
            ref = ops.conv3x3(x, w, B, H, H, **kw)[0]
            fixup_knob(4)
            outs = [ops.conv3x3(x, w, B, H, H, **kw)[0] for _ in range(2)]
            torch.cuda.synchronize()
            for y in outs:
                assert torch.equal(y, ref) if exact else _rel(y, ref) < 2e-3
            assert _rel(ref, _conv_ref(x, w, B, H, H, **kw)) < 6e-3
    assert _tail_is_zero(ops)


@pytest.mark.parametrize("kind,shape", [("gemm", (2048, 1280, 10240)), ("gemm", (1024, 1280, 5120)), ("conv", (8, 16, 1280, 1280)), ("conv", (8, 16, 2560, 1280))])
def test_splitk_fixup_loader_consumer_kernel_bit_identical(kind, shape, fixup_knob):
    """The launch-by-launch plan family splits K too (loader / consumer kernel, 128-row tiles, 4 consumer waves): its K slices take the
    same in-kernel reduction -- bit-identical to the slab path there as well (what the VAE and every single-stream caller run)."""
    from gm_diffusion import hip_ops as ops

    g = torch.Generator().manual_seed(sum(shape))
    if kind == "gemm":
        M, N, K = shape
        a = torch.randn(M, K, generator=g).bfloat16().to(DEV)
        w = (torch.randn(N, K, generator=g) * 0.03).bfloat16().to(DEV)
        b = torch.randn(N, generator=g).to(DEV)
        r = torch.randn(M, N, generator=g).bfloat16().to(DEV)
        plan = ops.gemm_plan_info(torch.bfloat16, M, N, K)
        run = lambda: ops.gemm_nt(a, w, bias=b, residual=r)
    else:
        B, H, ci, co = shape
        x = torch.randn(B, H * H, ci, generator=g).bfloat16().to(DEV)
        w = (torch.randn(co, 9 * ci, generator=g) * 0.02).bfloat16().to(DEV)
        b = torch.randn(co, generator=g).to(DEV)
        rb = torch.randn(B, co, generator=g).to(DEV)
        plan = ops.gemm_plan_info(torch.bfloat16, B * H * H, co, 9 * ci)
        run = lambda: ops.conv3x3(x, w, B, H, H, bias=b, rowbias=rb)[0]
    assert plan[2] == 244 and 2 <= plan[3] <= 4, plan  # the loader / consumer kernel with K slices
    fixup_knob(0)
    ref = run()
    fixup_knob(4)
    outs = [run() for _ in range(3)]
    torch.cuda.synchronize()
    for y in outs:
        assert torch.equal(y, ref)
    assert _tail_is_zero(ops)


def test_splitk_fixup_under_a_second_streams_load(fixup_knob):
    """The finisher spins on an arrival counter while the producers of its tile run on other CUs.  Replayed from a captured graph
    beside a second stream that keeps the chip busy with its own split launches (the shipped regime: two UNet forwards side by side),
    every replay must give the quiet run's bits, and the counters must end at zero."""
    from gm_diffusion import hip_ops as ops

    g = torch.Generator().manual_seed(77)
    shapes = [(2048, 1280, 1280), (1024, 1280, 5120), (8192, 640, 2560), (4096, 640, 640 * 4)]
    prob = []
    for M, N, K in shapes:
        prob.append((torch.randn(M, K, generator=g).bfloat16().to(DEV), (torch.randn(N, K, generator=g) * 0.03).bfloat16().to(DEV),
                     torch.randn(N, generator=g).to(DEV)))
    fixup_knob(4)

    def run_all():
        return [ops.gemm_nt(a, w, bias=b) for a, w, b in prob]

    with ops.plan_family(1):
        quiet = run_all()
        torch.cuda.synchronize()
        side = ops.side_stream(DEV)
        gr, ws = torch.cuda.CUDAGraph(), ops.new_workspace(torch.device(DEV))
        with ops.workspace_scope(ws), torch.cuda.graph(gr):
            outs = run_all()
        ws2 = ops.new_workspace(torch.device(DEV))
        for it in range(40):
            with torch.cuda.stream(side), ops.workspace_scope(ws2):
                for _ in range(2):
                    run_all()
            gr.replay()
            if it % 8 == 7:
                torch.cuda.synchronize()
                for y, q in zip(outs, quiet):
                    assert torch.equal(y, q), f"replay {it}"
    torch.cuda.synchronize()
    for t in (ws, ws2):
        assert int(t[-(ops.WS_TAIL_BYTES // 4):].view(torch.int32).abs().max()) == 0


def test_plan_families_agree_and_are_scoped_to_the_thread():
    """gmd_gemm_plan_family: family 1 (co-running: 256-row tiles + K slices) and family 0 (launch-by-launch) compute the same
    product up to float32 summation order; the choice is the calling thread's and the scope restores it."""
    import threading

    from gm_diffusion import hip_ops as ops
    from gm_diffusion._native import lib

    g = torch.Generator().manual_seed(3)
    M, N, K = 2048, 1280, 1280
    a = torch.randn(M, K, generator=g).bfloat16().to(DEV)
    w = (torch.randn(N, K, generator=g) * 0.03).bfloat16().to(DEV)
    assert lib().gmd_gemm_plan_family(-1) == 0
    p0 = ops.gemm_plan_info(torch.bfloat16, M, N, K)
    y0 = ops.gemm_nt(a, w)
    seen = {}
    with ops.plan_family(1):
        p1 = ops.gemm_plan_info(torch.bfloat16, M, N, K)
        y1 = ops.gemm_nt(a, w)
        t = threading.Thread(target=lambda: seen.setdefault("other", lib().gmd_gemm_plan_family(-1)))
        t.start(); t.join()
    assert lib().gmd_gemm_plan_family(-1) == 0 and seen["other"] == 0
    assert p0 != p1 and p1[0] == 256 and p1[2] == 283 and p1[3] > 1
    assert _rel(y1, y0) < 2e-3 and _rel(y0.cpu(), a.double().cpu() @ w.double().cpu().T) < 6e-3

"""float32 contractions on the matrix cores (GMD_F32S / GMD_F32SW: three float16 products per float32 product,
csrc/gemm_split.hip, the float32 attention of csrc/attention_split.hip) against float64 references, against the exact
float32 FMA kernels, and pre-split weights against the in-kernel split (bit for bit)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def ops():
    from gm_diffusion import hip_ops

    return hip_ops


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(autouse=True)
def split_mode():
    o = ops()
    prev = o.set_f32_mode("split")
    yield
    o.set_f32_mode(prev)


# float32-grade: an exact fp32 FMA chain over K = 320..5120 terms sits at 1e-7..4e-7 relative to float64; the split path
# adds ~2^-22 per term (random signs): the bound below is 3x the exact kernel's own error at the deepest K.
TOL = 1.5e-6


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (128, 128, 64), (300, 200, 320), (8, 1280, 1280), (616, 320, 768), (4096, 320, 320),
                                   (2048, 2560, 320), (33, 4, 128), (70, 1000, 96), (1024, 1280, 5120), (256, 320, 5120), (2048, 640, 10240)])
def test_split_gemm_vs_float64(M, N, K):
    o = ops()
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    ref = a.double() @ w.double().t()
    ad, wd = a.to(DEV), w.to(DEV)
    got = o.gemm_nt(ad, wd)  # both operands split in the kernel, W unscaled (magnitude 1/sqrt(K): its lo half is partly subnormal)
    assert got.dtype == torch.float32 and rel_err(got, ref) < 3e-6, "plain"
    o.set_f32_mode("exact")
    e_exact = rel_err(o.gemm_nt(ad, wd), ref)  # the fp32 FMA chain's own distance from float64 (grows with K)
    o.set_f32_mode("split")
    ws = o.split_weights(wd)  # the product path: weight scaled by a power of two, split once
    got_w = o.gemm_nt(ad, ws)
    assert rel_err(got_w, ref) < max(4e-7, 1.5 * e_exact), "pre-split: no worse than 1.5x the exact float32 kernel"
    assert torch.equal(got_w, o.gemm_nt(ad, o.scale_weight(wd))), "pre-split weights must give the in-kernel split's bits"
    got = o.gemm_nt(ad, ws, bias=bias.to(DEV), residual=res.to(DEV), alpha=0.5)
    assert rel_err(got, 0.5 * ref + bias.double() + res.double()) < TOL, "bias+residual"
    got = o.gemm_nt(ad, ws, bias=bias.to(DEV), act=o.ACT_SILU)
    assert rel_err(got, F.silu(ref + bias.double())) < TOL, "silu"
    rpg = 7
    rb = torch.randn((M + rpg - 1) // rpg, N, generator=g)
    got = o.gemm_nt(ad, wd, rowbias=rb.to(DEV), rows_per_group=rpg)
    assert rel_err(got, ref + rb.double().repeat_interleave(rpg, 0)[:M]) < TOL, "rowbias"


def test_split_accuracy_is_float32_grade_not_float16_grade():
    """The same product with operands rounded to float16 once (the float16 path's arithmetic) is ~500x further from float64."""
    o = ops()
    g = torch.Generator().manual_seed(5)
    a, w = torch.randn(512, 1280, generator=g), torch.randn(640, 1280, generator=g) / 36.0
    ref = a.double() @ w.double().t()
    got = o.gemm_nt(a.to(DEV), w.to(DEV))
    half = a.half().double() @ w.half().double().t()
    e_split, e_half = rel_err(got, ref), rel_err(half, ref)
    assert e_split < 1e-6 and e_half > 100 * e_split, (e_split, e_half)
    e_scaled = rel_err(o.gemm_nt(a.to(DEV), o.split_weights(w.to(DEV))), ref)
    assert e_scaled < 4e-7, e_scaled


def test_split_small_weights_keep_their_precision_through_the_power_of_two_scale():
    """The lo half of a split operand is a float16 of ~2^-11 of the value: for weights of magnitude 1e-3 it would fall into
    float16's subnormal range and lose its bits.  split_weights / scale_weight store the weight scaled by a power of two
    (exact) and fold the inverse into alpha: the result must not depend on the weight's scale."""
    o = ops()
    g = torch.Generator().manual_seed(6)
    a = torch.randn(256, 640, generator=g) * torch.logspace(-2, 1, 640)[None, :]
    w = torch.randn(320, 640, generator=g)
    ref = a.double() @ w.double().t()
    ad = a.to(DEV)
    for scale in (1.0, 0.02, 2.0 ** -12, 300.0):
        wd = (w * scale).to(DEV)
        got = o.gemm_nt(ad, o.split_weights(wd))
        assert rel_err(got, ref * scale) < 5e-7, scale
        vt = o.gemm_nt(o.scale_weight(wd), ad)  # weight as the A operand (V^T = W_v x^T)
        assert rel_err(vt, (ref * scale).t()) < 5e-7, scale
    tiny = o.gemm_nt(ad, (w * 2.0 ** -12).to(DEV))  # unscaled tiny weights: still far better than float16, but not float32-grade
    assert 5e-7 < rel_err(tiny, ref * 2.0 ** -12) < 1e-4


def test_split_batched_and_swapped():
    o = ops()
    g = torch.Generator().manual_seed(9)
    Bn, N, C = 3, 72, 128
    x = torch.randn(Bn, N, C, generator=g)
    wv = torch.randn(C, C, generator=g) / math.sqrt(C)
    ld = 80
    vt = o.gemm_nt(wv.to(DEV), x.to(DEV), ldc=ld)  # V^T[b] = W_v @ x[b]^T: both operands plain float32
    ref = torch.einsum("ck,bnk->bcn", wv.double(), x.double())
    assert vt.shape == (Bn, C, ld) and rel_err(vt[:, :, :N], ref) < TOL
    y = o.gemm_nt(x.to(DEV), torch.stack([wv, wv * 2, wv * 3]).to(DEV))
    ref2 = torch.stack([x[i].double() @ (wv.double() * (i + 1)).t() for i in range(3)])
    assert rel_err(y, ref2) < TOL


@pytest.mark.parametrize("B,H,W,Cin,Cout,mode", [
    (2, 8, 8, 64, 64, "s1"), (1, 16, 12, 128, 320, "s1"), (2, 8, 8, 32, 128, "s2"), (1, 9, 7, 64, 64, "s2"),
    (2, 4, 6, 96, 64, "up"), (1, 8, 8, 64, 64, "pad1"), (1, 7, 9, 32, 64, "pad1"), (1, 64, 64, 320, 320, "s1"),
    (2, 8, 8, 64, 4, "s1"), (8, 8, 8, 1280, 1280, "s1"), (2, 32, 32, 1280, 640, "s1"), (2, 64, 64, 640, 320, "s1"),
])
def test_split_conv3x3_vs_float64(B, H, W, Cin, Cout, mode):
    o = ops()
    g = torch.Generator().manual_seed(H * W + Cin + Cout)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    bias = torch.randn(Cout, generator=g)
    xd, wd = x.double(), w.double()
    if mode == "s1":
        ref, kw = F.conv2d(xd, wd, bias.double(), padding=1), {}
    elif mode == "s2":
        ref, kw = F.conv2d(xd, wd, bias.double(), stride=2, padding=1), dict(stride=2)
    elif mode == "up":
        ref, kw = F.conv2d(F.interpolate(xd, scale_factor=2.0, mode="nearest"), wd, bias.double(), padding=1), dict(upsample=True)
    else:
        ref, kw = F.conv2d(F.pad(xd, (0, 1, 0, 1)), wd, bias.double(), stride=2, padding=0), dict(stride=2, pad_mode=1)
    Ho, Wo = ref.shape[-2:]
    xl = x.permute(0, 2, 3, 1).reshape(B, H * W, Cin).contiguous().to(DEV)
    wl = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous().to(DEV)
    tb = torch.randn(B, Cout, generator=g)
    res = torch.randn(B, Ho * Wo, Cout, generator=g)
    full = (ref + tb.double()[:, :, None, None]).permute(0, 2, 3, 1).reshape(B, Ho * Wo, Cout) + res.double()
    y, ho, wo = o.conv3x3(xl, wl, B, H, W, bias=bias.to(DEV), rowbias=tb.to(DEV), residual=res.to(DEV), **kw)
    assert (ho, wo) == (Ho, Wo) and rel_err(y, full) < TOL
    y2, _, _ = o.conv3x3(xl, o.split_weights(wl), B, H, W, bias=bias.to(DEV), rowbias=tb.to(DEV), residual=res.to(DEV), **kw)
    assert rel_err(y2, full) < 5e-7
    y3, _, _ = o.conv3x3(xl, o.scale_weight(wl), B, H, W, bias=bias.to(DEV), rowbias=tb.to(DEV), residual=res.to(DEV), **kw)
    assert torch.equal(y3, y2), "pre-split conv weights must give the in-kernel split's bits"


def test_split_epilogue_fuzz():
    """Seeded sweep over shapes that straddle every epilogue path of the split kernels (row-contiguous LDS epilogue on full
    tiles, register epilogue on ragged edges, split-K slabs, 64x64 tiles) with all combinations of bias / row bias /
    residual / activation / alpha."""
    o = ops()
    rng = np.random.default_rng(31)
    g = torch.Generator().manual_seed(31)
    acts = [(o.ACT_NONE, lambda z: z), (o.ACT_SILU, F.silu), (o.ACT_QUICK_GELU, lambda z: z * torch.sigmoid(1.702 * z))]
    cases = 0
    for _ in range(36):
        M = int(rng.choice([128, 256, 384, 640, 1000, 1024, 2048, 4096, 16384]))
        N = int(rng.choice([64, 128, 160, 320, 328, 640, 1280]))
        K = int(rng.choice([32, 96, 320, 640, 1536, 2560]))
        if M * N * K > 16384 * 640 * 640:
            continue
        use_bias, use_rb, use_res, presplit = (bool(v) for v in rng.integers(0, 2, 4))
        act, fn = acts[int(rng.integers(0, 3))]
        alpha = float(rng.choice([1.0, 0.5]))
        a = torch.randn(M, K, generator=g)
        w = torch.randn(N, K, generator=g) / math.sqrt(K)
        ref = alpha * (a.double() @ w.double().t())
        kw = {}
        if use_bias:
            b = torch.randn(N, generator=g)
            kw["bias"] = b.to(DEV)
            ref = ref + b.double()
        if use_rb:
            rpg = int(rng.choice([16, 48, 64, 100, 256, M]))
            rb = torch.randn((M + rpg - 1) // rpg, N, generator=g)
            kw.update(rowbias=rb.to(DEV), rows_per_group=rpg)
            ref = ref + rb.double().repeat_interleave(rpg, 0)[:M]
        if use_res:
            r = torch.randn(M, N, generator=g)
            kw["residual"] = r.to(DEV)
            ref = ref + r.double()
        wd = w.to(DEV)
        got = o.gemm_nt(a.to(DEV), o.split_weights(wd) if presplit else wd, alpha=alpha, act=act, **kw)
        assert got.shape == (M, N)
        assert rel_err(got, fn(ref)) < TOL, (M, N, K, use_bias, use_rb, use_res, act, alpha, presplit)
        cases += 1
    assert cases >= 28


def test_split_rejects_what_it_cannot_do():
    o = ops()
    a = torch.randn(64, 48, device=DEV)
    w = torch.randn(64, 48, device=DEV)
    y = o.gemm_nt(a, w)  # K % 32 != 0: the exact kernel takes it
    assert rel_err(y, a.double() @ w.double().t()) < 2e-6
    with pytest.raises(o.HipExtensionError):
        o.split_weights(w)
    ws = o.split_weights(torch.randn(64, 64, device=DEV))
    with pytest.raises(o.HipExtensionError):
        o.gemm_nt(ws, torch.randn(64, 64, device=DEV))  # a pre-split weight is a W operand only
    with pytest.raises(o.HipExtensionError):
        o.gemm_nt(torch.randn(64, 64, device=DEV).half(), ws)


# ---------------------------------------------------------------------------------------------
# float32 attention (attention_split.hip)
# ---------------------------------------------------------------------------------------------
def _attn_ref(q, k, v, heads, scale):
    B, Nq, C = q.shape
    d = C // heads
    qh = q.double().view(B, Nq, heads, d).transpose(1, 2)
    kh = k.double().view(B, -1, heads, d).transpose(1, 2)
    vh = v.double().view(B, -1, heads, d).transpose(1, 2)
    p = torch.softmax(qh @ kh.transpose(-1, -2) * scale, -1)
    return (p @ vh).transpose(1, 2).reshape(B, Nq, C)


def _vt(v, ld):
    B, Nk, C = v.shape
    out = torch.zeros(B, C, ld)
    out[:, :, :Nk] = v.transpose(1, 2)
    return out


@pytest.mark.parametrize("D", [40, 64, 80, 160])
@pytest.mark.parametrize("Nq,Nk", [(256, 256), (64, 77), (16, 16), (200, 130), (1024, 1024), (4096, 77)])
def test_split_attention_vs_float64(D, Nq, Nk):
    o = ops()
    H, B = 2, 2
    g = torch.Generator().manual_seed(D + Nq + Nk)
    q = torch.randn(B, Nq, H * D, generator=g)
    k = torch.randn(B, Nk, H * D, generator=g)
    v = torch.randn(B, Nk, H * D, generator=g)
    scale = D ** -0.5
    ref = _attn_ref(q, k, v, H, scale)
    ld = (Nk + 3) // 4 * 4
    got = o.attention(q.to(DEV), k.to(DEV), _vt(v, ld).to(DEV), H, Nk, scale)
    assert got.dtype == torch.float32 and rel_err(got, ref) < 1e-6, rel_err(got, ref)


def test_split_attention_fused_qk_buffer_poisoned_padding_and_sharp_rows():
    """Q and K as column ranges of one fused projection buffer; NaN-poisoned V^T row padding beyond Nk (ldvt > Nk) must never
    reach the output; logits large enough that single keys dominate (rescale branch of the online softmax), all negative
    logits, and one spiked key per row."""
    o = ops()
    H, D, B, N = 8, 40, 2, 333
    C = H * D
    g = torch.Generator().manual_seed(77)
    qk = torch.randn(B, N, 2 * C, generator=g)
    v = torch.randn(B, N, C, generator=g)
    ld = 336 + 8
    vt = _vt(v, ld)
    vt[:, :, N:] = float("nan")
    for mul in (1.0, 6.0, 25.0):
        x = qk.clone()
        x[:, :, :C] *= mul
        ref = _attn_ref(x[:, :, :C], x[:, :, C:], v, H, D ** -0.5)
        got = o.attention(x.to(DEV), x.to(DEV), vt.to(DEV), H, N, D ** -0.5, k_col=C)
        assert torch.isfinite(got).all() and rel_err(got, ref) < 1e-6, mul
    x = qk.clone()
    x[:, :, C:] = -x[:, :, :C].abs() * 3  # all logits strongly negative
    x[:, :, :C] = x[:, :, :C].abs()
    x[:, 200, C:] = 4.0                   # ... except one key that every query prefers, late in the sequence
    ref = _attn_ref(x[:, :, :C], x[:, :, C:], v, H, D ** -0.5)
    got = o.attention(x.to(DEV), x.to(DEV), vt.to(DEV), H, N, D ** -0.5, k_col=C)
    assert rel_err(got, ref) < 1e-6


def test_split_attention_full_size_properties():
    """4096 tokens (512^2), 8 heads of 40: rows of ones for V == 1, linear in V, invariant to the key order."""
    o = ops()
    H, D, B, N = 8, 40, 2, 4096
    C = H * D
    g = torch.Generator().manual_seed(3)
    qk = (torch.randn(B, N, 2 * C, generator=g)).to(DEV)
    v = torch.randn(B, N, C, generator=g)
    ones = o.attention(qk, qk, torch.ones(B, C, N, device=DEV), H, N, D ** -0.5, k_col=C)
    assert float((ones - 1).abs().max()) < 4e-6  # 4096-term float32 row sums
    vt = v.transpose(1, 2).contiguous().to(DEV)
    o1 = o.attention(qk, qk, vt, H, N, D ** -0.5, k_col=C)
    o2 = o.attention(qk, qk, (2 * vt + 1).contiguous(), H, N, D ** -0.5, k_col=C)
    assert rel_err(o2, 2 * o1 + 1) < 1e-6
    perm = torch.randperm(N, generator=g).to(DEV)
    kperm = qk.clone()
    kperm[:, :, C:] = qk[:, perm, C:]
    o3 = o.attention(qk, kperm, vt[:, :, perm].contiguous(), H, N, D ** -0.5, k_col=C)
    assert rel_err(o3, o1) < 3e-6  # two summation orders of 4096 mostly cancelling terms (|o| ~ 0.02 |v|)
    # against the exact composition on one (batch, head)
    ref = _attn_ref(qk[:1, :, :D].cpu(), qk[:1, :, C:C + D].cpu(), v[:1, :, :D], 1, D ** -0.5)
    assert rel_err(o1[:1, :, :D], ref) < 1e-6


def test_split_attention_rejects_other_modes():
    o = ops()
    q = torch.randn(1, 64, 32, device=DEV)
    with pytest.raises(o.HipExtensionError):
        o.attention(q, q, torch.randn(1, 32, 64, device=DEV), 1, 64, 0.125)  # head dim 32 is not instantiated for float32
    o.set_f32_mode("exact")
    q = torch.randn(1, 64, 80, device=DEV)
    with pytest.raises(o.HipExtensionError):
        o.attention(q, q, torch.randn(1, 80, 64, device=DEV), 2, 64, 0.15)


@pytest.mark.parametrize("M,C", [(300, 320), (4096, 640), (64, 1280), (33, 64), (16384, 320)])
def test_split_fused_geglu_epilogue(M, C):
    """ff.net.0.proj with the value / gate rows interleaved in 16-row groups: value * gelu_erf(gate) out of the GEMM epilogue
    (full tiles through the LDS row epilogue, ragged tiles from registers), against float64."""
    o = ops()
    g = torch.Generator().manual_seed(M + C)
    a = torch.randn(M, C, generator=g)
    wf = torch.randn(8 * C, C, generator=g) / math.sqrt(C)
    bf = torch.randn(8 * C, generator=g)
    half = 4 * C
    wi = torch.stack([wf[:half].reshape(half // 16, 16, -1), wf[half:].reshape(half // 16, 16, -1)], 1).reshape(2 * half, -1)
    bi = torch.stack([bf[:half].reshape(half // 16, 16), bf[half:].reshape(half // 16, 16)], 1).reshape(2 * half)
    z = a.double() @ wf.double().t() + bf.double()
    ref = z[:, :half] * F.gelu(z[:, half:])
    got = o.gemm_nt(a.to(DEV), o.split_weights(wi.to(DEV)), bias=bi.to(DEV), act=o.ACT_GEGLU)
    assert got.shape == (M, half) and rel_err(got, ref) < 1e-6
    got2 = o.gemm_nt(a.to(DEV), wi.to(DEV), bias=bi.to(DEV), act=o.ACT_GEGLU)  # plain float32 W, split in the kernel
    assert rel_err(got2, ref) < 3e-6


def test_new_lds_dma_kernels_are_repeatable_beside_another_stream():
    """The round-3 kernels stage operands with LDS-DMA behind waits and barriers (gemm_split_kernel: GEMM, conv3x3, split-K;
    ff_fused_kernel: ring + the H tile handed from the first product to the second through LDS; attn_split_kernel: register-staged
    hi / lo planes): an ordering bug shows up as an occasional wrong tile, not as a steady error.  Same inputs, many launches, a
    second stream keeping the memory system busy: every output must be bit-identical to the first."""
    o = ops()
    g = torch.Generator().manual_seed(23)
    a = torch.randn(4096, 640, generator=g).to(DEV)
    w = o.split_weights((torch.randn(640, 640, generator=g) * 0.04).to(DEV))
    ak = torch.randn(256, 5120, generator=g).to(DEV)           # split-K plan (few tiles, deep K)
    wk = o.split_weights((torch.randn(320, 5120, generator=g) * 0.02).to(DEV))
    xc = torch.randn(2, 32 * 32, 320, generator=g).to(DEV)
    wc = o.split_weights((torch.randn(320, 9 * 320, generator=g) * 0.02).to(DEV))
    qk = torch.randn(2, 1024, 640, generator=g).to(DEV)
    vt = torch.randn(2, 320, 1024, generator=g).to(DEV)
    xh = torch.randn(4096, 320, generator=g).bfloat16().to(DEV)
    rh = torch.randn(4096, 320, generator=g).bfloat16().to(DEV)
    w1 = (torch.randn(2560, 320, generator=g) * 0.05).bfloat16().to(DEV)
    b1 = torch.randn(2560, generator=g).to(DEV)
    w2 = (torch.randn(320, 1280, generator=g) * 0.03).bfloat16().to(DEV)
    b2 = torch.randn(320, generator=g).to(DEV)
    xa, wa = torch.randn(8192, 320, generator=g).bfloat16().to(DEV), torch.randn(320, 320, generator=g).bfloat16().to(DEV)
    cases = {
        "gemm": lambda: o.gemm_nt(a, w),
        "gemm split-K": lambda: o.gemm_nt(ak, wk),
        "conv": lambda: o.conv3x3(xc, wc, 2, 32, 32)[0],
        "attention": lambda: o.attention(qk, qk, vt, 8, 1024, 40 ** -0.5, k_col=320),
        "fused feed-forward": lambda: o.ff_geglu_fused(xh, w1, b1, w2, b2, rh),
    }
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    for name, fn in cases.items():
        first = fn().clone()
        differ = 0
        for it in range(30):
            if it % 3 == 0:
                with torch.cuda.stream(side):
                    for _ in range(3):
                        o.gemm_nt(xa, wa)
            differ += int(not torch.equal(fn(), first))
        torch.cuda.synchronize()
        assert differ == 0, name


def test_split_range_limit_is_guarded_and_the_exact_mode_has_none():
    """The float32 "split" path takes every operand as f16 hi + f16 lo: an activation beyond 65504 splits into inf / -inf and the
    product is NaN.  The kernels do not test for it; the consumers of the models' outputs do (hip_ops.check_split_range, called by
    both pipelines and decode_to_hdr): it must raise a HipExtensionError that names the remedy, and the exact kernels must give
    the finite result on the same operands."""
    from gm_diffusion._native import HipExtensionError

    o = ops()
    g = torch.Generator().manual_seed(2)
    x = torch.randn(256, 64, generator=g)
    x[3, 5] = 1.0e5  # beyond float16's largest finite value
    w = torch.randn(128, 64, generator=g) * 0.05
    prev = o.set_f32_mode("split")
    try:
        y = o.gemm_nt(x.to(DEV), o.split_weights(w.to(DEV)))
        assert not torch.isfinite(y[3]).all()          # the documented limit of the split path ...
        assert torch.isfinite(y[:3]).all() and torch.isfinite(y[4:]).all()  # ... confined to the row that holds the value
        with pytest.raises(HipExtensionError, match="GMD_F32_MODE=exact"):
            o.check_split_range("gemm_nt output", y)
        o.check_split_range("finite rows", y[4:].contiguous())  # no error
        o.set_f32_mode("exact")
        ye = o.gemm_nt(x.to(DEV), w.to(DEV))
        assert torch.isfinite(ye).all() and rel_err(ye, x.double() @ w.double().t()) < 1e-5
        o.check_split_range("exact output", ye)
    finally:
        o.set_f32_mode(prev)


def test_pipeline_raises_on_split_range_overflow_and_runs_in_exact_mode():
    """Same limit end to end: latents of magnitude 1e6 overflow float16 in conv_in's operand split.  The dual pipeline must raise
    (not return NaN latents) under "split"; placed on the device under "exact" the same call is finite."""
    from gm_diffusion._native import HipExtensionError
    from gm_diffusion.components import AutoencoderKL, PNDMScheduler, UNet2DConditionModel
    from gm_diffusion.pipelines import StableDiffusionDualUNetPipeline
    from oracle import fixtures

    o = ops()

    def pipe():
        def hip(cls, om):
            m = cls(**vars(om.config))
            m.load_state_dict(om.state_dict())
            return m.to(DEV, torch.float32)

        p = StableDiffusionDualUNetPipeline(
            vae=hip(AutoencoderKL, fixtures.build_vae("tiny")), text_encoder=None, tokenizer=None,
            unet=hip(UNet2DConditionModel, fixtures.build_unet("tiny", 4)), gm_unet=hip(UNet2DConditionModel, fixtures.build_unet("tiny", 8)),
            scheduler=PNDMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", skip_prk_steps=True, steps_offset=1,
                                    set_alpha_to_one=False), safety_checker=None, feature_extractor=None, requires_safety_checker=False)
        p.set_progress_bar_config(disable=True)
        return p

    g = torch.Generator().manual_seed(4)
    pe, ne = torch.randn(1, 77, 64, generator=g).to(DEV), torch.randn(1, 77, 64, generator=g).to(DEV)
    lat = (torch.randn(1, 4, 16, 16, generator=g) * 1.0e6).to(DEV)
    kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, latents=lat, height=128, width=128, num_inference_steps=2, output_type="latent")
    prev = o.set_f32_mode("split")
    try:
        with pytest.raises(HipExtensionError, match="float16's range"):
            pipe()(**kw)
        o.set_f32_mode("exact")
        sdr, gm = pipe()(**kw)
        assert torch.isfinite(sdr).all() and torch.isfinite(gm).all()
    finally:
        o.set_f32_mode(prev)


def test_module_keeps_the_f32_mode_it_was_prepared_for():
    """hip_ops.F32_MODE is a process-wide switch, a module's weights are laid out (pre-split, power-of-two scaled, channel padded)
    for ONE mode when they reach the device.  A model prepared under one mode must give bit-identical results after the switch
    was flipped (bench.py keeps split and exact pipelines alive side by side) -- not fail, and not run one mode's weights through
    the other mode's kernels."""
    from gm_diffusion.components import AutoencoderKL, UNet2DConditionModel
    from oracle import fixtures

    o = ops()
    ou, ov = fixtures.build_unet("tiny", 4), fixtures.build_vae("tiny")
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 4, 16, 16, generator=g).to(DEV)
    ctx = torch.randn(2, 77, ou.config.cross_attention_dim, generator=g).to(DEV)
    z = torch.randn(2, 4, 8, 8, generator=g).to(DEV)
    for own, other in (("split", "exact"), ("exact", "split")):
        prev = o.set_f32_mode(own)
        try:
            hu = UNet2DConditionModel(**vars(ou.config)); hu.load_state_dict(ou.state_dict()); hu = hu.to(DEV, torch.float32)
            hv = AutoencoderKL(**vars(ov.config)); hv.load_state_dict(ov.state_dict()); hv = hv.to(DEV, torch.float32)
            a = hu(x, 301, encoder_hidden_states=ctx, return_dict=False)[0]
            da = hv.decode(z, return_dict=False)[0]
            assert hu._f32_mode == own and hv._f32_mode == own
            o.set_f32_mode(other)
            b = hu(x, 301, encoder_hidden_states=ctx, return_dict=False)[0]
            db = hv.decode(z, return_dict=False)[0]
            assert o.F32_MODE == other  # the scope restores the caller's mode
            assert torch.equal(a, b) and torch.equal(da, db), own
        finally:
            o.set_f32_mode(prev)


@pytest.mark.parametrize("B,H,ci,co", [(8, 32, 64, 640), (4, 64, 32, 320), (2, 64, 64, 256)])
def test_split_kernels_emit_producer_statistics_for_groupnorm(B, H, ci, co):
    """Round 4: the float32 matrix-core kernels emit the GroupNorm column statistics out of their row epilogue like the 16-bit
    kernels (float32 GroupNorm then reads its tensor once): sums of the STORED values per 64 rows x 10 channels against float64,
    and GroupNorm from them against the statistics-launch path on the same tensor."""
    o = ops()
    g = torch.Generator().manual_seed(B + H + co)
    x = torch.randn(B, H * H, ci, generator=g).to(DEV)
    w = o.split_weights((torch.randn(co, 9 * ci, generator=g) * 0.05).to(DEV))
    b = torch.randn(co, generator=g).to(DEV)
    tb = torch.randn(B, co, generator=g).to(DEV)
    prev = o.set_f32_mode("split")
    try:
        y, _, _ = o.conv3x3(x, w, B, H, H, bias=b, rowbias=tb, colstats=True)
        if co % o.COLSTATS_BUCKET:  # 256 channels: no whole number of 10-channel buckets -> no statistics, the plain paths
            assert getattr(y, "_colstats", None) is None
            ga, be = torch.randn(co, generator=g).to(DEV), torch.randn(co, generator=g).to(DEV)
            ref = F.silu(F.group_norm(y.double().view(B, H * H, co).permute(0, 2, 1), 32, ga.double(), be.double(), 1e-5)).permute(0, 2, 1)
            assert rel_err(o.groupnorm(y, B, 32, ga, be, 1e-5, silu=True), ref) < 2e-6
            return
        assert getattr(y, "_colstats", None) is not None, "the launch must take the statistics path"
        st, _ = y._colstats
        if True:
            yd = y.double().view(B * H * H // 64, 64, co // o.COLSTATS_BUCKET, o.COLSTATS_BUCKET)
            s1, s2 = yd.sum((1, 3)), (yd * yd).sum((1, 3))
            assert float((st[..., 0].double() - s1).abs().max()) < 1e-3 * float(s1.abs().max() + 1)
            assert float(((st[..., 1].double() - s2).abs() / s2).max()) < 1e-5
        ga, be = torch.randn(co, generator=g).to(DEV), torch.randn(co, generator=g).to(DEV)
        before = o.colstats_uses
        a = o.groupnorm(y, B, 32, ga, be, 1e-5, silu=True)
        assert o.colstats_uses == before + 1
        bref = o.groupnorm(y.clone(), B, 32, ga, be, 1e-5, silu=True)  # no statistics attached: statistics launch
        ref = F.silu(F.group_norm(y.double().view(B, H * H, co).permute(0, 2, 1), 32, ga.double(), be.double(), 1e-5)).permute(0, 2, 1)
        assert rel_err(a, ref) < 2e-6 and rel_err(bref, ref) < 2e-6
    finally:
        o.set_f32_mode(prev)


@pytest.mark.parametrize("case", ["conv 8x32x32 64->640", "conv 3x24x20 64->320 s2", "gemm 8192x640x640 res", "gemm 1000x328x320 ragged", "geglu 2048x2560x320",
                                  "gemm 1024x1280x2560 k-slices"])
def test_split_loader_converter_kernel_is_bit_identical_to_the_in_register_split(case):
    """gemm_split_lc_kernel (round 4: the loader waves split a landed activation tile once, in place in LDS; the consumers read ready
    f16 fragments) forms the same three products per accumulator in the same order as gemm_split_kernel (every wave splits its
    fragments in registers): results must be equal bit for bit -- full and ragged tiles, convolution addressing, GEGLU, K slices."""
    import os

    from gm_diffusion._native import lib
    o = ops()
    g = torch.Generator().manual_seed(len(case))
    prev_env = os.environ.get("GMD_TUNING")
    os.environ["GMD_TUNING"] = "1"
    prev = o.set_f32_mode("split")
    try:
        if case.startswith("conv"):
            B, H, W, ci, co, kw = (8, 32, 32, 64, 640, {}) if "8x32" in case else (3, 24, 20, 64, 320, dict(stride=2))
            x = torch.randn(B, H * W, ci, generator=g).to(DEV)
            w = o.split_weights((torch.randn(co, 9 * ci, generator=g) * 0.05).to(DEV))
            b, tb = torch.randn(co, generator=g).to(DEV), torch.randn(B, co, generator=g).to(DEV)
            fn = lambda: o.conv3x3(x, w, B, H, W, bias=b, rowbias=tb, **kw)[0]
        else:
            M, N, K = {"gemm 8192x640x640 res": (8192, 640, 640), "gemm 1000x328x320 ragged": (1000, 328, 320), "geglu 2048x2560x320": (2048, 2560, 320),
                       "gemm 1024x1280x2560 k-slices": (1024, 1280, 2560)}[case]
            a = torch.randn(M, K, generator=g).to(DEV)
            w = o.split_weights((torch.randn(N, K, generator=g) * 0.05).to(DEV))
            b = torch.randn(N, generator=g).to(DEV)
            kw = dict(bias=b)
            if "res" in case:
                kw["residual"] = torch.randn(M, N, generator=g).to(DEV)
            if "geglu" in case:
                kw["act"] = o.ACT_GEGLU
            fn = lambda: o.gemm_nt(a, w, **kw)
        assert lib().gmd_gemm_plan_override(0, 0, 9, 0) == 0      # in-register split (round-3 kernel)
        y0 = fn()
        assert lib().gmd_gemm_plan_override(0, 0, 244, 0) == 0    # loader / converter kernel wherever it is instantiated
        y1 = fn()
        assert torch.isfinite(y0).all() and torch.equal(y0, y1)
    finally:
        lib().gmd_gemm_plan_override(0, 0, 0, 0)
        o.set_f32_mode(prev)
        if prev_env is None:
            os.environ.pop("GMD_TUNING", None)
        else:
            os.environ["GMD_TUNING"] = prev_env


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_producer_statistics_are_exact_under_a_second_streams_load(dtype):
    """Round-4 fault: with another stream's UNet graph running beside them, 1 launch in 200-1500 of the float32
    kernels returned producer statistics with three adjacent buckets wrong (lanes 48..63 of one accumulator of the column pass; the
    output tensor itself was right), which made graphs + two streams differ from eager by ~1e-4.  tools/stress_colstats.py replays
    GEMM / conv launches with statistics from a graph beside a whole UNet forward on the side stream and compares every launch's
    output and statistics bit for bit with a quiet run; 1,440 launches here (the old build failed 22 of 4,320 on the same box)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DTYPE=dtype, BG="unet", NREP="40", INNER="12")
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_colstats.py")], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-1500:])
    assert "launches with differing statistics" in p.stdout


# ---- round 4: float32 activations stored pre-split between a producer and the contraction that reads them (GMD_F32SA) ----
@pytest.mark.parametrize("case", ["gemm 8192x640x640 res", "gemm 32768x320x1280", "gemm 1000x328x320 ragged", "gemm 2048x1280x1280 (64x64 tiles)",
                                  "gemm 1024x1280x2560 k-slices", "geglu 8192x5120x640", "conv 8x32x32 64->640 rowbias", "conv 3x24x20 64->320 s2",
                                  "conv 4x16x16 640->1280 k-slices", "batched W operand"])
def test_presplit_activation_operand_is_bit_identical_to_the_in_kernel_split(case):
    """GMD_F32SA: the A operand arrives as [hi 64 B | lo 64 B] per 32 elements (what the GroupNorm / LayerNorm / GEGLU producers store);
    the kernel reads ready float16 fragments.  Same split values, same products, same order -> the same bits as the plain operand."""
    o = ops()
    g = torch.Generator().manual_seed(len(case))
    if case.startswith("conv"):
        B, H, W_, ci, co, kw = {"conv 8x32x32 64->640 rowbias": (8, 32, 32, 64, 640, {}), "conv 3x24x20 64->320 s2": (3, 24, 20, 64, 320, dict(stride=2)),
                                "conv 4x16x16 640->1280 k-slices": (4, 16, 16, 640, 1280, {})}[case]
        x = torch.randn(B, H * W_, ci, generator=g).to(DEV)
        w = o.split_weights((torch.randn(co, 9 * ci, generator=g) * 0.03).to(DEV))
        b = torch.randn(co, generator=g).to(DEV)
        if "rowbias" in case:
            kw = dict(kw, rowbias=torch.randn(B, co, generator=g).to(DEV))
        ref = o.conv3x3(x, w, B, H, W_, bias=b, **kw)[0]
        got = o.conv3x3(o.split_activation(x), w, B, H, W_, bias=b, **kw)[0]
        got2 = o.conv3x3(o.split_activation(x).view(B, H * W_, ci), w, B, H, W_, bias=b, x_split=True, **kw)[0]  # a view: explicit flag
        assert torch.equal(ref, got) and torch.equal(ref, got2)
        return
    if case == "batched W operand":  # V^T = Wv n^T with the activation as the (batched) W operand: it takes the pre-split WEIGHT path
        a = (torch.randn(320, 320, generator=g) * 0.05).to(DEV)
        n = torch.randn(4, 1024, 320, generator=g).to(DEV)
        ref = o.gemm_nt(a, n)
        ns = o.split_activation(n.view(-1, 320)).view(4, 1024, 320)
        ns._split, ns._alpha = True, 1.0
        assert torch.equal(ref, o.gemm_nt(a, ns))
        return
    M, N, K = {"gemm 8192x640x640 res": (8192, 640, 640), "gemm 32768x320x1280": (32768, 320, 1280), "gemm 1000x328x320 ragged": (1000, 328, 320),
               "gemm 2048x1280x1280 (64x64 tiles)": (2048, 1280, 1280), "gemm 1024x1280x2560 k-slices": (1024, 1280, 2560), "geglu 8192x5120x640": (8192, 5120, 640)}[case]
    a = torch.randn(M, K, generator=g).to(DEV)
    w = o.split_weights((torch.randn(N, K, generator=g) * 0.03).to(DEV))
    kw = dict(bias=torch.randn(N, generator=g).to(DEV))
    if "res" in case:
        kw["residual"] = torch.randn(M, N, generator=g).to(DEV)
    if case.startswith("geglu"):
        kw["act"] = o.ACT_GEGLU
    ref = o.gemm_nt(a, w, **kw)
    assert torch.equal(ref, o.gemm_nt(o.split_activation(a), w, **kw))
    with pytest.raises(o.HipExtensionError):  # the format exists for pre-split weights only
        o.gemm_nt(o.split_activation(a), torch.randn(N, K, generator=g).to(DEV))


def test_producers_store_the_presplit_activation_layout():
    """LayerNorm, both large-slab GroupNorm paths and the GEGLU epilogue with ``split_out``: the bytes they store are
    gmd_split_weights of the plain result (so the contraction that follows reads exactly the values the plain path would split)."""
    o = ops()
    g = torch.Generator().manual_seed(5)
    as_bytes = lambda t: t.contiguous().view(torch.uint8)
    # LayerNorm
    x = torch.randn(4096, 640, generator=g).to(DEV)
    ga, be = torch.randn(640, generator=g).to(DEV), torch.randn(640, generator=g).to(DEV)
    y, ys = o.layernorm(x, ga, be), o.layernorm(x, ga, be, split_out=True)
    assert o.is_asplit(ys) and not o.is_asplit(y) and torch.equal(as_bytes(ys), as_bytes(o.split_activation(y)))
    # GroupNorm: statistics-launch path (no producer statistics attached) and producer-statistics path
    B, H, C = 4, 32, 320
    xg = torch.randn(B, H * H, C, generator=g).to(DEV)
    ga, be = torch.randn(C, generator=g).to(DEV), torch.randn(C, generator=g).to(DEV)
    y, ys = o.groupnorm(xg, B, 32, ga, be, 1e-5, silu=True), o.groupnorm(xg, B, 32, ga, be, 1e-5, silu=True, split_out=True)
    assert o.is_asplit(ys) and torch.equal(as_bytes(ys), as_bytes(o.split_activation(y)))
    w = o.split_weights((torch.randn(C, 9 * 32, generator=g) * 0.05).to(DEV))
    B, H = 4, 64  # 512 tiles of 128 x 160: the statistics-emitting plan
    xc = torch.randn(B, H * H, 32, generator=g).to(DEV)
    yc, _, _ = o.conv3x3(xc, w, B, H, H, colstats=True)
    assert getattr(yc, "_colstats", None) is not None
    before = o.colstats_uses
    y, ys = o.groupnorm(yc, B, 32, ga, be, 1e-5, silu=True), o.groupnorm(yc, B, 32, ga, be, 1e-5, silu=True, split_out=True)
    assert o.colstats_uses == before + 2 and o.is_asplit(ys) and torch.equal(as_bytes(ys), as_bytes(o.split_activation(y)))
    # small slabs take the single-launch kernel: plain layout, no mark (the caller's contraction then splits in the kernel)
    xs = torch.randn(2, 64, 1280, generator=g).to(DEV)
    gs, bs = torch.randn(1280, generator=g).to(DEV), torch.randn(1280, generator=g).to(DEV)
    assert not o.is_asplit(o.groupnorm(xs, 2, 32, gs, bs, 1e-5, silu=True, split_out=True))
    # GEGLU epilogue
    a = torch.randn(8192, 640, generator=g).to(DEV)
    wf = o.split_weights((torch.randn(5120, 640, generator=g) * 0.03).to(DEV))
    bf = torch.randn(5120, generator=g).to(DEV)
    f, fs = o.gemm_nt(a, wf, bias=bf, act=o.ACT_GEGLU), o.gemm_nt(a, wf, bias=bf, act=o.ACT_GEGLU, split_out=True)
    assert o.is_asplit(fs) and torch.equal(as_bytes(fs), as_bytes(o.split_activation(f)))
    # float32 attention epilogue (self-attention on a fused q|k buffer, d = 40; cross-attention with 77 keys, d = 80)
    for (Bq, Nq, Nk, heads, d, fused) in [(2, 1024, 1024, 8, 40, True), (2, 256, 77, 8, 80, False)]:
        C2 = heads * d
        if fused:
            qk = torch.randn(Bq, Nq, 2 * C2, generator=g).to(DEV)
            q_, k_, kc = qk, qk, C2
        else:
            q_, k_, kc = torch.randn(Bq, Nq, C2, generator=g).to(DEV), torch.randn(Bq, Nk, C2, generator=g).to(DEV), 0
        vt = torch.randn(Bq, C2, (Nk + 3) // 4 * 4, generator=g).to(DEV)
        oa, os_ = o.attention(q_, k_, vt, heads, Nk, d ** -0.5, k_col=kc), o.attention(q_, k_, vt, heads, Nk, d ** -0.5, k_col=kc, split_out=True)
        assert o.is_asplit(os_) and torch.equal(as_bytes(os_), as_bytes(o.split_activation(oa)))
    # a launch whose plan cannot write the layout returns the plain tensor, unmarked
    small = o.gemm_nt(a[:96], wf, bias=bf, act=o.ACT_GEGLU, split_out=True)
    assert not o.is_asplit(small) and torch.equal(small, f[:96])
    # only contractions read the layout
    with pytest.raises(o.HipExtensionError):
        o.layernorm(ys, ga, be)  # (the mark lives on the tensor object: views are flagged by the caller, a_split= / x_split=)


def test_unet_forward_is_bit_identical_with_and_without_presplit_activations():
    """The whole SD-1.5-width float32 UNet forward (matrix-core mode) with the pre-split activation format on and off: same bits."""
    from gm_diffusion.components import UNet2DConditionModel

    o = ops()
    u = UNet2DConditionModel(in_channels=8).init_random(3, device=DEV).to(DEV, torch.float32)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 8, 32, 32, generator=g).to(DEV)
    ehs = torch.randn(2, 77, 768, generator=g).to(DEV)
    prev = o.USE_F32SA
    try:
        o.USE_F32SA = True
        a = u(x, 500, encoder_hidden_states=ehs, return_dict=False)[0].clone()
        o.USE_F32SA = False
        b = u(x, 500, encoder_hidden_states=ehs, return_dict=False)[0].clone()
    finally:
        o.USE_F32SA = prev
    assert torch.isfinite(a).all() and torch.equal(a, b)

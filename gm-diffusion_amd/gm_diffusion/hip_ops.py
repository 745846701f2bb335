"""
Thin torch-tensor front end of the C ABI (include/gmd_hip.h).

torch is used here for device memory and streams only: every function validates
its operands on the host, allocates the output with ``torch.empty`` and launches
the hand-written HIP kernel on ``torch.cuda.current_stream()``.  Nothing in this
module computes with torch ops, and nothing falls back to the CPU: tensors that
are not on a HIP device raise ``HipExtensionError``.

Activations are channels-last ``[B, H*W, C]`` (or ``[rows, C]``) in bf16 (MFMA path) or
float32 (parity path); biases / norm parameters / statistics are float32.
"""
from __future__ import annotations

import math
import os
import threading
import time

import torch

from . import profiling
from ._native import ACT_GEGLU, ACT_NONE, ACT_QUICK_GELU, ACT_SILU, GMD_BF16, GMD_F16, GMD_F32, GMD_F32S, GMD_F32SA, GMD_F32SW, HipExtensionError, check, lib

__all__ = [
    "ACT_NONE", "ACT_SILU", "ACT_GEGLU", "ACT_QUICK_GELU", "embedding_lookup", "dpm_step", "ddpm_step", "HipExtensionError", "dtype_code", "gemm_nt", "conv3x3", "attention", "softmax_rows", "set_f32_mode", "f32_split", "split_weights", "scale_weight", "split_attention_ok", "ff_fused_ok", "ff_geglu_fused", "gemm_qkv_vt", "dup_batch",
    "groupnorm_scale_shift", "groupnorm_apply", "groupnorm", "groupnorm_split", "layernorm", "geglu", "timestep_embedding",
    "concat_channels", "cast", "pack_unet_input", "unpack_nchw", "latent_step", "cfg_std_ratio", "hdr_tail",
    "apply_gm_to_sdr", "tmo", "gamut_compress", "stage1_chain", "discretize_u16", "quantize_u8",
]


def dtype_code(dt):
    if dt == torch.float32:
        return GMD_F32
    if dt == torch.bfloat16:
        return GMD_BF16
    if dt == torch.float16:
        return GMD_F16
    raise HipExtensionError(f"unsupported dtype {dt}: the HIP kernels take float32, bfloat16 or float16")


HALF_DTYPES = (torch.bfloat16, torch.float16)

# How float32 contractions (gemm_nt, conv3x3, attention) run:
#   "split" (default): on the matrix cores, every float32 operand taken as f16 hi + f16 lo and each product formed as three
#            float16 MFMA passes with float32 accumulation (GMD_F32S / GMD_F32SW, csrc/gemm_split.hip): float32-grade results
#            (~2^-22 relative per term) at matrix-core speed -- the reference's own float32 numerics
#            (scripts/inference/experiments/formal_improved.py:199), the path that meets the 1e-3 latent-RMS gate;
#   "exact": the float32 FMA kernels on the vector units (bit-for-bit an fp32 FMA chain, ~12x slower end to end).
_F32_DEFAULT = os.environ.get("GMD_F32_MODE", "split")  # the process default; read through f32_mode() / hip_ops.F32_MODE
if _F32_DEFAULT not in ("split", "exact"):
    raise HipExtensionError(f"GMD_F32_MODE={_F32_DEFAULT!r}: expected 'split' or 'exact'")
# Per-thread state: a module call that runs under its own float32 mode (components: _in_own_f32_mode), a graph capture's workspace
# (workspace_scope) and the plan family of a co-running forward (plan_family) are scoped to the CALLING THREAD -- two host threads
# driving a "split" and an "exact" module at the same time never see each other's mode.
_tls = threading.local()


def f32_mode():
    """The float32 contraction mode in effect for the calling thread: its scoped override (f32_mode_scope) or the process default."""
    return getattr(_tls, "f32_mode", None) or _F32_DEFAULT


def __getattr__(name):  # hip_ops.F32_MODE stays readable (tests, tools): the calling thread's effective mode
    if name == "F32_MODE":
        return f32_mode()
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


def set_f32_mode(mode):
    """Switch the PROCESS DEFAULT of float32 contractions between "split" (matrix cores, three float16 products) and "exact" (vector
    FMA); returns the previous default.  Models prepare their weights for the mode in effect when they are placed on the device and
    keep running under it (f32_mode_scope), whatever the default becomes later."""
    global _F32_DEFAULT
    if mode not in ("split", "exact"):
        raise HipExtensionError("f32 mode must be 'split' or 'exact'")
    prev, _F32_DEFAULT = _F32_DEFAULT, mode
    return prev


class f32_mode_scope:
    """Run the ``with`` block under ``mode`` on the calling thread only (other threads keep theirs)."""

    def __init__(self, mode):
        if mode not in ("split", "exact"):
            raise HipExtensionError("f32 mode must be 'split' or 'exact'")
        self.mode = mode

    def __enter__(self):
        self.prev = getattr(_tls, "f32_mode", None)
        _tls.f32_mode = self.mode
        return self

    def __exit__(self, *exc):
        _tls.f32_mode = self.prev
        return False


def f32_split():
    return f32_mode() == "split"


class plan_family:
    """Select the launch-plan family of the calling thread's 16-bit contractions for the ``with`` block (gmd_gemm_plan_family):
    1 = co-running (the dual-UNet pipeline's two overlapped forwards), 0 = a launch that has the chip to itself (the default)."""

    def __init__(self, family):
        self.family = 1 if family else 0

    def __enter__(self):
        self.prev = lib().gmd_gemm_plan_family(self.family)
        return self

    def __exit__(self, *exc):
        lib().gmd_gemm_plan_family(self.prev)
        return False


def check_split_range(what, *tensors, module=None):
    """Range guard of the float32 "split" path.  Every operand of a split contraction is taken as f16 hi + f16 lo: a value
    beyond float16's largest finite number (65504) becomes hi = inf, lo = -inf and the product NaN -- the exact float32 kernels
    have no such limit.  The kernels do not test for it (the split is already the vector-unit bottleneck); the models'
    consumers -- both pipelines on their final latents, ``hdr.decode_to_hdr`` on the decoded images -- call this once per run
    on their OUTPUTS (a NaN reaches them through every following layer).  One reduction and one host synchronisation; only for
    float32 tensors produced under "split".  Raises instead of returning poisoned images."""
    mode = getattr(module, "_f32_mode", None) or f32_mode()
    if mode != "split":
        return
    for t in tensors:
        if t is None or not torch.is_tensor(t) or t.dtype != torch.float32 or not t.is_cuda:
            continue
        if not bool(torch.isfinite(t).all()):
            raise HipExtensionError(
                f"{what}: non-finite values out of the float32 'split' path (matrix cores, three float16 products per float32 "
                "product).  Its operands must stay inside float16's range: an activation beyond 65504 splits into inf / -inf and "
                "poisons the product.  Re-run with GMD_F32_MODE=exact (or hip_ops.set_f32_mode('exact') before the models are "
                "placed on the device): the exact float32 kernels have no range limit.")


def split_weights(w):
    """float32 [N, K] (K % 32 == 0) -> the pre-split hi/lo float16 layout of gmd_split_weights, held in a float32 tensor of
    the same shape (same byte count) that is marked ``_split``: gemm_nt / conv3x3 then take it as the GMD_F32SW W operand."""
    _dev(w)
    _f32(w, "split_weights input")
    if w.dim() != 2 or w.shape[1] % 32:
        raise HipExtensionError("split_weights: [N, K] with K a multiple of 32")
    ws = scale_weight(w)
    out = torch.empty_like(w)
    check(lib().gmd_split_weights(_ptr(ws), _ptr(out), w.shape[0], w.shape[1], w.shape[1], _stream()), "gmd_split_weights")
    out._split = True
    out._alpha = ws._alpha
    return out


def scale_weight(w):
    """``w * 2^s`` with the largest magnitude brought into [2^12, 2^13), marked ``_alpha = 2^-s`` (gemm_nt / conv3x3 fold it
    into their alpha).  The lo half of a split operand is a float16 of ~2^-11 of the value: for typical weight magnitudes
    (1e-2) it would sit in float16's subnormal range and lose most of its bits; scaled, every weight down to 2^-15 of the
    largest keeps a full 11-bit lo.  A power of two, so the scaling itself is exact.  Weight preparation only (reads the
    maximum back to the host)."""
    _dev(w)
    _f32(w, "scale_weight input")
    amax = float(w.abs().max()) if w.numel() else 0.0
    s = 0 if amax == 0.0 or not math.isfinite(amax) else max(-60, min(60, 13 - math.frexp(amax)[1]))
    out = w * (2.0 ** s) if s else w.clone()
    out._alpha = 2.0 ** -s
    return out


# Round 4: float32 activations stored PRE-SPLIT between a producer and the contraction that reads them (GMD_F32SA, gmd_hip.h): the
# GroupNorm / LayerNorm apply kernels and the GEGLU epilogue write [hi 64 B | lo 64 B] per 32 elements, the split kernels then read both
# operands as ready float16 fragments.  Results are bit-identical to the in-kernel split.  GMD_F32SA=0 switches the format off (A/B).
USE_F32SA = os.environ.get("GMD_F32SA", "1") != "0"


def is_asplit(t):
    """The tensor holds float32 values in the pre-split activation layout (only ever the A / W operand of a contraction)."""
    return bool(getattr(t, "_asplit", False))


def _mark_asplit(t):
    t._asplit = True
    return t


def want_split_out(dtype, row_len):
    """A producer may store its float32 output pre-split: split mode on the matrix cores, rows of whole 32-element chunks."""
    return USE_F32SA and dtype == torch.float32 and f32_mode() == "split" and row_len % 32 == 0


def split_activation(x):
    """float32 [..., C] (C % 32 == 0) -> the same values in the pre-split activation layout (tests / tools; the product's producers
    write the layout themselves)."""
    _dev(x)
    _f32(x, "split_activation input")
    C = x.shape[-1]
    if C % 32:
        raise HipExtensionError("split_activation: rows must be multiples of 32 elements")
    x = x.contiguous()
    out = torch.empty_like(x)
    check(lib().gmd_split_weights(_ptr(x), _ptr(out), x.numel() // C, C, C, _stream()), "gmd_split_weights")
    return _mark_asplit(out)


def _contract_code(a, w, K, a_split=False):
    """dtype code of a contraction of ``a`` with the W operand ``w`` over K."""
    if a.dtype != torch.float32:
        if getattr(w, "_split", False):
            raise HipExtensionError("a pre-split float32 weight met a 16-bit activation")
        return dtype_code(a.dtype)
    if a_split:
        if not getattr(w, "_split", False):
            raise HipExtensionError("a pre-split activation needs a pre-split weight operand (GMD_F32SA)")
        return GMD_F32SA
    if getattr(w, "_split", False):
        return GMD_F32SW
    return GMD_F32S if (f32_mode() == "split" and K % 32 == 0) else GMD_F32


def is_half(dt):
    """The two 16-bit element types of the matrix-core path (same kernels, layouts and launch plans)."""
    return dt in HALF_DTYPES


def _dev(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise HipExtensionError(
                "gm_diffusion (MI355X build): tensor is on %s; the compute path is hand-written HIP and has no CPU fallback" % t.device)
        if not t.is_contiguous():
            raise HipExtensionError("non-contiguous tensor passed to a HIP kernel")


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


_WS = {}
# 96 MB of split-K scratch + the tail the library reserves for its arrival counters (gmd_hip.h "WORKSPACE CONTRACT": zero when first
# handed over, left zero by every launch).  The same number goes to every launch AND every plan query.
WS_TAIL_BYTES = 65536
WORKSPACE_BYTES = (96 << 20) + WS_TAIL_BYTES


class workspace_scope:
    """Route every split-K launch the calling thread issues inside the ``with`` block to ``ws`` (a float32 device tensor of
    WORKSPACE_BYTES).  HIP-graph capture uses it: all captures run on torch's one capture stream, so the per-stream
    table below would hand the SAME scratch to two graphs that are later replayed concurrently on different streams."""

    def __init__(self, ws):
        self.ws = ws

    def __enter__(self):
        self.prev = getattr(_tls, "ws_override", None)
        _tls.ws_override = self.ws
        return self.ws

    def __exit__(self, *exc):
        _tls.ws_override = self.prev
        return False


def new_workspace(device):
    ws = torch.empty(WORKSPACE_BYTES // 4, dtype=torch.float32, device=device)
    ws[-(WS_TAIL_BYTES // 4):].zero_()  # the arrival counters of the in-kernel split-K reduction start at zero
    return ws


def _workspace(device):
    """Persistent float32 scratch, one per (device, stream) so that concurrent streams never share it; it lets
    under-filled GEMM/conv launches split K."""
    ov = getattr(_tls, "ws_override", None)
    if ov is not None:
        return ov
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _WS.get(key)
    if ws is None:
        ws = _WS[key] = new_workspace(device)
    return ws


def _rowbias(rb):
    """rowbias may be a float32 tensor [G, N] or a (tensor [G, ld], column offset) pair selecting N columns of a wider
    matrix (all ResnetBlock2D time-embedding projections of a UNet are produced by ONE GEMM)."""
    if rb is None:
        return None, 0
    if isinstance(rb, tuple):
        t, col = rb
        _dev(t)
        _f32(t, "rowbias")
        return t.data_ptr() + col * 4, t.shape[-1]
    _dev(rb)
    _f32(rb, "rowbias")
    return rb.data_ptr(), rb.shape[-1]


def _f32(t, name):
    if t is not None and t.dtype != torch.float32:
        raise HipExtensionError(f"{name} must be float32")
    return t


def _timed(kind):
    """(timer, start event) when a KernelTimer is active and wants ``kind``, else (None, None)."""
    tm = profiling.active()
    if tm is None or not tm.wants(kind):
        return None, None
    return tm, tm.begin()


# ----------------------------------------------------------------------------------------------
# dense contractions
# ----------------------------------------------------------------------------------------------
# Column statistics for a following GroupNorm (gmd_hip.h "Column statistics"): a producer launch called with ``colstats=True``
# leaves {sum, sum of squares} per 64-row block and per bucket of COLSTATS_BUCKET channels when its plan can (the 128-row ring
# kernels' row epilogue); the tensor it returns then carries them as ``._colstats`` and ``groupnorm`` skips its statistics pass.
# 10 divides every SD-1.5 group size (320/32, 640/32, 960/32, 1280/32, 1920/32, 2560/32) and half of a 160-column tile.
COLSTATS_BUCKET = 10
USE_COLSTATS = os.environ.get("GMD_COLSTATS", "1") != "0"  # A/B switch (tests, tools): off = separate statistics launches
colstats_uses = 0     # GroupNorm calls served from producer statistics (tests assert the path is really taken)


def _colstats_buffer(want, dtype, M, N, K, batch, out_dtype, device, code=None):
    """Buffer for the producer's column statistics when this launch can emit them: the 16-bit types, and (round 4) float32 on the
    matrix cores -- ``code`` is the launch's dtype code (GMD_F32S / GMD_F32SW there)."""
    if not want or not USE_COLSTATS or out_dtype != dtype or M % 64 or N % COLSTATS_BUCKET:
        return None
    if is_half(dtype):
        code = dtype_code(dtype)
    elif code not in (GMD_F32S, GMD_F32SW, GMD_F32SA):
        return None
    if not lib().gmd_gemm_colstats_plan(code, M, N, K, batch, WORKSPACE_BYTES, COLSTATS_BUCKET):
        return None
    return torch.empty((M // 64, N // COLSTATS_BUCKET, 2), dtype=torch.float32, device=device)


def gemm_plan_info(dtype, M, N, K, batch=1, geglu=False):
    """(tile rows, tile columns, kernel code, K slices) a 16-bit gemm_nt / conv3x3 launch of these dimensions takes (conv: M =
    B*Hout*Wout, N = Cout, K = 9*Cin); kernel code 283 = the ping-pong kernel.  For tests and measurement tools."""
    import ctypes

    out = (ctypes.c_int * 4)()
    check(lib().gmd_gemm_plan_info(dtype_code(dtype), M, N, K, batch, WORKSPACE_BYTES, int(geglu), ctypes.addressof(out)), "gmd_gemm_plan_info")
    return tuple(out)


def stamp(buf, k, row=None):
    """Measurement only: write the device's 100 MHz counter into ``buf[row[0], k]`` (``buf``: int64 [rows, stride] device tensor,
    ``row``: int32 device scalar or None = row 0) at this point of the current stream -- also inside a graph capture (gmd_stamp)."""
    stride = buf.shape[-1] if buf.dim() == 2 else 0
    check(lib().gmd_stamp(buf.data_ptr(), _ptr(row), stride, int(k), _stream()), "gmd_stamp")


def carry_colstats(dst, src):
    """``dst`` is a view of ``src`` with the same rows x channels content: keep the producer statistics attached."""
    st = getattr(src, "_colstats", None)
    if st is not None:
        dst._colstats = st
    return dst


def gemm_nt(a, w, bias=None, rowbias=None, rows_per_group=0, residual=None, alpha=1.0, act=ACT_NONE,
            out_dtype=None, out=None, ldc=None, colstats=False, a_split=False, split_out=False):
    """``act(alpha * a @ w.T + bias + rowbias[m // rows_per_group] + residual)``.

    a: [M, K] or [batch, M, K]; w: [N, K] or [batch, N, K] (a 2-D operand is shared by the batch)."""
    _dev(a, w, bias, residual, out)
    rb_ptr, rb_ld = _rowbias(rowbias)
    if a.dtype != w.dtype:
        raise HipExtensionError(f"gemm_nt: dtype mismatch {a.dtype} vs {w.dtype}")
    batch = 1
    if a.dim() == 3 or w.dim() == 3:
        batch = a.shape[0] if a.dim() == 3 else w.shape[0]
    M, K = a.shape[-2], a.shape[-1]
    N = w.shape[-2]
    if w.shape[-1] != K:
        raise HipExtensionError(f"gemm_nt: K mismatch {a.shape} vs {w.shape}")
    a_split = bool(a_split) or is_asplit(a)   # ``a_split``: a VIEW of a pre-split activation (views do not carry the mark)
    dt = _contract_code(a, w, K, a_split)
    if getattr(a, "_split", False):
        raise HipExtensionError("gemm_nt: a pre-split weight can only be the W operand")
    alpha = float(alpha) * getattr(a, "_alpha", 1.0) * getattr(w, "_alpha", 1.0)  # weights stored scaled by a power of two
    sA = M * K if a.dim() == 3 else 0
    sW = N * K if w.dim() == 3 else 0
    out_dtype = out_dtype or a.dtype
    ldc = ldc or (N // 2 if act == ACT_GEGLU else N)  # the fused GEGLU epilogue writes [M, N/2]
    if out is None:
        out = torch.empty((batch, M, ldc) if batch > 1 or a.dim() == 3 or w.dim() == 3 else (M, ldc),
                          dtype=out_dtype, device=a.device)
    sC = M * ldc
    sR = 0
    if residual is not None:
        if residual.dtype != a.dtype or residual.shape[-1] != N or residual.shape[-2] != M:
            raise HipExtensionError("gemm_nt: residual must be [M, N] of the input dtype")
        sR = M * N if residual.dim() == 3 else 0
    if bias is not None and bias.numel() != N:
        raise HipExtensionError("gemm_nt: bias must have N elements")
    ws = _workspace(a.device) if batch == 1 else None
    # pre-split OUTPUT (the A operand of the next contraction): where the launch's plan can write it, else the plain tensor
    oc = dtype_code(out_dtype)
    c_split = False
    if (split_out and dt in (GMD_F32S, GMD_F32SW, GMD_F32SA) and batch == 1 and a.dim() == 2 and w.dim() == 2 and residual is None and not colstats
            and want_split_out(out_dtype, ldc) and ldc == (N // 2 if act == ACT_GEGLU else N)
            and lib().gmd_gemm_out_split_ok(M, N, K, int(act == ACT_GEGLU), WORKSPACE_BYTES)):
        oc, c_split = GMD_F32SA, True
    st = None
    if colstats and batch == 1 and a.dim() == 2 and w.dim() == 2 and act != ACT_GEGLU and ldc == N:
        st = _colstats_buffer(True, a.dtype, M, N, K, 1, out_dtype, a.device, code=dt)
    tm = profiling.active()
    tm = tm if tm is not None and tm.wants("gemm_nt") else None
    t0 = tm.begin() if tm else None
    check(lib().gmd_gemm_nt(_ptr(a), _ptr(w), _ptr(out), dt, oc, M, N, K, K, K, ldc, batch, sA, sW, sC,
                            _ptr(_f32(bias, "bias")), rb_ptr, rows_per_group, rb_ld,
                            _ptr(residual), N, sR, float(alpha), act, _ptr(st), COLSTATS_BUCKET if st is not None else 0,
                            _ptr(ws), WORKSPACE_BYTES, _stream()), "gmd_gemm_nt")
    if tm:
        es = a.element_size()
        tm.end("gemm_nt", 2.0 * batch * M * N * K, batch * (M * K + N * K + M * N) * es, t0)
    if st is not None:
        out._colstats = (st, N)
    elif getattr(out, "_colstats", None) is not None:  # a reused `out=` buffer must not keep an earlier producer's statistics
        del out._colstats
    if c_split:
        _mark_asplit(out)
    elif getattr(out, "_asplit", False):  # a reused `out=` buffer must not keep an earlier launch's layout mark
        del out._asplit
    return out


# Fused Q|K|V projection of a self-attention: the V column tiles leave the projection TRANSPOSED (gmd_gemm_qkv_vt), so the batched
# V^T GEMM of every attention disappears.  GMD_QKV_VT=0 keeps the two launches (A/B measurements, tests).
USE_QKV_VT = os.environ.get("GMD_QKV_VT", "1") != "0"


def gemm_qkv_vt(a, w, vt_col0, tokens):
    """a: [B*tokens, K] (16-bit, or float32 with pre-split weights), w: [N, K] = the stacked q | k | v projection weights.  Returns (qk [B*tokens, vt_col0],
    vt [B, N - vt_col0, tokens]) from ONE launch, or None when this launch's plan cannot write transposed V tiles (the caller then
    runs the two projections separately)."""
    _dev(a, w)
    M, K = a.shape
    N = w.shape[0]
    if not (USE_QKV_VT and a.dtype == w.dtype and a.dim() == 2 and w.dim() == 2 and w.shape[1] == K and tokens % 64 == 0 and M % tokens == 0
            and vt_col0 % 8 == 0):
        return None
    if is_half(a.dtype):
        if K % 64:
            return None
        code = dtype_code(a.dtype)
    else:  # float32 on the matrix cores: pre-split weights (and, with the pre-split activation format, a pre-split A operand)
        if not (a.dtype == torch.float32 and getattr(w, "_split", False) and K % 32 == 0):
            return None
        code = GMD_F32SA if is_asplit(a) else GMD_F32SW
    if not lib().gmd_gemm_qkv_vt_ok(code, M, N, K, vt_col0, tokens, WORKSPACE_BYTES):
        return None
    qk = torch.empty((M, vt_col0), dtype=a.dtype, device=a.device)
    vt = torch.empty((M // tokens, N - vt_col0, tokens), dtype=a.dtype, device=a.device)
    ws = _workspace(a.device)
    tm = profiling.active()
    tm = tm if tm is not None and tm.wants("gemm_nt") else None
    t0 = tm.begin() if tm else None
    check(lib().gmd_gemm_qkv_vt(_ptr(a), _ptr(w), _ptr(qk), _ptr(vt), code, M, N, K, vt_col0, vt_col0, tokens, tokens,
                                float(getattr(w, "_alpha", 1.0)), _ptr(ws), WORKSPACE_BYTES, _stream()), "gmd_gemm_qkv_vt")
    if tm:
        tm.end("gemm_nt", 2.0 * M * N * K, (M * K + N * K + M * N) * a.element_size(), t0)
    return qk, vt


# The GEGLU feed-forward as one launch (csrc/ff_fused.hip) where the kernel is instantiated; GMD_FUSED_FF=0 keeps the two
# GEMM launches (A/B measurements, tests).
USE_FUSED_FF = os.environ.get("GMD_FUSED_FF", "1") != "0"
# One 128-row workgroup per CU: below ~7/8 of the chip's 256 CUs the two tiled GEMM launches (two workgroups per CU, 512+ tiles)
# are faster -- tools/bench_ff.py: M = 32768 90 vs 124 us, M = 16384 (half the chip) 82 vs 67 us.
FUSED_FF_MIN_ROWS = int(os.environ.get("GMD_FUSED_FF_MIN_ROWS", str(224 * 128)))
# ... inside a co-running forward (plan family 1) from half the chip up: beside the other stream's kernels CU-time, not the launch's
# own time, is what counts, and 128 fused workgroups cost less of it than the 1280 + 256 of the two launches (round 5, interleaved
# A/B of whole bench.py runs: 752.7 -> 751.3 ms per batch, 3 of 3 rounds; fusion off altogether: 757.4)
FUSED_FF_MIN_ROWS_CO_RUN = int(os.environ.get("GMD_FUSED_FF_MIN_ROWS_CO_RUN", str(128 * 128)))


def ff_fused_ok(x, C, min_rows=None):
    if min_rows is None:
        min_rows = FUSED_FF_MIN_ROWS_CO_RUN if lib().gmd_gemm_plan_family(-1) == 1 else FUSED_FF_MIN_ROWS
    return (USE_FUSED_FF and is_half(x.dtype) and x.dim() == 2 and x.shape[0] >= min_rows and
            bool(lib().gmd_ff_geglu_fused_supported(dtype_code(x.dtype), x.shape[0], C)))


def ff_geglu_fused(x, w1i, b1i, w2, b2, residual):
    """``(value * gelu(gate)) @ w2.T + b2 + residual`` with ``[value | gate] = x @ w1i.T + b1i`` (interleaved rows)."""
    _dev(x, w1i, b1i, w2, b2, residual)
    M, C = x.shape
    if w1i.shape != (8 * C, C) or w2.shape != (C, 4 * C) or residual.shape != x.shape or not (x.dtype == w1i.dtype == w2.dtype == residual.dtype):
        raise HipExtensionError("ff_geglu_fused: operand shapes / dtypes inconsistent")
    y = torch.empty_like(x)
    tm = profiling.active()
    tm = tm if tm is not None and tm.wants("gemm_nt") else None
    t0 = tm.begin() if tm else None
    check(lib().gmd_ff_geglu_fused(_ptr(x), _ptr(w1i), _ptr(_f32(b1i, "b1")), _ptr(w2), _ptr(_f32(b2, "b2")), _ptr(residual), _ptr(y),
                                   dtype_code(x.dtype), M, C, _stream()), "gmd_ff_geglu_fused")
    if tm:  # both products' algorithmic FLOPs; bytes: x, residual, y once + the weights
        es = x.element_size()
        tm.end("gemm_nt", 2.0 * M * C * 8 * C + 2.0 * M * 4 * C * C, (3 * M * C + 12 * C * C) * es, t0)
    return y


def conv3x3(x, w, B, H, W, bias=None, rowbias=None, residual=None, stride=1, upsample=False, pad_mode=0, out_dtype=None,
            colstats=False, x_split=False):
    """x: [B, H*W, Cin]; w: [Cout, 9*Cin] (tap-major); returns ([B, Hout*Wout, Cout], Hout, Wout)."""
    _dev(x, w, bias, residual)
    rb_ptr, rb_ld = _rowbias(rowbias)
    cin, cout = x.shape[-1], w.shape[0]
    if x.numel() != B * H * W * cin or w.shape[1] != 9 * cin or x.dtype != w.dtype:
        raise HipExtensionError(f"conv3x3: shape/dtype mismatch x={tuple(x.shape)} w={tuple(w.shape)} B,H,W={B},{H},{W}")
    if upsample:
        ho, wo = 2 * H, 2 * W
    elif pad_mode == 1:
        ho, wo = (H + 1 - 3) // 2 + 1, (W + 1 - 3) // 2 + 1
    else:
        ho, wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    out_dtype = out_dtype or x.dtype
    y = torch.empty((B, ho * wo, cout), dtype=out_dtype, device=x.device)
    if residual is not None and (residual.numel() != y.numel() or residual.dtype != x.dtype):
        raise HipExtensionError("conv3x3: residual shape/dtype mismatch")
    if rowbias is not None and (rowbias[0] if isinstance(rowbias, tuple) else rowbias).shape[0] != B:
        raise HipExtensionError("conv3x3: rowbias must have one row per sample")
    ws = _workspace(x.device)
    x_split = bool(x_split) or is_asplit(x)
    code = _contract_code(x, w, cin, x_split)
    st = _colstats_buffer(colstats, x.dtype, B * ho * wo, cout, 9 * cin, 1, out_dtype, x.device, code=code)
    tm = profiling.active()
    tm = tm if tm is not None and tm.wants("conv3x3") else None
    t0 = tm.begin() if tm else None
    check(lib().gmd_conv3x3(_ptr(x), _ptr(w), _ptr(y), code, dtype_code(out_dtype), B, H, W, cin, cout,
                            stride, int(upsample), pad_mode, _ptr(_f32(bias, "bias")), rb_ptr, rb_ld,
                            _ptr(residual), float(getattr(w, "_alpha", 1.0)), _ptr(st), COLSTATS_BUCKET if st is not None else 0,
                            _ptr(ws), WORKSPACE_BYTES, _stream()), "gmd_conv3x3")
    if tm:
        tm.end("conv3x3", 2.0 * B * ho * wo * cout * 9 * cin, x.numel() * x.element_size() + w.numel() * w.element_size()
               + y.numel() * y.element_size(), t0)
    if st is not None:
        y._colstats = (st, cout)
    return y, ho, wo


USE_CONV_GN_FUSION = os.environ.get("GMD_CONV_GN", "1") != "0"  # (the switch: A/B measurements only)


def conv3x3_groupnorm(x, w, B, H, W, groups, gamma, beta, eps, silu=True, bias=None, rowbias=None, residual=None, want_raw=False,
                      colstats=False, split_out=False):
    """GroupNorm(+SiLU) of conv3x3(x): returns (raw or None, normalised).  Where the convolution runs split-K and the group slices
    are small (the 16x16 / 8x8 UNet levels) ONE GroupNorm launch sums the partial slabs, applies the convolution's epilogue and
    normalises (gmd_conv3x3_groupnorm: no reduce launch, the raw tensor is written only if ``want_raw``); everywhere else this
    is conv3x3 (+ producer statistics if ``colstats``) followed by groupnorm -- bit-identical either way."""
    _dev(x, w, bias, residual, gamma, beta)
    cin, cout = x.shape[-1], w.shape[0]
    code = _contract_code(x, w, cin, is_asplit(x)) if x.dtype == w.dtype and w.shape[1] == 9 * cin else -1
    if not (USE_CONV_GN_FUSION and code >= 0 and
            lib().gmd_conv3x3_gn_fusable(code, B, H, W, cin, cout, 1, 0, 0, groups, WORKSPACE_BYTES)):
        y, _, _ = conv3x3(x, w, B, H, W, bias=bias, rowbias=rowbias, residual=residual, colstats=colstats)
        return (y if want_raw else None), groupnorm(y, B, groups, gamma, beta, eps, silu=silu, split_out=split_out)
    if x.numel() != B * H * W * cin:
        raise HipExtensionError(f"conv3x3_groupnorm: shape mismatch x={tuple(x.shape)} B,H,W={B},{H},{W}")
    if residual is not None and (residual.numel() != B * H * W * cout or residual.dtype != x.dtype):
        raise HipExtensionError("conv3x3_groupnorm: residual shape/dtype mismatch")
    if rowbias is not None and (rowbias[0] if isinstance(rowbias, tuple) else rowbias).shape[0] != B:
        raise HipExtensionError("conv3x3_groupnorm: rowbias must have one row per sample")
    rb_ptr, rb_ld = _rowbias(rowbias)
    yn = torch.empty((B, H * W, cout), dtype=x.dtype, device=x.device)
    yr = torch.empty_like(yn) if want_raw else None
    ws = _workspace(x.device)
    tm = profiling.active()
    tm = tm if tm is not None and tm.wants("conv3x3") else None
    t0 = tm.begin() if tm else None
    check(lib().gmd_conv3x3_groupnorm(_ptr(x), _ptr(w), _ptr(yr), _ptr(yn), code, B, H, W, cin, cout, 1, 0, 0, _ptr(_f32(bias, "bias")),
                                      rb_ptr, rb_ld, _ptr(residual), float(getattr(w, "_alpha", 1.0)), groups, float(eps),
                                      _ptr(_f32(gamma, "gamma")), _ptr(_f32(beta, "beta")), int(silu), _ptr(ws), WORKSPACE_BYTES, _stream()),
          "gmd_conv3x3_groupnorm")
    if tm:  # counted as the convolution it replaces (its FLOPs; the GroupNorm's bytes ride along)
        tm.end("conv3x3", 2.0 * B * H * W * cout * 9 * cin, x.numel() * x.element_size() + w.numel() * w.element_size()
               + yn.numel() * yn.element_size(), t0)
    return yr, yn


SPLIT_ATTENTION_HEAD_DIMS = (40, 64, 80, 160)  # float32 flash kernel (attention_split.hip): SD-1.5's head dims and SDXL's 64


def split_attention_ok(dtype, d):
    """True when ``attention`` takes float32 tensors of this head dim (F32_MODE 'split')."""
    return dtype == torch.float32 and f32_split() and d in SPLIT_ATTENTION_HEAD_DIMS


def attention(q, k, vt, heads, nk, scale, k_col=0, causal=False, split_out=False):
    """q: [B, Nq, ldq] with Q in columns [0, H*D); k: [B, Nk, ldk] with K in columns [k_col, k_col+H*D)
    (q and k may be the same fused-projection buffer); vt: [B, H*D, ldvt] (V transposed, keys contiguous).
    ``split_out`` (float32 matrix-core mode): store O pre-split for the out-projection (the result carries the mark, is_asplit)."""
    _dev(q, k, vt)
    B, nq = q.shape[0], q.shape[1]
    hd = vt.shape[1]
    d = hd // heads
    ldq, ldk = q.shape[2], k.shape[2]
    if k.shape[1] < nk or vt.shape[2] < nk or k_col + hd > ldk or hd > ldq:
        raise HipExtensionError("attention: operand shapes inconsistent")
    o = torch.empty((B, nq, hd), dtype=q.dtype, device=q.device)
    if q.dtype == torch.float32:
        if not (f32_split() and d in SPLIT_ATTENTION_HEAD_DIMS):
            raise HipExtensionError("attention: float32 runs on the split (three float16 products) kernel, head dims 40/64/80/160, "
                                    "in F32_MODE 'split' only; the exact float32 path composes gemm_nt + softmax_rows")
        so = bool(split_out) and want_split_out(q.dtype, hd)
        code = GMD_F32SA if so else GMD_F32S
    else:
        so = False
        code = dtype_code(q.dtype)
    tm = profiling.active()
    tm = tm if tm is not None and tm.wants("attention") else None
    t0 = tm.begin() if tm else None
    check(lib().gmd_attention(_ptr(q), _ptr(k) + k_col * k.element_size(), _ptr(vt), _ptr(o), code, B, heads, d,
                              nq, nk, ldq, ldk, vt.shape[2], hd, nq * ldq, k.shape[1] * ldk, hd * vt.shape[2], nq * hd,
                              float(scale), int(bool(causal)), _stream()), "gmd_attention")
    if tm:  # QK^T + PV, algorithmic head dim (padding not counted)
        tm.end("attention", 4.0 * B * nq * nk * hd, (2 * B * nq * hd + 2 * B * nk * hd) * q.element_size(), t0)
    return _mark_asplit(o) if so else o


def softmax_rows(s, cols, scale, out_dtype, ldp=None, causal_nq=0):
    _dev(s)
    _f32(s, "softmax_rows input")
    lds_ = s.shape[-1]
    rows = s.numel() // lds_
    ldp = ldp or lds_
    p = torch.empty(s.shape[:-1] + (ldp,), dtype=out_dtype, device=s.device)
    check(lib().gmd_softmax_rows(_ptr(s), lds_, _ptr(p), dtype_code(out_dtype), ldp, rows, cols, float(scale), int(causal_nq), _stream()),
          "gmd_softmax_rows")
    return p


# ----------------------------------------------------------------------------------------------
# normalisation / elementwise
# ----------------------------------------------------------------------------------------------
def groupnorm_scale_shift(x, B, groups, gamma, beta, eps):
    """x: [B, HW, C] -> float32 [B, C, 2] = {rstd*gamma, beta - mean*rstd*gamma}."""
    _dev(x, gamma, beta)
    C = x.shape[-1]
    HW = x.numel() // (B * C)
    nsplit = lib().gmd_groupnorm_nsplit(HW)
    ws = torch.empty(B * nsplit * groups * 2, dtype=torch.float32, device=x.device)
    ss = torch.empty((B, C, 2), dtype=torch.float32, device=x.device)
    check(lib().gmd_groupnorm_stats(_ptr(x), dtype_code(x.dtype), B, HW, C, groups, float(eps), _ptr(_f32(gamma, "gamma")),
                                    _ptr(_f32(beta, "beta")), _ptr(ws), _ptr(ss), _stream()), "gmd_groupnorm_stats")
    return ss


def groupnorm_apply(x, B, ss, silu):
    _dev(x, ss)
    C = x.shape[-1]
    HW = x.numel() // (B * C)
    y = torch.empty_like(x)
    check(lib().gmd_groupnorm_apply(_ptr(x), _ptr(y), dtype_code(x.dtype), B, HW, C, _ptr(ss), int(silu), _stream()),
          "gmd_groupnorm_apply")
    return y


# bytes of one (sample, group) slab up to which the single-launch kernel wins (tools/bench_gn.py, MI355X): 5-21 us against
# 14-27 us for the three-launch path on the 16x16 / 8x8 UNet levels.  The kernel walks one group's channels of every pixel,
# so it needs long enough channel rows: with 16-byte accesses (C/G a multiple of 8 bf16) it wins up to 40 KiB, with 8- or
# 4-byte accesses only up to 24 KiB; beyond that the strided re-reads lose to the row-contiguous split kernels.
GN_FUSED_MAX_SLAB = 24 * 1024
GN_FUSED_MAX_SLAB_VEC16 = 40 * 1024


def _usable_colstats(x, B, HW, C, cpg):
    """Producer statistics attached to ``x`` (one producer, or the two halves of a channel concatenation) when they cover
    exactly this tensor: ((stats_a, Ca), (stats_b, Cb) | None), else None."""
    st = getattr(x, "_colstats", None)
    if st is None or HW % 64 or cpg % COLSTATS_BUCKET:
        return None
    parts = st if isinstance(st, list) else [st]
    rows = B * HW // 64
    if sum(c for _, c in parts) != C or any(t.shape[0] != rows or t.shape[1] * COLSTATS_BUCKET != c for t, c in parts):
        return None
    return (parts[0], parts[1] if len(parts) == 2 else None) if len(parts) <= 2 else None


def groupnorm(x, B, groups, gamma, beta, eps, silu=False, split_out=False):
    """GroupNorm(+SiLU).  A tensor that still carries its producer's column statistics (``colstats=True`` of gemm_nt /
    conv3x3, possibly through concat_channels) is normalised in one pass over it; otherwise small group slabs (16x16 / 8x8
    UNet levels) take the single-launch fused kernel and larger ones the split-statistics path (partial + apply)."""
    _dev(x, gamma, beta)
    C = x.shape[-1]
    HW = x.numel() // (B * C)
    epw = 2 if is_half(x.dtype) else 1
    cpg = C // groups
    vec16 = (cpg * x.element_size()) % 16 == 0 and (C * x.element_size()) % 16 == 0
    # algorithmic HBM bytes of a GroupNorm: the activation is read once and written once (the statistics pass of the split
    # path re-reads it: that second read is what the roofline fraction of this kind exposes)
    nbytes = 2 * x.numel() * x.element_size()
    if is_asplit(x):
        raise HipExtensionError("groupnorm: the input is a pre-split activation (only contractions read that layout)")
    # ``split_out``: store the float32 result pre-split for the contraction that reads it (the two large-slab paths; the single-launch
    # kernel of the small levels keeps the plain layout) -- the result then carries the mark (is_asplit)
    so = bool(split_out) and want_split_out(x.dtype, C)
    st = _usable_colstats(x, B, HW, C, cpg)
    if st is not None:
        global colstats_uses
        colstats_uses += 1
        (sa, ca), sb = st
        y = torch.empty_like(x)
        tm, t0 = _timed("groupnorm")
        check(lib().gmd_groupnorm_colstats(_ptr(x), _ptr(y), GMD_F32SA if so else dtype_code(x.dtype), B, HW, C, groups, float(eps),
                                           _ptr(_f32(gamma, "gamma")), _ptr(_f32(beta, "beta")), _ptr(sa), ca,
                                           _ptr(sb[0]) if sb is not None else None, COLSTATS_BUCKET, int(silu), _stream()),
              "gmd_groupnorm_colstats")
        if tm:
            tm.end("groupnorm", 0.0, nbytes, t0)
        return _mark_asplit(y) if so else y
    if cpg % epw == 0 and HW * cpg * x.element_size() <= (GN_FUSED_MAX_SLAB_VEC16 if vec16 else GN_FUSED_MAX_SLAB):
        y = torch.empty_like(x)
        tm, t0 = _timed("groupnorm")
        check(lib().gmd_groupnorm_fused(_ptr(x), _ptr(y), dtype_code(x.dtype), B, HW, C, groups, float(eps), _ptr(_f32(gamma, "gamma")),
                                        _ptr(_f32(beta, "beta")), int(silu), _stream()), "gmd_groupnorm_fused")
        if tm:
            tm.end("groupnorm", 0.0, nbytes, t0)
        return y
    nsplit = lib().gmd_groupnorm_nsplit(HW)
    ws = torch.empty(B * nsplit * groups * 2, dtype=torch.float32, device=x.device)
    y = torch.empty_like(x)
    tm, t0 = _timed("groupnorm")
    check(lib().gmd_groupnorm_split(_ptr(x), _ptr(y), GMD_F32SA if so else dtype_code(x.dtype), B, HW, C, groups, float(eps), _ptr(_f32(gamma, "gamma")),
                                    _ptr(_f32(beta, "beta")), _ptr(ws), int(silu), _stream()), "gmd_groupnorm_split")
    if tm:
        tm.end("groupnorm", 0.0, nbytes, t0)
    return _mark_asplit(y) if so else y


def groupnorm_split(x, B, groups, gamma, beta, eps, silu=False):
    return groupnorm_apply(x, B, groupnorm_scale_shift(x, B, groups, gamma, beta, eps), silu)


def layernorm(x, gamma, beta, eps=1e-5, split_out=False):
    """``split_out``: store the float32 result pre-split for the projections that read it (the result carries the mark, is_asplit)."""
    _dev(x, gamma, beta)
    if is_asplit(x):
        raise HipExtensionError("layernorm: the input is a pre-split activation (only contractions read that layout)")
    C = x.shape[-1]
    so = bool(split_out) and want_split_out(x.dtype, C)
    y = torch.empty_like(x)
    tm, t0 = _timed("layernorm")
    check(lib().gmd_layernorm(_ptr(x), _ptr(y), GMD_F32SA if so else dtype_code(x.dtype), x.numel() // C, C, _ptr(_f32(gamma, "gamma")),
                              _ptr(_f32(beta, "beta")), float(eps), _stream()), "gmd_layernorm")
    if tm:
        tm.end("layernorm", 0.0, 2 * x.numel() * x.element_size(), t0)
    return _mark_asplit(y) if so else y


def geglu(x):
    _dev(x)
    F2 = x.shape[-1]
    y = torch.empty(x.shape[:-1] + (F2 // 2,), dtype=x.dtype, device=x.device)
    check(lib().gmd_geglu(_ptr(x), _ptr(y), dtype_code(x.dtype), x.numel() // F2, F2 // 2, _stream()), "gmd_geglu")
    return y


def timestep_embedding(t_dev, B, dim, dtype, flip_sin_to_cos=True, freq_shift=0.0):
    """t_dev: float32 device scalar (shape [1])."""
    _dev(t_dev)
    _f32(t_dev, "timestep")
    out = torch.empty((B, dim), dtype=dtype, device=t_dev.device)
    check(lib().gmd_timestep_embedding(_ptr(t_dev), _ptr(out), dtype_code(dtype), B, dim, int(flip_sin_to_cos),
                                       float(freq_shift), _stream()), "gmd_timestep_embedding")
    return out


def concat_channels(a, b):
    _dev(a, b)
    ca, cb = a.shape[-1], b.shape[-1]
    rows = a.numel() // ca
    if b.numel() // cb != rows or a.dtype != b.dtype:
        raise HipExtensionError("concat_channels: row count / dtype mismatch")
    out = torch.empty(a.shape[:-1] + (ca + cb,), dtype=a.dtype, device=a.device)
    tm, t0 = _timed("concat")
    check(lib().gmd_concat_channels(_ptr(a), ca, _ptr(b), cb, _ptr(out), dtype_code(a.dtype), rows, _stream()),
          "gmd_concat_channels")
    if tm:
        tm.end("concat", 0.0, 2 * out.numel() * out.element_size(), t0)
    sa, sb = getattr(a, "_colstats", None), getattr(b, "_colstats", None)
    if sa is not None and sb is not None and not isinstance(sa, list) and not isinstance(sb, list):
        out._colstats = [sa, sb]  # a following GroupNorm reads the two producers' statistics side by side
    return out


_SIDE_STREAMS = {}


def side_stream(device):
    """The second HIP stream of ``device`` (the GM UNet's, stable_diffusion_dual_unet.py).  ONE per device and process, shared by
    every pipeline object: HIP multiplexes its streams onto a few hardware queues in the order they are first used, and a stream
    drawn from torch's pool after a number of others (graph captures take several) can land on the queue the main stream uses --
    the two UNets then run one after the other.  Measured in bench.py: a second pipeline with its own late stream 2036 ms per batch,
    with this shared one 1817 ms (float32 path).  Default priority and HIP's default queue count on purpose: a high-priority GM
    stream costs 1393 instead of 851 ms per batch, GPU_MAX_HW_QUEUES=8 1158 ms, =2 852 ms (tools/ab_queues.sh, bfloat16)."""
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    s = _SIDE_STREAMS.get(key)
    if s is None:
        s = _SIDE_STREAMS[key] = _stream_beside_current(device)
    return s


_SPIN_TICKS = 1_000_000  # ~0.4 ms of torch.cuda._sleep at the MI355X shader clock
_CAPTURES_IN_FLIGHT = 0  # HIP-graph captures this package has open (components: graphed_forward) -- on ANY stream or thread
_CAPTURES_LOCK = threading.Lock()  # the count is deliberately process-wide (a capture on another thread matters too): guarded
_PROBE_WARNED = False


class capture_in_flight:
    """Marks a HIP-graph capture of this package as open: the side-stream probe synchronises, which would invalidate a capture
    in progress on another stream, so it does not run while one is open."""

    def __enter__(self):
        global _CAPTURES_IN_FLIGHT
        with _CAPTURES_LOCK:
            _CAPTURES_IN_FLIGHT += 1

    def __exit__(self, *exc):
        global _CAPTURES_IN_FLIGHT
        with _CAPTURES_LOCK:
            _CAPTURES_IN_FLIGHT -= 1
        return False


def _probe_note(msg):
    global _PROBE_WARNED
    if not _PROBE_WARNED:
        _PROBE_WARNED = True
        import warnings

        warnings.warn("gm_diffusion: " + msg + " -- the GM UNet's stream may share the hardware queue of the main stream (the two "
                      "UNets then run one after the other, ~20 % slower; results are unaffected).  GMD_SIDE_STREAM_SKIP=<n> picks "
                      "the (n+1)-th new stream without probing.", RuntimeWarning, stacklevel=3)


def _stream_beside_current(device, candidates=8):
    """A new stream that really runs BESIDE the current one.  Of the streams a process creates, about every fourth lands on the
    hardware queue of the null stream (tools/stream_queue_probe.py: pool streams 6, 10, 14, ... of a fresh process), so the
    first candidate is not taken on trust: one single-thread spinning kernel on each of the two streams must take the time of one
    spin, not of two.  ~3 ms, once per device and process.  The probe is wall-clock based, so it is an optimisation with escape
    hatches, never a reason to fail: it does not run while a graph capture is open (its synchronisation would invalidate the
    capture) or when GMD_SIDE_STREAM_SKIP=<n> names the stream to take (the (n+1)-th created here; a host that knows its queue
    layout, or a shared / busy GPU where timing is noise); when no candidate passes -- or the probe cannot run -- the first
    candidate is used and ONE RuntimeWarning says so."""
    skip = os.environ.get("GMD_SIDE_STREAM_SKIP")
    if skip is not None:
        try:
            n = max(0, int(skip))
        except ValueError:  # a malformed escape hatch is no reason to fail either: say so once and probe as usual
            _probe_note(f"GMD_SIDE_STREAM_SKIP={skip!r} is not an integer; ignored")
            n = None
        if n is not None:
            junk = [torch.cuda.Stream(device=device) for _ in range(n)]
            st = torch.cuda.Stream(device=device)
            del junk
            return st
    first = torch.cuda.Stream(device=device)
    try:
        with torch.cuda.device(device):
            if _CAPTURES_IN_FLIGHT > 0 or torch.cuda.is_current_stream_capturing():
                _probe_note("side-stream probe skipped (a HIP-graph capture is in progress)")
                return first
            cur = torch.cuda.current_stream()

            def wall(fn, st):
                cur.synchronize(); st.synchronize()  # the two streams of the probe only: no device-wide synchronisation
                t0 = time.perf_counter()
                fn()
                cur.synchronize(); st.synchronize()
                return time.perf_counter() - t0

            def both(st):
                with torch.cuda.stream(st):
                    torch.cuda._sleep(_SPIN_TICKS)
                torch.cuda._sleep(_SPIN_TICKS)

            torch.cuda._sleep(_SPIN_TICKS)
            one = min(wall(lambda: torch.cuda._sleep(_SPIN_TICKS), first) for _ in range(2))
            st = first
            for _ in range(candidates):
                both(st)
                if min(wall(lambda: both(st), st) for _ in range(2)) < 1.5 * one:
                    return st
                st = torch.cuda.Stream(device=device)
            _probe_note(f"none of {candidates} candidate streams ran beside the current stream in the probe (busy or shared GPU?)")
    except Exception as e:  # pragma: no cover -- the probe is an optimisation, never a reason to fail
        _probe_note(f"side-stream probe failed ({e!r})")
    return first


def dup_batch(t):
    """[B, ...] -> [2B, ...] with both halves equal to ``t`` (one read, two writes; the CFG duplication of a shared-prefix tensor)."""
    _dev(t)
    out = torch.empty((2 * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    nbytes = t.numel() * t.element_size()
    if nbytes % 16 or os.environ.get("GMD_DUP_KERNEL", "1") == "0":  # (the switch: A/B measurements only)
        out[: t.shape[0]].copy_(t)
        out[t.shape[0]:].copy_(t)
        return out
    check(lib().gmd_dup_batch(_ptr(t), _ptr(out), nbytes, _stream()), "gmd_dup_batch")
    return out


def embedding_lookup(ids, table, pos):
    """ids: integer [B, T] on the device; table [vocab, C], pos [>=T, C] -> [B, T, C] = table[ids] + pos[:T]."""
    _dev(ids, table, pos)
    if table.dtype != pos.dtype or table.shape[1] != pos.shape[1] or pos.shape[0] < ids.shape[1]:
        raise HipExtensionError("embedding_lookup: table / position shapes or dtypes inconsistent")
    B, T = ids.shape
    ids32 = ids.to(torch.int32).contiguous()
    out = torch.empty((B, T, table.shape[1]), dtype=table.dtype, device=table.device)
    check(lib().gmd_embedding_lookup(_ptr(ids32), _ptr(table), _ptr(pos), _ptr(out), dtype_code(table.dtype), B * T, T, table.shape[1],
                                     table.shape[0], _stream()), "gmd_embedding_lookup")
    return out


def cast(x, dtype):
    _dev(x)
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    check(lib().gmd_cast(_ptr(x), dtype_code(x.dtype), _ptr(out), dtype_code(dtype), x.numel(), _stream()), "gmd_cast")
    return out


# ----------------------------------------------------------------------------------------------
# latent-side
# ----------------------------------------------------------------------------------------------
def pack_unet_input(src0, src1, dup, cp, dtype, out=None):
    """src0 [B,C0,h,w] (+ src1 [B,C1,h,w]) float32 NCHW -> [dup*B, h*w, cp] channels-last of `dtype`
    (written into `out` when given: the static input buffer of a captured graph)."""
    _dev(src0, src1, out)
    _f32(src0, "src0")
    _f32(src1, "src1")
    B, c0 = src0.shape[0], src0.shape[1]
    hw = src0.shape[2] * src0.shape[3]
    c1 = 0 if src1 is None else src1.shape[1]
    if out is None:
        out = torch.empty((dup * B, hw, cp), dtype=dtype, device=src0.device)
    elif tuple(out.shape) != (dup * B, hw, cp) or out.dtype != dtype:
        raise HipExtensionError("pack_unet_input: `out` has the wrong shape/dtype")
    check(lib().gmd_pack_unet_input(_ptr(src0), c0, _ptr(src1), c1, B, hw, dup, _ptr(out), cp, dtype_code(dtype), _stream()),
          "gmd_pack_unet_input")
    return out


def unpack_nchw(x, B, C, h, w):
    """x: [B, h*w, ld] -> float32 [B, C, h, w] (first C channels)."""
    _dev(x)
    out = torch.empty((B, C, h, w), dtype=torch.float32, device=x.device)
    check(lib().gmd_unpack_nchw(_ptr(x), dtype_code(x.dtype), x.shape[-1], B, C, h * w, _ptr(out), _stream()), "gmd_unpack_nchw")
    return out


def cfg_std_ratio(eps_pair, guidance_scale):
    _dev(eps_pair)
    B = eps_pair.shape[0] // 2
    ratio = torch.empty(B, dtype=torch.float32, device=eps_pair.device)
    check(lib().gmd_cfg_std_ratio(_ptr(_f32(eps_pair, "eps")), B, eps_pair[0].numel(), float(guidance_scale), _ptr(ratio), _stream()),
          "gmd_cfg_std_ratio")
    return ratio


def latent_step(eps_in, x, mode, coefs, do_cfg, guidance_scale, cur_sample=None, hist=(), ratio=None,
                guidance_rescale=0.0, want_x0=False):
    """Fused CFG + x0 + PLMS update.  coefs = (sample_coeff, alpha_delta, denom, sqrt_alpha, sqrt_one_minus_alpha).
    Returns (eps, x_prev, x0|None)."""
    _dev(eps_in, x, cur_sample, ratio, *hist)
    for t in (eps_in, x, cur_sample, *hist):
        _f32(t, "latent tensors")
    B = x.shape[0]
    chw = x.shape[1:].numel()
    eps_out = torch.empty_like(x)
    x_prev = torch.empty_like(x)
    x0 = torch.empty_like(x) if want_x0 else None
    h = list(hist) + [None] * (3 - len(hist))
    sc, ad, dn, sa, s1 = (float(c) for c in coefs)
    check(lib().gmd_latent_step(_ptr(eps_in), _ptr(x), _ptr(cur_sample), _ptr(h[0]), _ptr(h[1]), _ptr(h[2]), B, chw,
                                int(do_cfg), float(guidance_scale), _ptr(ratio), float(guidance_rescale), mode,
                                sc, ad, dn, sa, s1, _ptr(eps_out), _ptr(x_prev), _ptr(x0), _stream()), "gmd_latent_step")
    return eps_out, x_prev, x0


def dpm_step(eps_in, x, order, coefs, do_cfg, guidance_scale, m1=None, ratio=None, guidance_rescale=0.0, want_x0=False):
    """Fused CFG + x0 + DPM-Solver++ (orders 1-2) update.  coefs = (sigma_s0, alpha_s0, c_x, c_m, c_h, inv_r0, sqrt_alpha,
    sqrt_one_minus_alpha).  Returns (m0, x_prev, x0|None)."""
    _dev(eps_in, x, m1, ratio)
    for t in (eps_in, x, m1):
        _f32(t, "latent tensors")
    B = x.shape[0]
    chw = x.shape[1:].numel()
    m0 = torch.empty_like(x)
    x_prev = torch.empty_like(x)
    x0 = torch.empty_like(x) if want_x0 else None
    c = [float(v) for v in coefs]
    check(lib().gmd_dpm_step(_ptr(eps_in), _ptr(x), _ptr(m1), B, chw, int(do_cfg), float(guidance_scale), _ptr(ratio),
                             float(guidance_rescale), int(order), *c, _ptr(m0), _ptr(x_prev), _ptr(x0), _stream()), "gmd_dpm_step")
    return m0, x_prev, x0


def ddpm_step(eps_in, x, coefs, do_cfg, guidance_scale, noise=None, ratio=None, guidance_rescale=0.0, clip_range=None,
              want_x0=False):
    """Fused CFG + x0 + DDPM ancestral update.  coefs = (sched_sqrt_alpha, sched_sqrt_one_minus_alpha, x0_coeff, xt_coeff,
    noise_scale, sqrt_alpha, sqrt_one_minus_alpha); ``noise`` is None at the last step (t == 0); ``clip_range`` None = no
    clip_sample.  Returns (x_prev, x0|None)."""
    _dev(eps_in, x, noise, ratio)
    for t in (eps_in, x, noise):
        _f32(t, "latent tensors")
    if noise is not None and noise.shape != x.shape:
        raise HipExtensionError("ddpm_step: noise must have the sample's shape")
    B = x.shape[0]
    chw = x.shape[1:].numel()
    x_prev = torch.empty_like(x)
    x0 = torch.empty_like(x) if want_x0 else None
    sa, s1, c0, ct, ns, pa, p1 = (float(v) for v in coefs)
    check(lib().gmd_ddpm_step(_ptr(eps_in), _ptr(x), _ptr(noise), B, chw, int(do_cfg), float(guidance_scale), _ptr(ratio),
                              float(guidance_rescale), sa, s1, int(clip_range is not None), float(clip_range or 0.0), c0, ct, ns, pa, p1,
                              _ptr(x_prev), _ptr(x0), _stream()), "gmd_ddpm_step")
    return x_prev, x0


# ----------------------------------------------------------------------------------------------
# HDR tail
# ----------------------------------------------------------------------------------------------
def hdr_tail(sdr_dec, gm_dec, layout, B, H, W, qmax=99.0, eps=1 / 64, clamp=False,
             want=("sdr", "gm", "sdr_u8", "gm_u8", "hdr", "hdr_file", "hdr_u16")):
    """layout 0: [B,3,H,W]; 1: [B,H*W,3]; 2: [B,H*W,4].  Returns dict of [B,H,W,3] tensors."""
    _dev(sdr_dec, gm_dec)
    if sdr_dec.dtype != gm_dec.dtype:
        raise HipExtensionError("hdr_tail: dtype mismatch")
    dev = sdr_dec.device
    shp = (B, H, W, 3)
    kinds = {"sdr": torch.float32, "gm": torch.float32, "sdr_u8": torch.uint8, "gm_u8": torch.uint8,
             "hdr": torch.float32, "hdr_file": torch.float32, "hdr_u16": torch.uint16}
    out = {k: torch.empty(shp, dtype=kinds[k], device=dev) for k in want}
    g = lambda k: _ptr(out.get(k))
    tm, t0 = _timed("hdr_tail")
    check(lib().gmd_hdr_tail(_ptr(sdr_dec), _ptr(gm_dec), dtype_code(sdr_dec.dtype), layout, B, H, W, float(qmax), float(eps),
                             1 if clamp else 0, g("sdr"), g("gm"), g("sdr_u8"), g("gm_u8"), g("hdr"), g("hdr_file"),
                             g("hdr_u16"), _stream()), "gmd_hdr_tail")
    if tm:  # 2 x 3 float32 read per pixel + every requested output written once
        px = B * H * W
        tm.end("hdr_tail", 0.0, px * 2 * 3 * sdr_dec.element_size() + sum(v.numel() * v.element_size() for v in out.values()), t0)
    return out


def apply_gm_to_sdr(gm, sdr, qmax=9, eps=1 / 64, clamp=True):
    _dev(gm, sdr)
    gm, sdr = torch.broadcast_tensors(gm, sdr)
    gm, sdr = _f32(gm.contiguous(), "gm"), _f32(sdr.contiguous(), "sdr")
    out = torch.empty_like(sdr)
    check(lib().gmd_apply_gm_to_sdr(_ptr(gm), _ptr(sdr), _ptr(out), sdr.numel(), float(qmax), float(eps), int(clamp), _stream()),
          "gmd_apply_gm_to_sdr")
    return out


def tmo(x, kind, qmax=0.0, mu=500.0):
    _dev(x)
    _f32(x, "tmo input")
    out = torch.empty_like(x)
    check(lib().gmd_tmo(_ptr(x), _ptr(out), x.numel(), kind, float(qmax), float(mu), _stream()), "gmd_tmo")
    return out


def gamut_compress(x):
    _dev(x)
    _f32(x, "gamut_compress input")
    if x.dim() != 4 or x.shape[1] != 3:
        raise HipExtensionError("gamut_compress expects (B, 3, H, W)")
    out = torch.empty_like(x)
    check(lib().gmd_gamut_compress(_ptr(x), _ptr(out), x.shape[0], x.shape[2] * x.shape[3], _stream()), "gmd_gamut_compress")
    return out


def stage1_chain(gm, sdr, qmax):
    _dev(gm, sdr)
    _f32(gm, "gm")
    _f32(sdr, "sdr")
    out = torch.empty_like(sdr)
    check(lib().gmd_stage1_chain(_ptr(gm), _ptr(sdr), _ptr(out), sdr.shape[0], sdr.shape[2] * sdr.shape[3], float(qmax), _stream()),
          "gmd_stage1_chain")
    return out


def discretize_u16(x, codes=False):
    _dev(x)
    _f32(x, "discretize input")
    outf = torch.empty_like(x)
    outc = torch.empty(x.shape, dtype=torch.uint16, device=x.device) if codes else None
    check(lib().gmd_discretize_u16(_ptr(x), _ptr(outf), _ptr(outc), x.numel(), _stream()), "gmd_discretize_u16")
    return (outf, outc) if codes else outf


def quantize_u8(x):
    _dev(x)
    _f32(x, "quantize input")
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    check(lib().gmd_quantize_u8(_ptr(x), _ptr(out), x.numel(), _stream()), "gmd_quantize_u8")
    return out


def gemm_raw(a_ptr, w_ptr, c_ptr, dtype, out_dtype, M, N, K, lda, ldw, ldc, batch=1, sA=0, sW=0, sC=0,
             bias=None, alpha=1.0, act=ACT_NONE, exact=False):
    """Pointer-level gmd_gemm_nt for strided sub-blocks (per-head attention products of the parity path).
    a_ptr/w_ptr/c_ptr are integer device addresses; the caller keeps the owning tensors alive."""
    _dev(bias)
    code = GMD_F32S if (dtype == torch.float32 and f32_mode() == "split" and K % 32 == 0 and not exact) else dtype_code(dtype)
    check(lib().gmd_gemm_nt(a_ptr, w_ptr, c_ptr, code, dtype_code(out_dtype), M, N, K, lda, ldw, ldc, batch,
                            sA, sW, sC, _ptr(_f32(bias, "bias")), None, 0, 0, None, 0, 0, float(alpha), act, None, 0, None, 0, _stream()),
          "gmd_gemm_nt")


def rgbe_encode(rgb):
    """float32 [..., 3] non-negative RGB -> uint8 [..., 4] Radiance RGBE pixels."""
    _dev(rgb)
    _f32(rgb, "rgbe input")
    if rgb.shape[-1] != 3:
        raise HipExtensionError("rgbe_encode expects [..., 3]")
    out = torch.empty(rgb.shape[:-1] + (4,), dtype=torch.uint8, device=rgb.device)
    check(lib().gmd_rgbe_encode(_ptr(rgb), _ptr(out), rgb.numel() // 3, _stream()), "gmd_rgbe_encode")
    return out

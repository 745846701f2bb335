"""
ctypes binding of libgmd_hip.so (C ABI declared in include/gmd_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` /
``make -C gm-diffusion_amd/csrc``.  There is NO fallback: if the shared library is
missing or fails to load, :func:`lib` raises ``HipExtensionError`` and every
compute entry point of the package fails loudly.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GMD_LIB_OVERRIDE") or os.path.join(_HERE, "libgmd_hip.so")  # override: kernel-debug builds only
ABI_VERSION = 11

GMD_F32, GMD_BF16, GMD_F16, GMD_F32S, GMD_F32SW, GMD_F32SA = 0, 1, 2, 3, 4, 5
ACT_NONE, ACT_SILU, ACT_GEGLU, ACT_QUICK_GELU = 0, 1, 2, 3

P, I, L, F = c_void_p, c_int, c_int64, c_float

# name -> argtypes; every function returns int (status) unless listed in _RESTYPES
SIGNATURES = {
    "gmd_abi_version": [],
    "gmd_last_error": [],
    "gmd_hdr_tail": [P, P, I, I, I, I, I, F, F, I, P, P, P, P, P, P, P, P],
    "gmd_apply_gm_to_sdr": [P, P, P, L, F, F, I, P],
    "gmd_tmo": [P, P, L, I, F, F, P],
    "gmd_gamut_compress": [P, P, I, L, P],
    "gmd_stage1_chain": [P, P, P, I, L, F, P],
    "gmd_discretize_u16": [P, P, P, L, P],
    "gmd_quantize_u8": [P, P, L, P],
    "gmd_rgbe_encode": [P, P, L, P],
    "gmd_rgbe_rle_bound": [I, I],
    "gmd_rgbe_rle_encode": [P, I, I, P, L, P],
    "gmd_latent_step": [P, P, P, P, P, P, I, L, I, F, P, F, I, F, F, F, F, F, P, P, P, P],
    "gmd_dpm_step": [P, P, P, I, L, I, F, P, F, I, F, F, F, F, F, F, F, F, P, P, P, P],
    "gmd_ddpm_step": [P, P, P, I, L, I, F, P, F, F, F, I, F, F, F, F, F, F, P, P, P],
    "gmd_cfg_std_ratio": [P, I, L, F, P, P],
    "gmd_pack_unet_input": [P, I, P, I, I, L, I, P, I, I, P],
    "gmd_unpack_nchw": [P, I, L, I, I, L, P, P],
    "gmd_gemm_plan_override": [I, I, I, I],
    "gmd_gemm_plan_family": [I],
    "gmd_splitk_fixup_max": [I],
    "gmd_gemm_nt": [P, P, P, I, I, I, I, I, L, L, L, I, L, L, L, P, P, I, L, P, L, L, F, I, P, I, P, L, P],
    "gmd_gemm_colstats_plan": [I, I, I, I, I, L, I],
    "gmd_gemm_plan_info": [I, I, I, I, I, L, I, P],
    "gmd_gemm_out_split_ok": [I, I, I, I, L],
    "gmd_gemm_qkv_vt_ok": [I, I, I, I, I, I, L],
    "gmd_gemm_qkv_vt": [P, P, P, P, I, I, I, I, L, I, I, L, F, P, L, P],
    "gmd_conv_patch_override": [I],
    "gmd_stamp": [P, P, I, I, P],
    "gmd_split_weights": [P, P, L, L, L, P],
    "gmd_ff_geglu_fused_supported": [I, L, I],
    "gmd_ff_geglu_fused": [P, P, P, P, P, P, P, I, L, I, P],
    "gmd_conv3x3": [P, P, P, I, I, I, I, I, I, I, I, I, I, P, P, L, P, F, P, I, P, L, P],
    "gmd_conv3x3_gn_fusable": [I, I, I, I, I, I, I, I, I, I, L],
    "gmd_conv3x3_groupnorm": [P, P, P, P, I, I, I, I, I, I, I, I, I, P, P, L, P, F, I, F, P, P, I, P, L, P],
    "gmd_attention": [P, P, P, P, I, I, I, I, I, I, L, L, L, L, L, L, L, L, F, I, P],
    "gmd_softmax_rows": [P, L, P, I, L, L, I, F, I, P],
    "gmd_groupnorm_nsplit": [L],
    "gmd_groupnorm_stats": [P, I, I, L, I, I, F, P, P, P, P, P],
    "gmd_groupnorm_apply": [P, P, I, I, L, I, P, I, P],
    "gmd_groupnorm_colstats": [P, P, I, I, L, I, I, F, P, P, P, I, P, I, I, P],
    "gmd_groupnorm_split": [P, P, I, I, L, I, I, F, P, P, P, I, P],
    "gmd_groupnorm_fused": [P, P, I, I, L, I, I, F, P, P, I, P],
    "gmd_layernorm": [P, P, I, L, I, P, P, F, P],
    "gmd_geglu": [P, P, I, L, I, P],
    "gmd_timestep_embedding": [P, P, I, I, I, I, F, P],
    "gmd_concat_channels": [P, I, P, I, P, I, L, P],
    "gmd_embedding_lookup": [P, P, P, P, I, L, I, I, I, P],
    "gmd_cast": [P, I, P, I, L, P],
    "gmd_dup_batch": [P, P, L, P],
}
_RESTYPES = {"gmd_last_error": c_char_p, "gmd_rgbe_rle_bound": c_int64}


class HipExtensionError(RuntimeError):
    """libgmd_hip.so is missing/unloadable, or a kernel call returned an error."""


_lib = None


def lib():
    """Load (once) and return the ctypes handle; raises HipExtensionError when unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipExtensionError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C gm-diffusion_amd/csrc`. gm_diffusion (MI355X build) has no CPU/eager fallback."
        )
    try:
        handle = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover - depends on the ROCm runtime
        raise HipExtensionError(f"cannot load {LIB_PATH}: {e}") from e
    for name, args in SIGNATURES.items():
        fn = getattr(handle, name)  # AttributeError here = ABI mismatch, let it propagate loudly
        fn.argtypes = args
        fn.restype = _RESTYPES.get(name, c_int)
    got = handle.gmd_abi_version()
    if got != ABI_VERSION:
        raise HipExtensionError(f"libgmd_hip.so ABI {got} != expected {ABI_VERSION}; rebuild")
    _lib = handle
    return handle


def check(rc, what=""):
    if rc != 0:
        msg = lib().gmd_last_error()
        raise HipExtensionError(f"{what}: gmd error {rc}: {msg.decode() if msg else '?'}")

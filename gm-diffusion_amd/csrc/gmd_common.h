// Shared device/host helpers for libgmd_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/gmd_hip.h"
#include "wg_trace.h"  // diagnostic build only; expands to nothing in the product

typedef unsigned short bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;  // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

void gmd_set_error(const char* fmt, ...);

#define GMD_REQUIRE(cond, ...)                    \
    do {                                          \
        if (!(cond)) {                            \
            gmd_set_error(__VA_ARGS__);           \
            return GMD_ERR_INVALID;               \
        }                                         \
    } while (0)

#define GMD_CHECK_LAUNCH(name)                                                     \
    do {                                                                           \
        hipError_t e_ = hipGetLastError();                                         \
        if (e_ != hipSuccess) {                                                    \
            gmd_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));   \
            return GMD_ERR_LAUNCH;                                                 \
        }                                                                          \
    } while (0)

__device__ __forceinline__ bf16x8 as_frag(uint4 v) { return __builtin_bit_cast(bf16x8, v); }

static inline bool gmd_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
// round-to-nearest-even in hardware (v_cvt_pk_bf16_f32 on gfx950); NaN stays NaN
typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    const bf16x2_hw v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, v);
}

// IEEE half: the second 16-bit element type of the MFMA path.  The reference's own half-precision runs are float16
// (scripts/stage2/experiments/batch_size_sweep.py: `.to(device, dtype=torch.float16)`); with 11 significand bits against
// bfloat16's 8 the latent drift over a PNDM trajectory is ~8x smaller at the same matrix-core rate.
typedef _Float16 f16_t;  // a distinct C++ type from bf16_t (= unsigned short), so overloads and Elem<> can tell them apart
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int kVec = 4;  // elements per 16 bytes
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static constexpr int kVec = 8;
    __device__ static __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

template <> struct Elem<f16_t> {
    static constexpr int kVec = 8;
    __device__ static __forceinline__ float ld(const f16_t* p) { return (float)*p; }
    __device__ static __forceinline__ void st(f16_t* p, float v) { *p = (f16_t)v; }
};

// Policy of a 16-bit element type on the matrix-core path: fragment type, MFMA forms, pair pack / unpack, the bit pattern
// of 1.0 and rounding of a float to the element's precision.
template <typename HT> struct Half;
template <> struct Half<bf16_t> {
    static constexpr unsigned kOne = 0x3F80u;
    __device__ static __forceinline__ unsigned pack2(float lo, float hi) { return pack_bf16x2(lo, hi); }
    __device__ static __forceinline__ void unpack2(unsigned w, float& lo, float& hi) { lo = __uint_as_float(w << 16); hi = __uint_as_float(w & 0xffff0000u); }
    __device__ static __forceinline__ float round(float x) { return (float)(__bf16)x; }
    __device__ static __forceinline__ f32x4 mfma16(uint4 a, uint4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    __device__ static __forceinline__ f32x16 mfma32(uint4 a, uint4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct Half<f16_t> {
    static constexpr unsigned kOne = 0x3C00u;
    // Not saturating: a value beyond 65504 becomes infinite.  The callers stay inside the range by construction: attention
    // stores O, a convex combination of V's (finite float16) values, and carries P <= 2^14 (the lagged-stabiliser window of
    // the float16 kernels); GEMM / conv outputs beyond the range are the documented limit of the float16 path.
    __device__ static __forceinline__ unsigned pack2(float lo, float hi) {
        const f16x2 v = {(_Float16)lo, (_Float16)hi};  // round-to-nearest-even conversions
        return __builtin_bit_cast(unsigned, v);
    }
    __device__ static __forceinline__ void unpack2(unsigned w, float& lo, float& hi) {
        const f16x2 v = __builtin_bit_cast(f16x2, w);
        lo = (float)v[0]; hi = (float)v[1];
    }
    __device__ static __forceinline__ float round(float x) { return (float)(_Float16)x; }
    __device__ static __forceinline__ f32x4 mfma16(uint4 a, uint4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    __device__ static __forceinline__ f32x16 mfma32(uint4 a, uint4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

// 16-byte vector <-> 8 (bf16 / f16) or 4 (f32) floats
__device__ __forceinline__ void load_vec(const bf16_t* p, float (&v)[8]) {
    uint4 r = *reinterpret_cast<const uint4*>(p);
    unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(w[i] << 16);
        v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ void store_vec(bf16_t* p, const float (&v)[8]) {
    uint4 r;
    r.x = pack_bf16x2(v[0], v[1]);
    r.y = pack_bf16x2(v[2], v[3]);
    r.z = pack_bf16x2(v[4], v[5]);
    r.w = pack_bf16x2(v[6], v[7]);
    *reinterpret_cast<uint4*>(p) = r;
}
__device__ __forceinline__ void load_vec(const f16_t* p, float (&v)[8]) {
    const uint4 r = *reinterpret_cast<const uint4*>(p);
    const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) Half<f16_t>::unpack2(w[i], v[2 * i], v[2 * i + 1]);
}
__device__ __forceinline__ void store_vec(f16_t* p, const float (&v)[8]) {
    uint4 r;
    r.x = Half<f16_t>::pack2(v[0], v[1]);
    r.y = Half<f16_t>::pack2(v[2], v[3]);
    r.z = Half<f16_t>::pack2(v[4], v[5]);
    r.w = Half<f16_t>::pack2(v[6], v[7]);
    *reinterpret_cast<uint4*>(p) = r;
}
__device__ __forceinline__ void load_vec(const float* p, float (&v)[4]) {
    float4 r = *reinterpret_cast<const float4*>(p);
    v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
}
__device__ __forceinline__ void store_vec(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}

// host side: dtype codes -> element types
// GroupNorm straight from split-K partial slabs (norm.hip, used by gmd_conv3x3_groupnorm in gemm.hip).  `dtype` is the ACTIVATION
// type (GMD_BF16 / GMD_F16 / GMD_F32).  _ok: the (sample, group) slice fits the register-resident kernel with 16-byte accesses.
bool gmd_gn_from_slabs_ok(int dtype, int B, int64_t HW, int C, int G);
int gmd_launch_gn_from_slabs(const float* ws, int ksplit, float alpha, const float* bias, const float* rowbias, int64_t ldrb,
                             const void* residual, void* Yraw, void* Ynorm, int dtype, int B, int64_t HW, int C, int G, float eps,
                             const float* gamma, const float* beta, int silu, hipStream_t stream);

static inline bool gmd_is_half(int dtype) { return dtype == GMD_BF16 || dtype == GMD_F16; }
static inline bool gmd_known_dtype(int dtype) { return dtype == GMD_F32 || gmd_is_half(dtype); }
// float32 tensors contracted on the matrix cores as three float16 passes (gemm_split.hip): both operands plain / W pre-split /
// both pre-split (GMD_F32SA, round 4: producers store activations as [hi 64 B | lo 64 B] per 32 elements)
static inline bool gmd_is_split(int dtype) { return dtype == GMD_F32S || dtype == GMD_F32SW || dtype == GMD_F32SA; }
// run f(T{}) with T = float / bf16_t / f16_t (the caller has validated the code)
template <typename F>
static inline void gmd_for_dtype(int dtype, F&& f) {
    if (dtype == GMD_BF16) f(bf16_t{});
    else if (dtype == GMD_F16) f(f16_t{});
    else f(float{});
}

// x * sigmoid(x).  The reciprocal is the hardware's (v_rcp_f32, <= 1 ulp) instead of an IEEE division (v_div_scale x2, v_rcp, four
// FMAs, v_div_fmas, v_div_fixup: 12 instructions per element): GroupNorm + SiLU runs this on every element of every resnet input.
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// ---- float32 -> (hi, lo) float16 pairs: x = hi + lo, hi = f16(x), lo = f16(x - hi) (gemm_split.hip) ----
typedef __attribute__((ext_vector_type(2))) _Float16 gmd_f16x2_t;
// (a, b) -> packed f16 pair hi and packed f16 pair lo
__device__ __forceinline__ void gmd_split2(float a, float b, unsigned& hi, unsigned& lo) {
    const gmd_f16x2_t h = {(_Float16)a, (_Float16)b};  // v_cvt_pk_f16_f32, round to nearest even
    const unsigned hw = __builtin_bit_cast(unsigned, h);
    float ra, rb;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(hw), "v"(a));                  // a - f32(h.lo)
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(hw), "v"(b));  // b - f32(h.hi)
    const gmd_f16x2_t l = {(_Float16)ra, (_Float16)rb};
    hi = hw;
    lo = __builtin_bit_cast(unsigned, l);
}
// Pre-split storage of a float32 tensor whose rows are multiples of 32 elements (GMD_F32SA activations, the layout of
// gmd_split_weights): every 32-element chunk (128 bytes, the same bytes as the float32 values) holds [hi 64 B | lo 64 B]; inside each
// half, 16-byte piece q (0..3) holds elements {4q .. 4q+3, 16+4q .. 16+4q+3} -- the k's lane group q of a 16x16x32 MFMA consumes.
// Stores elements e .. e+3 (e % 4 == 0; element index over the whole tensor) of the tensor at Y.
__device__ __forceinline__ void gmd_store_split4(float* Y, int64_t e, float v0, float v1, float v2, float v3) {
    unsigned char* cb = reinterpret_cast<unsigned char*>(Y + (e & ~(int64_t)31));
    const int ci = (int)(e & 31), off = ((ci & 15) >> 2) * 16 + (ci >> 4) * 8;
    unsigned h0, l0, h1, l1;
    gmd_split2(v0, v1, h0, l0);
    gmd_split2(v2, v3, h1, l1);
    *reinterpret_cast<uint2*>(cb + off) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(cb + 64 + off) = make_uint2(l0, l1);
}

// Wave-wide reductions on the data-parallel-primitive (DPP) path: quad permutes, then the two row mirrors, then the four
// 16-lane rows are met through v_readlane -- ~11 short VALU / SALU instructions instead of six ds_bpermute round trips
// through the LDS crossbar (__shfl_xor), whose dependent latency dominated the one-row-per-wave LayerNorm.  Fixed order:
// deterministic; every lane returns the same value.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// sum over each aligned group of 8 lanes (every lane of the group returns it), doubles, on the DPP path
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double group8_sum(double v) {
    v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);  // row_half_mirror
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f32<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_f32<0x141>(v);  // row_half_mirror: the other quad of each group of 8
    v += dpp_f32<0x140>(v);  // row_mirror: the other half of each row of 16
    const int iv = __float_as_int(v);
    return (__int_as_float(__builtin_amdgcn_readlane(iv, 0)) + __int_as_float(__builtin_amdgcn_readlane(iv, 16))) +
           (__int_as_float(__builtin_amdgcn_readlane(iv, 32)) + __int_as_float(__builtin_amdgcn_readlane(iv, 48)));
}
// value of lane l combined with lane l ^ 32 (the two halves of a wave) through v_permlane32_swap: no LDS crossbar trip
__device__ __forceinline__ float half_swap_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_swap_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f32<0xB1>(v));
    v = fmaxf(v, dpp_f32<0x4E>(v));
    v = fmaxf(v, dpp_f32<0x141>(v));
    v = fmaxf(v, dpp_f32<0x140>(v));
    const int iv = __float_as_int(v);
    return fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(iv, 0)), __int_as_float(__builtin_amdgcn_readlane(iv, 16))),
                 fmaxf(__int_as_float(__builtin_amdgcn_readlane(iv, 32)), __int_as_float(__builtin_amdgcn_readlane(iv, 48))));
}

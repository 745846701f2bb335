// HDR tail kernels: decode post-process, gain-map recomposition (Eq. 1), tone-mapping
// operators, gamut compression and the integer quantisers.  HBM-bound elementwise work, no LDS.
// The pipeline's own input (the VAE decoder's float32 [B,HW,4] image) takes hdr_tail_vec4_kernel:
// four pixels per thread, 16-byte loads, 16- / 8- / 4-byte stores of whole 12-element runs; the
// other layouts (planar NCHW, packed RGB, bf16) take the generic one-pixel-per-thread kernel.
//
// Float arithmetic follows the reference's operation order with FMA contraction disabled
// (this file is built with -ffp-contract=off) so results differ from torch-CPU only in the
// transcendental functions (powf / log1pf, <= ~1 ulp).
#include "gmd_common.h"
#include <math.h>

#pragma clang fp contract(off)

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }

// tone_mapping.py:68-71 / formal_improved.py:40-45
__device__ __forceinline__ float eq1(float sdr, float gm, float qmax, float eps, bool clamp_out) {
    float lin = powf(clamp01(sdr), 2.2f);
    float hdr = (lin + eps) * (1.0f + gm * qmax) - eps;
    if (clamp_out) hdr = fminf(fmaxf(hdr, 0.0f), qmax + 1.0f);
    return hdr;
}

// tone_mapping.py:33-36 with explicit mu (log1p(mu) evaluated in double on the host, like math.log1p)
__device__ __forceinline__ float mulog(float hdr, float qmax_p1, float mu, float inv_denominator_is_div) {
    float x = hdr / qmax_p1;
    float tm = log1pf(mu * x) / inv_denominator_is_div;
    return clamp01(tm);
}

__device__ __forceinline__ uint8_t u8_trunc(float x01) { return (uint8_t)(int)(x01 * 255.0f); }
__device__ __forceinline__ float u16_code(float x) { return rintf(fminf(fmaxf(x * 65535.0f, 0.0f), 65535.0f)); }

// in_layout: 0 = planar [B,3,HW]; 1 = interleaved [B,HW,3]; 2 = interleaved [B,HW,4] (4th channel ignored)
template <typename T>
__device__ __forceinline__ void load_px(const T* base, int layout, int64_t b, int64_t p, int64_t HW, float (&v)[3]) {
    if (layout == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = Elem<T>::ld(base + (b * 3 + c) * HW + p);
    } else {
        const int ld = layout == 1 ? 3 : 4;
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = Elem<T>::ld(base + (b * HW + p) * ld + c);
    }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void hdr_tail_kernel(
    const T* __restrict__ sdr_dec, const T* __restrict__ gm_dec, int layout, int B, int64_t HW,
    float qmax, float eps, int flags, float* __restrict__ sdr_img, float* __restrict__ gm_img,
    uint8_t* __restrict__ sdr_u8, uint8_t* __restrict__ gm_u8, float* __restrict__ hdr,
    float* __restrict__ hdr_file, uint16_t* __restrict__ hdr_u16) {
    const int64_t total = (int64_t)B * HW;
    const float qp1 = qmax + 1.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / HW, p = i - b * HW;
        float s[3], g[3];
        load_px(sdr_dec, layout, b, p, HW, s);
        load_px(gm_dec, layout, b, p, HW, g);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float sv = clamp01(s[c] / 2.0f + 0.5f);  // generate_hdr.py:227
            const float gv = clamp01(g[c] / 2.0f + 0.5f);  // generate_hdr.py:232
            const int64_t o = i * 3 + c;
            if (sdr_img) sdr_img[o] = sv;
            if (gm_img) gm_img[o] = gv;
            if (sdr_u8) sdr_u8[o] = u8_trunc(sv);
            if (gm_u8) gm_u8[o] = u8_trunc(gv);
            const float h = eq1(sv, gv, qmax, eps, flags & 1);
            if (hdr) hdr[o] = h;
            const float hf = h / qp1;  // generate_hdr.py:28
            if (hdr_file) hdr_file[o] = hf;
            if (hdr_u16) hdr_u16[o] = (uint16_t)u16_code(hf);
        }
    }
}

// Fast path of the fused tail for float32 [B,HW,4] input (what AutoencoderKL.decode_nhwc produces): FOUR pixels per thread.
// Loads are 16 bytes per pixel and operand; the 12 consecutive output elements of a thread are stored as three float4
// (float images), three uint32 (u8 bytes) or three uint2 (u16 codes) -- whole aligned runs instead of stride-3 scalars.
// Element arithmetic is the generic kernel's, expression for expression: both paths are bit-identical.
__global__ __launch_bounds__(kThreads) void hdr_tail_vec4_kernel(
    const float4* __restrict__ sdr_dec, const float4* __restrict__ gm_dec, int64_t npix4, float qmax, float eps, int flags,
    float* __restrict__ sdr_img, float* __restrict__ gm_img, uint8_t* __restrict__ sdr_u8, uint8_t* __restrict__ gm_u8,
    float* __restrict__ hdr, float* __restrict__ hdr_file, uint16_t* __restrict__ hdr_u16) {
    const float qp1 = qmax + 1.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix4; i += (int64_t)gridDim.x * blockDim.x) {
        float sv[12], gv[12], hv[12], hf[12];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 a = sdr_dec[i * 4 + k], b = gm_dec[i * 4 + k];
            const float s3[3] = {a.x, a.y, a.z}, g3[3] = {b.x, b.y, b.z};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int e = 3 * k + c;
                sv[e] = clamp01(s3[c] / 2.0f + 0.5f);  // generate_hdr.py:227
                gv[e] = clamp01(g3[c] / 2.0f + 0.5f);  // generate_hdr.py:232
                hv[e] = eq1(sv[e], gv[e], qmax, eps, flags & 1);
                hf[e] = hv[e] / qp1;                   // generate_hdr.py:28
            }
        }
        const int64_t o = i * 12;
        auto st_f32 = [&](float* dst, const float (&v)[12]) {
#pragma unroll
            for (int k = 0; k < 3; ++k) *reinterpret_cast<float4*>(dst + o + 4 * k) = make_float4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
        };
        auto st_u8 = [&](uint8_t* dst, const float (&v)[12]) {
#pragma unroll
            for (int k = 0; k < 3; ++k)
                *reinterpret_cast<uint32_t*>(dst + o + 4 * k) = (uint32_t)u8_trunc(v[4 * k]) | ((uint32_t)u8_trunc(v[4 * k + 1]) << 8) |
                                                                ((uint32_t)u8_trunc(v[4 * k + 2]) << 16) | ((uint32_t)u8_trunc(v[4 * k + 3]) << 24);
        };
        if (sdr_img) st_f32(sdr_img, sv);
        if (gm_img) st_f32(gm_img, gv);
        if (sdr_u8) st_u8(sdr_u8, sv);
        if (gm_u8) st_u8(gm_u8, gv);
        if (hdr) st_f32(hdr, hv);
        if (hdr_file) st_f32(hdr_file, hf);
        if (hdr_u16) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const uint32_t c0 = (uint32_t)(uint16_t)u16_code(hf[4 * k]), c1 = (uint32_t)(uint16_t)u16_code(hf[4 * k + 1]);
                const uint32_t c2 = (uint32_t)(uint16_t)u16_code(hf[4 * k + 2]), c3 = (uint32_t)(uint16_t)u16_code(hf[4 * k + 3]);
                *reinterpret_cast<uint2*>(hdr_u16 + o + 4 * k) = make_uint2(c0 | (c1 << 16), c2 | (c3 << 16));
            }
        }
    }
}

__global__ __launch_bounds__(kThreads) void apply_gm_kernel(const float* __restrict__ gm, const float* __restrict__ sdr,
                                                            float* __restrict__ out, int64_t n, float qmax, float eps, int clamp_out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = eq1(sdr[i], gm[i], qmax, eps, clamp_out);
}

__global__ __launch_bounds__(kThreads) void tmo_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n,
                                                       int kind, float qmax, float mu, float log1p_mu) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = in[i];
        float r;
        if (kind == 0) r = x / (qmax + 1.0f);                   // linear_scale_tmo
        else if (kind == 1) r = clamp01(x);                       // hard_clip_tmo
        else if (kind == 2) r = mulog(x, qmax + 1.0f, mu, log1p_mu);  // fix_mulog / random_tmo
        else if (kind == 3) {                                     // tmo_cuda
            const float c = clamp01(x / 10.0f);
            r = log1pf(mu * c) / log1p_mu;
        } else if (kind == 4) r = clamp01(x / 2.0f + 0.5f);       // decode post-process (gm.py:606)
        else if (kind == 5) r = x * mu;                           // 1/scaling_factor * latents (generate_hdr.py:225)
        else r = x / mu;                                          // latents / scaling_factor (gm.py:1094)
        out[i] = r;
    }
}

// tone_mapping.py:78-90: out[c] = sum_k in[k] * M[c][k], products accumulated in k order
__device__ __forceinline__ void gamut(const float (&in)[3], float (&out)[3]) {
    const float M[3][3] = {{1.660491f, -0.587641f, -0.072850f},
                           {-0.124550f, 1.132900f, -0.008349f},
                           {-0.018151f, -0.100579f, 1.118730f}};
#pragma unroll
    for (int c = 0; c < 3; ++c) out[c] = clamp01(in[0] * M[c][0] + in[1] * M[c][1] + in[2] * M[c][2]);
}

__global__ __launch_bounds__(kThreads) void gamut_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int64_t HW) {
    const int64_t total = (int64_t)B * HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / HW, p = i - b * HW;
        float v[3], r[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = in[(b * 3 + c) * HW + p];
        gamut(v, r);
#pragma unroll
        for (int c = 0; c < 3; ++c) out[(b * 3 + c) * HW + p] = r[c];
    }
}

__global__ __launch_bounds__(kThreads) void stage1_chain_kernel(const float* __restrict__ gm, const float* __restrict__ sdr,
                                                                float* __restrict__ out, int B, int64_t HW, float qmax, float log1p_mu) {
    const int64_t total = (int64_t)B * HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / HW, p = i - b * HW;
        float v[3], r[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int64_t o = (b * 3 + c) * HW + p;
            v[c] = mulog(eq1(sdr[o], gm[o], qmax, 1.0f / 64.0f, true), qmax + 1.0f, 500.0f, log1p_mu);
        }
        gamut(v, r);
#pragma unroll
        for (int c = 0; c < 3; ++c) out[(b * 3 + c) * HW + p] = r[c];
    }
}

__global__ __launch_bounds__(kThreads) void u16_kernel(const float* __restrict__ in, float* __restrict__ outf,
                                                       uint16_t* __restrict__ outc, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float q = u16_code(in[i]);
        if (outf) outf[i] = q / 65535.0f;
        if (outc) outc[i] = (uint16_t)q;
    }
}

__global__ __launch_bounds__(kThreads) void u8_kernel(const float* __restrict__ in, uint8_t* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = u8_trunc(in[i]);
}

// Radiance RGBE pixel (Ward's float2rgbe, the encoder behind cv2.imwrite("*.hdr") that the reference calls at
// scripts/inference/generate_hdr.py:27-30): v = max(r,g,b); v < 1e-32 -> 0,0,0,0; else v = frexp(v,&e) * 256 / v;
// bytes = (uint8)(c * v) (truncation), exponent byte = e + 128.
__global__ __launch_bounds__(kThreads) void rgbe_kernel(const float* __restrict__ rgb, uint8_t* __restrict__ out, int64_t npix) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        // RGBE has no sign: negative components (the unclamped Eq. 1 can produce down to -1/64) are stored as 0
        const float r = fmaxf(rgb[i * 3], 0.f), g = fmaxf(rgb[i * 3 + 1], 0.f), b = fmaxf(rgb[i * 3 + 2], 0.f);
        float v = fmaxf(r, fmaxf(g, b));
        uchar4 px = make_uchar4(0, 0, 0, 0);
        if (v >= 1e-32f) {
            int e;
            const float m = frexpf(v, &e);
            const float sc = m * 256.0f / v;
            px = make_uchar4((uint8_t)(int)(r * sc), (uint8_t)(int)(g * sc), (uint8_t)(int)(b * sc), (uint8_t)(e + 128));
        }
        *reinterpret_cast<uchar4*>(out + i * 4) = px;
    }
}

inline int grid_for(int64_t n) {
    int64_t g = (n + kThreads - 1) / kThreads;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" {

int gmd_hdr_tail(const void* sdr_dec, const void* gm_dec, int in_dtype, int in_layout, int B, int H, int W,
                 float qmax, float eps, int flags, float* sdr_img, float* gm_img, uint8_t* sdr_u8,
                 uint8_t* gm_u8, float* hdr, float* hdr_file, uint16_t* hdr_u16, gmd_stream_t stream) {
    GMD_REQUIRE(sdr_dec && gm_dec, "gmd_hdr_tail: null input");
    GMD_REQUIRE(B >= 0 && H >= 0 && W >= 0, "gmd_hdr_tail: negative shape");
    GMD_REQUIRE(in_layout >= 0 && in_layout <= 2, "gmd_hdr_tail: in_layout must be 0 (NCHW), 1 (NHWC3) or 2 (NHWC4)");
    GMD_REQUIRE(gmd_known_dtype(in_dtype), "gmd_hdr_tail: bad dtype %d", in_dtype);
    const int64_t HW = (int64_t)H * W;
    if ((int64_t)B * HW == 0) return GMD_OK;
    hipStream_t s = (hipStream_t)stream;
    const int grid = grid_for((int64_t)B * HW);
    const auto al = [](const void* p, int a) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) % a) == 0; };
    const bool vec4 = in_dtype == GMD_F32 && in_layout == 2 && ((int64_t)B * HW) % 4 == 0 && al(sdr_dec, 16) && al(gm_dec, 16) &&
                      al(sdr_img, 16) && al(gm_img, 16) && al(hdr, 16) && al(hdr_file, 16) && al(sdr_u8, 4) && al(gm_u8, 4) && al(hdr_u16, 8);
    if (vec4)
        hdr_tail_vec4_kernel<<<grid_for((int64_t)B * HW / 4), kThreads, 0, s>>>((const float4*)sdr_dec, (const float4*)gm_dec, (int64_t)B * HW / 4, qmax,
                                                                                eps, flags, sdr_img, gm_img, sdr_u8, gm_u8, hdr, hdr_file, hdr_u16);
    else
        gmd_for_dtype(in_dtype, [&](auto tag) {
            using T = decltype(tag);
            hdr_tail_kernel<T><<<grid, kThreads, 0, s>>>((const T*)sdr_dec, (const T*)gm_dec, in_layout, B, HW, qmax, eps, flags, sdr_img, gm_img, sdr_u8,
                                                         gm_u8, hdr, hdr_file, hdr_u16);
        });
    GMD_CHECK_LAUNCH("gmd_hdr_tail");
    return GMD_OK;
}

int gmd_apply_gm_to_sdr(const float* gm, const float* sdr, float* out, int64_t n, float qmax, float eps, int clamp,
                        gmd_stream_t stream) {
    GMD_REQUIRE(n >= 0, "gmd_apply_gm_to_sdr: negative n");
    if (n == 0) return GMD_OK;
    GMD_REQUIRE(gm && sdr && out, "gmd_apply_gm_to_sdr: null pointer");
    apply_gm_kernel<<<grid_for(n), kThreads, 0, (hipStream_t)stream>>>(gm, sdr, out, n, qmax, eps, clamp);
    GMD_CHECK_LAUNCH("gmd_apply_gm_to_sdr");
    return GMD_OK;
}

int gmd_tmo(const float* in, float* out, int64_t n, int kind, float qmax, float mu, gmd_stream_t stream) {
    GMD_REQUIRE(n >= 0, "gmd_tmo: negative n");
    GMD_REQUIRE(kind >= 0 && kind <= 6, "gmd_tmo: kind %d not in 0..6", kind);
    if (n == 0) return GMD_OK;
    GMD_REQUIRE(in && out, "gmd_tmo: null pointer");
    if (kind == 3) mu = 5000.0f;
    const float l = kind >= 4 ? 1.0f : (float)log1p((double)mu);  // python: math.log1p(mu) (double) then float32 division
    tmo_kernel<<<grid_for(n), kThreads, 0, (hipStream_t)stream>>>(in, out, n, kind, qmax, mu, l);
    GMD_CHECK_LAUNCH("gmd_tmo");
    return GMD_OK;
}

int gmd_gamut_compress(const float* in, float* out, int B, int64_t HW, gmd_stream_t stream) {
    GMD_REQUIRE(B >= 0 && HW >= 0, "gmd_gamut_compress: negative shape");
    if ((int64_t)B * HW == 0) return GMD_OK;
    GMD_REQUIRE(in && out, "gmd_gamut_compress: null pointer");
    gamut_kernel<<<grid_for((int64_t)B * HW), kThreads, 0, (hipStream_t)stream>>>(in, out, B, HW);
    GMD_CHECK_LAUNCH("gmd_gamut_compress");
    return GMD_OK;
}

int gmd_stage1_chain(const float* gm, const float* sdr, float* out, int B, int64_t HW, float qmax, gmd_stream_t stream) {
    GMD_REQUIRE(B >= 0 && HW >= 0, "gmd_stage1_chain: negative shape");
    if ((int64_t)B * HW == 0) return GMD_OK;
    GMD_REQUIRE(gm && sdr && out, "gmd_stage1_chain: null pointer");
    stage1_chain_kernel<<<grid_for((int64_t)B * HW), kThreads, 0, (hipStream_t)stream>>>(gm, sdr, out, B, HW, qmax,
                                                                                      (float)log1p(500.0));
    GMD_CHECK_LAUNCH("gmd_stage1_chain");
    return GMD_OK;
}

int gmd_discretize_u16(const float* in, float* out_float, uint16_t* out_codes, int64_t n, gmd_stream_t stream) {
    GMD_REQUIRE(n >= 0, "gmd_discretize_u16: negative n");
    if (n == 0) return GMD_OK;
    GMD_REQUIRE(in && (out_float || out_codes), "gmd_discretize_u16: null pointer");
    u16_kernel<<<grid_for(n), kThreads, 0, (hipStream_t)stream>>>(in, out_float, out_codes, n);
    GMD_CHECK_LAUNCH("gmd_discretize_u16");
    return GMD_OK;
}

int gmd_rgbe_encode(const float* rgb, uint8_t* out, int64_t npix, gmd_stream_t stream) {
    GMD_REQUIRE(npix >= 0, "gmd_rgbe_encode: negative size");
    if (npix == 0) return GMD_OK;
    GMD_REQUIRE(rgb && out && (reinterpret_cast<uintptr_t>(out) & 3) == 0, "gmd_rgbe_encode: null or unaligned pointer");
    rgbe_kernel<<<grid_for(npix), kThreads, 0, (hipStream_t)stream>>>(rgb, out, npix);
    GMD_CHECK_LAUNCH("gmd_rgbe_encode");
    return GMD_OK;
}

int gmd_quantize_u8(const float* in, uint8_t* out, int64_t n, gmd_stream_t stream) {
    GMD_REQUIRE(n >= 0, "gmd_quantize_u8: negative n");
    if (n == 0) return GMD_OK;
    GMD_REQUIRE(in && out, "gmd_quantize_u8: null pointer");
    u8_kernel<<<grid_for(n), kThreads, 0, (hipStream_t)stream>>>(in, out, n);
    GMD_CHECK_LAUNCH("gmd_quantize_u8");
    return GMD_OK;
}

}  // extern "C"

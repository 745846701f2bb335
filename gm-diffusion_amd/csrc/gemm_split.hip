// float32 contractions on the matrix cores: every float32 operand x is taken as hi + lo with hi = f16(x) and
// lo = f16(x - hi) (22 significand bits between them) and a product a*b is formed as three float16 MFMA passes
//     a_hi*b_hi + a_hi*b_lo + a_lo*b_hi          (a_lo*b_lo, <= 2^-22 of the product, is dropped)
// with float32 accumulation inside the MFMA.  Products of two 11-bit significands are exact in float32, so the result
// differs from an exact float32 FMA chain by ~2^-22 relative per term -- the same order as float32 rounding itself --
// at one third of the float16 matrix rate instead of the 1/16 the float32 vector units (gemm_f32_kernel) or the
// f32-input MFMA forms offer.  This is the path that meets the reference's float32 numerics (the reference only ever
// runs the dual-UNet pipeline in float32: scripts/inference/experiments/formal_improved.py:199) on the matrix cores.
//
//   C[m, n] = act(alpha * sum_k A[m, k] * W[n, k] + bias[n] + rowbias[m / rpg, n] + residual[m, n])      (all float32)
//
// Same operand addressing, LDS-DMA ring, XCD-aware tile order, conv3x3 implicit GEMM and split-K slabs as
// gemm_ring_kernel (gemm.hip); what differs:
//   * a K step is 32 elements: a tile row is 128 bytes of float32 (A, and W when it is given as plain float32) or
//     64 bytes hi + 64 bytes lo of a PRE-SPLIT weight (gmd_split_weights: done once when a model is placed on the device);
//   * fragments are split in registers after the LDS read: v_cvt_pk_f16_f32 for hi, v_fma_mix_f32 (x - hi with the
//     f16 operand converted on the fly) and a second v_cvt_pk_f16_f32 for lo: 16 vector instructions per 8-element
//     fragment against the 12 (TN = 4) or 15 (TN = 5) MFMAs that consume it;
//   * lane group q of a 16x16x32 MFMA owns k = {4q..4q+3, 16+4q..16+4q+3} of the step (both operands alike, so the sum
//     over k is unchanged): its two 16-byte float32 chunks are q and q+4, the conflict-free pattern of lds_off();
//   * range: an operand beyond the float16 range (|x| > 65504) becomes infinite -- the limit of the float16 path; the
//     exact kernel (GMD_F32) has no such limit and stays available.
#include "gemm_shared.h"
#include <stdlib.h>

namespace {

constexpr int BKS = 32;  // float32 elements per K step = 128 bytes

__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) { gmd_split2(a, b, hi, lo); }  // gmd_common.h
__device__ __forceinline__ void split8(const float4& x0, const float4& x1, uint4& hi, uint4& lo) {
    split2(x0.x, x0.y, hi.x, lo.x);
    split2(x0.z, x0.w, hi.y, lo.y);
    split2(x1.x, x1.y, hi.z, lo.z);
    split2(x1.z, x1.w, hi.w, lo.w);
}
__device__ __forceinline__ f32x4 mfma_h(uint4 a, uint4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// byte offset of (pixel, tap) of a float32 channels-last image, or kOOB for padding (see conv_tap_offset in gemm_shared.h)
__device__ __forceinline__ unsigned conv_tap_offset_f32(const GemmParams& p, bool valid, int b, int oy, int ox, int ky, int kx, int chunk) {
    if (!valid) return kOOB;
    int iy, ix;
    if (p.upsample) {
        const int uy = oy + ky - 1, ux = ox + kx - 1;
        if (uy < 0 || ux < 0 || uy >= 2 * p.Hin || ux >= 2 * p.Win) return kOOB;
        iy = uy >> 1;
        ix = ux >> 1;
    } else {
        iy = oy * p.stride + ky - p.pad_lo;
        ix = ox * p.stride + kx - p.pad_lo;
        if (iy < 0 || ix < 0 || iy >= p.Hin || ix >= p.Win) return kOOB;
    }
    return (unsigned)(((b * p.Hin + iy) * p.Win + ix) * p.Cin) * 4u + (unsigned)chunk * 16u;
}

// ------------------------------------------------------------------------------------------------
// epilogues (float32 in, float32 out)
// ------------------------------------------------------------------------------------------------
// From registers: a lane holds four consecutive columns of one row (operands swapped as in gemm.hip).
template <int TM, int TN>
__device__ __forceinline__ void epilogue_regs_f32(const GemmParams& p, const f32x4 (&acc)[TM][TN], int mw, int nw, int frow, int fq, int z,
                                                  int ks) {
    const bool vec_c = (p.ldc % 4 == 0) && (p.sC % 4 == 0);
    const bool vec_r = p.residual != nullptr && (p.ldr % 4 == 0) && (p.sR % 4 == 0);
    const bool vec_n = (p.N % 4 == 0);
    const bool vec_rb = vec_n && (p.ldrb % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.rowbias) & 15) == 0);
    if constexpr (TN % 2 == 0) {
        if (p.act == GMD_ACT_GEGLU) {  // ragged tiles of the fused GEGLU projection (full tiles: epilogue_rows_geglu_f32)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = mw + i * 16 + frow;
                if (m >= p.M) continue;
#pragma unroll
                for (int j = 0; j < TN; j += 2) {
                    const int n = nw + j * 16 + fq * 4;  // interleaved column of the value group (N % 32 == 0: a pair is whole)
                    if (n >= p.N) continue;
                    float o4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float hv = acc[i][j][e] * p.alpha + (p.bias ? p.bias[n + e] : 0.f);
                        const float g = acc[i][j + 1][e] * p.alpha + (p.bias ? p.bias[n + 16 + e] : 0.f);
                        o4[e] = hv * (0.5f * g * (1.0f + erff(g * 0.70710678118654752440f)));
                    }
                    float* o = (float*)p.C + (int64_t)z * p.sC + (int64_t)m * p.ldc + (nw + j * 16) / 2 + fq * 4;
                    *reinterpret_cast<float4*>(o) = make_float4(o4[0], o4[1], o4[2], o4[3]);
                }
            }
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = mw + i * 16 + frow;
        if (m >= p.M) continue;
        const float* rb = p.rowbias ? p.rowbias + (int64_t)(m / p.rows_per_group) * p.ldrb : nullptr;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = nw + j * 16 + fq * 4;
            if (n >= p.N) continue;
            const int nvalid = p.N - n < 4 ? p.N - n : 4;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (p.ksplit > 1) {  // raw partial sums; the epilogue runs in splitk_reduce_f32_kernel
                float* o = p.ws + ((int64_t)ks * p.M + m) * p.N + n;
                if (nvalid == 4 && vec_n) *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
                else
                    for (int e = 0; e < nvalid; ++e) o[e] = v[e];
                continue;
            }
            float add[4] = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) {
                if (nvalid == 4 && vec_n) {
                    const float4 t = *reinterpret_cast<const float4*>(p.bias + n);
                    add[0] = t.x; add[1] = t.y; add[2] = t.z; add[3] = t.w;
                } else {
                    for (int e = 0; e < nvalid; ++e) add[e] = p.bias[n + e];
                }
            }
            float rv[4] = {0.f, 0.f, 0.f, 0.f};
            if (p.residual) {
                const float* res = (const float*)p.residual + (int64_t)z * p.sR + (int64_t)m * p.ldr + n;
                if (nvalid == 4 && vec_r) {
                    const float4 t = *reinterpret_cast<const float4*>(res);
                    rv[0] = t.x; rv[1] = t.y; rv[2] = t.z; rv[3] = t.w;
                } else {
                    for (int e = 0; e < nvalid; ++e) rv[e] = res[e];
                }
            }
            if (rb) {
                if (nvalid == 4 && vec_rb) {
                    const float4 t = *reinterpret_cast<const float4*>(rb + n);
                    rv[0] += t.x; rv[1] += t.y; rv[2] += t.z; rv[3] += t.w;
                } else {
                    for (int e = 0; e < nvalid; ++e) rv[e] += rb[n + e];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e] * p.alpha + add[e] + rv[e], p.act);
            float* o = (float*)p.C + (int64_t)z * p.sC + (int64_t)m * p.ldc + n;
            if (nvalid == 4 && vec_c) *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
            else
                for (int e = 0; e < nvalid; ++e) o[e] = v[e];
        }
    }
}

// Row-contiguous form through a per-wave LDS strip (full tiles): the store path is transaction-bound, so the
// accumulators are parked in LDS, read back row-major and leave as 16-byte stores over whole 64*TN-byte row segments, the
// residual read in the same shape.  SLAB: raw partial sums into this slice's split-K slab instead (no epilogue terms).
// Strip rows are 16 rows at a time per TM tile pair: 32 rows x (TN*16 + 4) floats per wave.
// float32 twin of epilogue_cols_vt (gemm.hip): a V column tile of a fused Q|K|V projection leaves transposed -- lane = column, 32 tokens
// = 128 contiguous bytes of vt_out[sample][column][token ...] per pass
template <int TM, int TN>
__device__ __forceinline__ void epilogue_cols_vt_f32(const GemmParams& p, const f32x4 (&acc)[TM][TN], float* strip, int mw, int nw, int lane) {
    constexpr int NCOL = TN * 16, ROWF = NCOL + 4;
    const int frow = lane & 15, fq = lane >> 4;
    const int b = mw / p.vt_tokens, tok0 = mw - b * p.vt_tokens;
#pragma unroll
    for (int h = 0; h < TM / 2; ++h) {
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                *reinterpret_cast<float4*>(strip + (i2 * 16 + frow) * ROWF + j * 16 + fq * 4) =
                    make_float4(acc[2 * h + i2][j][0], acc[2 * h + i2][j][1], acc[2 * h + i2][j][2], acc[2 * h + i2][j][3]);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < (NCOL + 63) / 64; ++k) {
            const int cc = lane + 64 * k < NCOL ? lane + 64 * k : NCOL - 1;  // every lane reads: no EXEC change around the loop
            const float bz = p.bias ? p.bias[nw + cc] : 0.f;
            float v[32];
#pragma unroll
            for (int r = 0; r < 32; ++r) v[r] = strip[r * ROWF + cc] * p.alpha + bz;
            if (lane + 64 * k < NCOL) {
                float* o = (float*)p.vt_out + ((int64_t)b * (p.N - p.vt_col0) + (nw + cc - p.vt_col0)) * p.vt_ld + tok0 + h * 32;
#pragma unroll
                for (int q = 0; q < 8; ++q) *reinterpret_cast<float4*>(o + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// (PLAIN: no activation -- a compile-time fact for the item loop, as in epilogue_rows of gemm.hip: with the run-time switch the
// never-taken SiLU / quick-GELU bodies made every unrolled item hundreds of instructions and the epilogue instruction-fetch-bound.)
template <int TM, int TN, bool SLAB, bool PLAIN>
__device__ __forceinline__ void epilogue_rows_f32_impl(const GemmParams& p, const f32x4 (&acc)[TM][TN], float* strip, int mw, int nw, int lane,
                                                       int z, int ks) {
    static_assert(TM % 2 == 0, "halves of two 16-row tiles");
    if (!SLAB && p.vt_out && nw >= p.vt_col0) {  // wave-uniform: a V column tile of a fused Q|K|V projection
        epilogue_cols_vt_f32<TM, TN>(p, acc, strip, mw, nw, lane);
        return;
    }
    constexpr int NCOL = TN * 16, ROWF = NCOL + 4, CH = NCOL / 4;
    constexpr int ITER = (32 * CH + 63) / 64;
    const int frow = lane & 15, fq = lane >> 4;
    int g0 = 0, grem = 0;
    if (!SLAB && p.rowbias) {
        g0 = mw / p.rows_per_group;
        grem = mw - g0 * p.rows_per_group;
    }
    float* slab = SLAB ? p.ws + (int64_t)ks * p.M * p.N : nullptr;
    float cs[(NCOL + 63) / 64] = {}, cq[(NCOL + 63) / 64] = {};  // column sums of this wave tile (p.colstats only; round 4)
#pragma unroll
    for (int h = 0; h < TM / 2; ++h) {
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                *reinterpret_cast<float4*>(strip + (i2 * 16 + frow) * ROWF + j * 16 + fq * 4) =
                    make_float4(acc[2 * h + i2][j][0], acc[2 * h + i2][j][1], acc[2 * h + i2][j][2], acc[2 * h + i2][j][3]);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < ITER; ++t) {
            const int idx = lane + 64 * t;
            const int r = idx / CH, c = idx - r * CH;
            if (32 * CH % 64 != 0 && r >= 32) continue;
            const int m = mw + h * 32 + r, n = nw + c * 4;
            const float4 a = *reinterpret_cast<const float4*>(strip + r * ROWF + c * 4);
            if (SLAB) {
                *reinterpret_cast<float4*>(slab + (int64_t)m * p.N + n) = a;
                continue;
            }
            float v[4] = {a.x, a.y, a.z, a.w};
            float add[4] = {0.f, 0.f, 0.f, 0.f};
            if (p.residual) {
                const float4 w = *reinterpret_cast<const float4*>((const float*)p.residual + (int64_t)z * p.sR + (int64_t)m * p.ldr + n);
                add[0] = w.x; add[1] = w.y; add[2] = w.z; add[3] = w.w;
            }
            if (p.rowbias) {
                const int ro = h * 32 + r;
                const int grp = p.rows_per_group >= TM * 16 ? g0 + (grem + ro >= p.rows_per_group ? 1 : 0) : m / p.rows_per_group;
                const float4 t0 = *reinterpret_cast<const float4*>(p.rowbias + (int64_t)grp * p.ldrb + n);
                add[0] += t0.x; add[1] += t0.y; add[2] += t0.z; add[3] += t0.w;
            }
            float bz[4] = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) {
                const float4 t0 = *reinterpret_cast<const float4*>(p.bias + n);
                bz[0] = t0.x; bz[1] = t0.y; bz[2] = t0.z; bz[3] = t0.w;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e] * p.alpha + bz[e] + add[e], PLAIN ? GMD_ACT_NONE : p.act);  // same association as epilogue_regs_f32
            if (p.c_split) gmd_store_split4((float*)p.C, (int64_t)m * p.ldc + n, v[0], v[1], v[2], v[3]);  // (batch 1, ldc = row length: host-checked)
            else *reinterpret_cast<float4*>((float*)p.C + (int64_t)z * p.sC + (int64_t)m * p.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
            if (p.colstats)  // the stored values go back to the strip for the column pass below
                *reinterpret_cast<float4*>(strip + r * ROWF + c * 4) = make_float4(v[0], v[1], v[2], v[3]);
        }
        __builtin_amdgcn_wave_barrier();
        if (!SLAB && p.colstats) {  // lane -> column (and column 64 + lane): 32 conflict-free reads each, fixed order (gemm_shared.h)
            colstats_pass<NCOL, ROWF>(strip, lane, cs, cq);
            __builtin_amdgcn_wave_barrier();
        }
    }
    // producer statistics for a following GroupNorm: {sum, sum of squares} of the STORED values over this wave tile's TM*16 = 64
    // rows (one row block of the statistics) and each bucket of cs_bucket adjacent columns
    if (!SLAB && p.colstats)
        colstats_store<NCOL, ROWF>(strip, lane, cs, cq, p.cs_bucket,
                                   p.colstats + ((int64_t)(mw / (TM * 16)) * (p.N / p.cs_bucket) + nw / p.cs_bucket) * 2);
}

template <int TM, int TN, bool SLAB>
__device__ __forceinline__ void epilogue_rows_f32(const GemmParams& p, const f32x4 (&acc)[TM][TN], float* strip, int mw, int nw, int lane, int z, int ks) {
    if (SLAB || p.act == GMD_ACT_NONE) epilogue_rows_f32_impl<TM, TN, SLAB, true>(p, acc, strip, mw, nw, lane, z, ks);
    else epilogue_rows_f32_impl<TM, TN, SLAB, false>(p, acc, strip, mw, nw, lane, z, ks);
}

// GEGLU (GEGLU.forward of diffusers: value * gelu_erf(gate)): the W rows are interleaved at load time in 16-row [value | gate]
// groups, so tile j holds 16 value columns and tile j+1 the matching gate columns IN THE SAME LANE; the product is formed in
// registers and only the [M, N/2] result is written -- the [tokens, 8C] float32 projection (335 MB per level-0 block at batch 8)
// never reaches HBM.  Full tiles only (checked by the caller); erf by libm's erff: this is the float32 path.
template <int TM, int TN>
__device__ __forceinline__ void epilogue_rows_geglu_f32(const GemmParams& p, const f32x4 (&acc)[TM][TN], float* strip, int mw, int nw, int lane,
                                                        int z) {
    static_assert(TM % 2 == 0 && TN % 2 == 0, "halves of two 16-row tiles; value/gate tile pairs");
    constexpr int NCOL = TN * 8, ROWF = NCOL + 4, CH = NCOL / 4;  // output columns of this wave
    constexpr int ITER = (32 * CH + 63) / 64;
    const int frow = lane & 15, fq = lane >> 4;
    float bz[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const float4 t = p.bias ? *reinterpret_cast<const float4*>(p.bias + nw + j * 16 + fq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        bz[j][0] = t.x; bz[j][1] = t.y; bz[j][2] = t.z; bz[j][3] = t.w;
    }
#pragma unroll
    for (int h = 0; h < TM / 2; ++h) {
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int j = 0; j < TN; j += 2) {
                float o4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float hv = acc[2 * h + i2][j][e] * p.alpha + bz[j][e];
                    const float g = acc[2 * h + i2][j + 1][e] * p.alpha + bz[j + 1][e];
                    o4[e] = hv * (0.5f * g * (1.0f + erff(g * 0.70710678118654752440f)));
                }
                *reinterpret_cast<float4*>(strip + (i2 * 16 + frow) * ROWF + (j / 2) * 16 + fq * 4) = make_float4(o4[0], o4[1], o4[2], o4[3]);
            }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < ITER; ++t) {
            const int idx = lane + 64 * t;
            const int r = idx / CH, c = idx - r * CH;
            if (32 * CH % 64 != 0 && r >= 32) continue;
            const float4 o4 = *reinterpret_cast<const float4*>(strip + r * ROWF + c * 4);
            if (p.c_split) gmd_store_split4((float*)p.C, (int64_t)(mw + h * 32 + r) * p.ldc + nw / 2 + c * 4, o4.x, o4.y, o4.z, o4.w);
            else *reinterpret_cast<float4*>((float*)p.C + (int64_t)z * p.sC + (int64_t)(mw + h * 32 + r) * p.ldc + nw / 2 + c * 4) = o4;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// split-K: sum the float32 partial slabs ws[s][m][n] in a fixed order, then the fused epilogue
__global__ __launch_bounds__(256) void splitk_reduce_f32_kernel(const GemmParams p) {
    const int NC = (p.N + 3) / 4;
    const int64_t total = (int64_t)p.M * NC;
    const int64_t slab = (int64_t)p.M * p.N;
    const bool vec = (p.N % 4 == 0) && (p.ldc % 4 == 0) && (p.residual == nullptr || p.ldr % 4 == 0) &&
                     (p.rowbias == nullptr || (p.ldrb % 4 == 0 && (reinterpret_cast<uintptr_t>(p.rowbias) & 15) == 0));
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / NC), n = (int)(i - (int64_t)m * NC) * 4;
        const int nvalid = p.N - n < 4 ? p.N - n : 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        const float* src = p.ws + (int64_t)m * p.N + n;
        for (int s = 0; s < p.ksplit; ++s, src += slab) {
            if (nvalid == 4 && vec) {
                const float4 a = *reinterpret_cast<const float4*>(src);
                v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
            } else {
                for (int j = 0; j < nvalid; ++j) v[j] += src[j];
            }
        }
        const float* rb = p.rowbias ? p.rowbias + (int64_t)(m / p.rows_per_group) * p.ldrb : nullptr;
        const float* res = p.residual ? (const float*)p.residual + (int64_t)m * p.ldr + n : nullptr;
        float* o = (float*)p.C + (int64_t)m * p.ldc + n;
        for (int j = 0; j < nvalid; ++j) {
            float x = v[j] * p.alpha;
            float add = 0.f;
            if (p.bias) x += p.bias[n + j];
            if (res) add += res[j];
            if (rb) add += rb[n + j];
            v[j] = apply_act(x + add, p.act);  // (acc*alpha + bias) + (residual + rowbias): the association of the tile epilogues
        }
        if (nvalid == 4 && vec) *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
        else
            for (int j = 0; j < nvalid; ++j) o[j] = v[j];
    }
}

// ------------------------------------------------------------------------------------------------
// ring kernel: WM x WN waves, wave tile (TM*16) x (TN*16), NST-stage LDS ring filled by LDS-DMA with counted waits
// ------------------------------------------------------------------------------------------------
// ASPLIT (round 4, GMD_F32SA): the A operand arrives pre-split like a weight -- its producer (GroupNorm / LayerNorm apply, the GEGLU
// epilogue: gmd_store_split4) wrote [hi | lo] per 32 elements -- and the fragment is two 16-byte reads instead of two reads + 16
// conversion instructions.  Same values, same products, same order: bit-identical to the in-kernel split.
template <bool CONV, bool WSPLIT, int WM, int WN, int TM, int TN, int NST, bool ASPLIT = false>
__global__ __launch_bounds__(WM* WN * 64, (WM * WN) >= 4 ? (WM * WN) / 4 : 1) void gemm_split_kernel(const GemmParams p) {
    constexpr int NWAVES = WM * WN;
    constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
    constexpr int RPP = NWAVES * 8;                 // tile rows written per staging pass (8 rows per wave instruction)
    constexpr int NA = (BM + RPP - 1) / RPP;        // A staging slots per thread
    constexpr int NW = (BN + RPP - 1) / RPP;        // W staging slots per thread (the last may be invalid for some waves)
    constexpr bool A_RAGGED = (BM % RPP) != 0, W_RAGGED = (BN % RPP) != 0;
    constexpr int D = NST - 1;                      // prefetch distance in tiles
    static_assert(!A_RAGGED && BN % 8 == 0 && NST >= 2 && NST <= 3, "bad ring geometry");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int kStage = (BM + BN) * 128;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid / WN, wc = wid % WN;
    int m0, n0;
    {   // XCD-aware tile order (gemm_ring_kernel)
        const int nwg = gridDim.x, id = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, j = id >> 3;
        const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
        const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
        if (p.M >= p.N || CONV) {
            const int mt = L / tiles_n;
            m0 = mt * BM;
            n0 = (L - mt * tiles_n) * BN;
        } else {
            const int nt = L / tiles_m;
            n0 = nt * BN;
            m0 = (L - nt * tiles_m) * BM;
        }
    }
    const int z = p.ksplit > 1 ? 0 : blockIdx.z;
    const int ks = p.ksplit > 1 ? blockIdx.z : 0;
    const int srow = tid >> 3;                                    // 0 .. RPP-1
    const int chunk = (tid & 7) ^ ((srow >> 1) & 7);              // swizzled SOURCE chunk (RPP is a multiple of 16)
    const int wuni = __builtin_amdgcn_readfirstlane(wid);

    unsigned aoff[NA], woff[NW];
    int pb[NA], py[NA], px[NA];
    bool pv[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int m = m0 + srow + RPP * i;
        pv[i] = m < p.M;
        pb[i] = py[i] = px[i] = 0;
        if (CONV) {
            if (pv[i]) {
                const int hw = p.Hout * p.Wout;
                if (((hw & (hw - 1)) | (p.Wout & (p.Wout - 1))) == 0) {
                    const int sh = __builtin_ctz(hw), sw = __builtin_ctz(p.Wout);
                    pb[i] = m >> sh;
                    const int rem = m & (hw - 1);
                    py[i] = rem >> sw;
                    px[i] = rem & (p.Wout - 1);
                } else {
                    pb[i] = m / hw;
                    const int rem = m - pb[i] * hw;
                    py[i] = rem / p.Wout;
                    px[i] = rem - py[i] * p.Wout;
                }
            }
            aoff[i] = kOOB;
        } else {
            aoff[i] = pv[i] ? (unsigned)m * (unsigned)p.lda * 4u + (unsigned)chunk * 16u : kOOB;
        }
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int rl = srow + RPP * i;
        const int n = n0 + rl;
        woff[i] = (rl < BN && n < p.N) ? (unsigned)n * (unsigned)p.ldw * 4u + (unsigned)chunk * 16u : kOOB;
    }
    const bool w_last = !W_RAGGED || ((NW - 1) * RPP + wuni * 8 < BN);

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk_total = p.K / BKS;
    const int per = (nk_total + p.ksplit - 1) / p.ksplit;
    const int kt_begin = ks * per;
    const int nk = (kt_begin + per <= nk_total ? per : nk_total - kt_begin);
    int tap = 0, c0 = 0, cb0 = 0;
    bool newtap = true;
    if (CONV) {
        const int sb = p.cblk / BKS, per_cb = 9 * sb;
        const int cbi = kt_begin / per_cb, rem = kt_begin - cbi * per_cb;
        tap = rem / sb;
        cb0 = cbi * p.cblk;
        c0 = cb0 + (rem - tap * sb) * BKS;
    }

    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);
    u32x4 dA, dW;
    {
        const uint64_t ba = (uint64_t)((const float*)p.A + (int64_t)z * p.sA), bw = (uint64_t)((const float*)p.W + (int64_t)z * p.sW);
        dA = u32x4{(unsigned)__builtin_amdgcn_readfirstlane((unsigned)ba), (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(ba >> 32) & 0xffffu),
                   (unsigned)__builtin_amdgcn_readfirstlane(p.a_bytes), 0x00020000u};
        dW = u32x4{(unsigned)__builtin_amdgcn_readfirstlane((unsigned)bw), (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(bw >> 32) & 0xffffu),
                   (unsigned)__builtin_amdgcn_readfirstlane(p.w_bytes), 0x00020000u};
    }
    auto dma16 = [&](const u32x4& desc, unsigned lds_addr, unsigned voff, unsigned soff) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds"
                     :
                     : "v"(voff), "s"(lds_addr), "s"(desc), "s"(soff)
                     : "memory", "m0");
#pragma clang diagnostic pop
    };
    auto dma_tile = [&](int kt, int stage_idx) {
        unsigned kbytes = (unsigned)(kt_begin + kt) * (BKS * 4);
        unsigned abytes = kbytes;
        if (CONV) {
            if (newtap) {
                const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
                for (int i = 0; i < NA; ++i) aoff[i] = conv_tap_offset_f32(p, pv[i], pb[i], py[i], px[i], ky, kx, chunk);
                newtap = false;
            }
            abytes = (unsigned)c0 * 4u;
            kbytes = (unsigned)(tap * p.Cin + c0) * 4u;  // weights are [Cout][tap][Cin], 4 bytes per k in either layout
        }
        const unsigned stage = lds_base + (unsigned)stage_idx * kStage + (unsigned)wuni * (8 * 128);
#pragma unroll
        for (int i = 0; i < NA; ++i) dma16(dA, stage + i * (RPP * 128), aoff[i], abytes);
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            if (i + 1 < NW || w_last) dma16(dW, stage + BM * 128 + i * (RPP * 128), woff[i], kbytes);
        }
        if (CONV) {
            c0 += BKS;
            if (c0 >= cb0 + p.cblk) {
                c0 = cb0;
                ++tap;
                newtap = true;
                if (tap == 9) { tap = 0; cb0 += p.cblk; c0 = cb0; }
            }
        }
    };
    auto wait_tiles = [&](int ahead) {
        if (ahead <= 0) { wait_vmcnt<0>(); return; }
        if (w_last) {
            if (ahead == 1) wait_vmcnt<NA + NW>();
            else wait_vmcnt<2 * (NA + NW)>();
        } else {
            if (ahead == 1) wait_vmcnt<NA + NW - 1>();
            else wait_vmcnt<2 * (NA + NW - 1)>();
        }
    };

    const int frow = lane & 15, fq = lane >> 4;
    auto compute = [&](int stage_idx) {
        const unsigned char* sA = smem + stage_idx * kStage;
        const unsigned char* sW = sA + BM * 128;
        uint4 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = wr * (TM * 16) + i * 16 + frow;
            if (ASPLIT) {
                ah[i] = *reinterpret_cast<const uint4*>(sA + lds_off(row, fq));
                al[i] = *reinterpret_cast<const uint4*>(sA + lds_off(row, 4 + fq));
                continue;
            }
            const float4 x0 = *reinterpret_cast<const float4*>(sA + lds_off(row, fq));
            const float4 x1 = *reinterpret_cast<const float4*>(sA + lds_off(row, 4 + fq));
            split8(x0, x1, ah[i], al[i]);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int row = wc * (TN * 16) + j * 16 + frow;
            if (WSPLIT) {
                bh[j] = *reinterpret_cast<const uint4*>(sW + lds_off(row, fq));
                bl[j] = *reinterpret_cast<const uint4*>(sW + lds_off(row, 4 + fq));
            } else {
                const float4 x0 = *reinterpret_cast<const float4*>(sW + lds_off(row, fq));
                const float4 x1 = *reinterpret_cast<const float4*>(sW + lds_off(row, 4 + fq));
                split8(x0, x1, bh[j], bl[j]);
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                acc[i][j] = mfma_h(bl[j], ah[i], acc[i][j]);  // D[n][m]: lane = (m, 4 n's); small terms first
                acc[i][j] = mfma_h(bh[j], al[i], acc[i][j]);
                acc[i][j] = mfma_h(bh[j], ah[i], acc[i][j]);
            }
    };

    if (nk > 0) {
#pragma unroll
        for (int t = 0; t < D; ++t)
            if (t < nk) dma_tile(t, t);
        int st_cur = 0, st_fill = D % NST;
        for (int kt = 0; kt < nk; ++kt) {
            const int rem = nk - 1 - kt;
            wait_tiles(rem < D - 1 ? rem : D - 1);
            __syncthreads();
            if (kt + D < nk) dma_tile(kt + D, st_fill);
            compute(st_cur);
            st_cur = st_cur + 1 == NST ? 0 : st_cur + 1;
            st_fill = st_fill + 1 == NST ? 0 : st_fill + 1;
        }
    }
    const int mw = m0 + wr * (TM * 16), nw = n0 + wc * (TN * 16);
    constexpr int kStrip = 32 * (TN * 16 + 4);  // floats per wave
    constexpr bool strips_fit = (TM % 2 == 0) && ((size_t)NWAVES * kStrip * 4 <= (size_t)NST * kStage);
    if constexpr (strips_fit) {
        const bool full = m0 + BM <= p.M && n0 + BN <= p.N && (p.N & 3) == 0;
        if constexpr (TN % 2 == 0) {
            if (p.act == GMD_ACT_GEGLU && full) {
                __syncthreads();
                epilogue_rows_geglu_f32<TM, TN>(p, acc, reinterpret_cast<float*>(smem) + wid * kStrip, mw, nw, lane, z);
                return;
            }
        }
        if (p.ksplit > 1 && full) {
            __syncthreads();
            epilogue_rows_f32<TM, TN, true>(p, acc, reinterpret_cast<float*>(smem) + wid * kStrip, mw, nw, lane, z, ks);
            return;
        }
        const bool rows_ok = p.ksplit <= 1 && full && (p.ldc & 3) == 0 && (p.sC & 3) == 0 &&
                             (p.residual == nullptr || ((p.ldr & 3) == 0 && (p.sR & 3) == 0)) &&
                             (p.rowbias == nullptr || ((p.ldrb & 3) == 0 && (reinterpret_cast<uintptr_t>(p.rowbias) & 15) == 0)) &&
                             (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0;
        if (rows_ok) {
            __syncthreads();  // every wave is done with the K-loop stages: the strips overwrite them
            epilogue_rows_f32<TM, TN, false>(p, acc, reinterpret_cast<float*>(smem) + wid * kStrip, mw, nw, lane, z, ks);
            return;
        }
    }
    epilogue_regs_f32<TM, TN>(p, acc, mw, nw, frow, fq, z, ks);
}

// ------------------------------------------------------------------------------------------------
// Loader / converter form (round 4): ONE workgroup per CU = 4 consumer waves (2 x 2, wave tile 64 x 16 TN) + 4 loader waves, tile
// 128 x (32 TN), four 36 KB LDS stages, a K step of 32 elements, pre-split weights (GMD_F32SW) only.
//
// Why: in gemm_split_kernel above every wave issues its own LDS-DMA (an instruction that blocks while the CU's memory queue is full:
// gemm.hip, gemm_pp_kernel), reads float32 fragments and splits them in registers -- 16 vector instructions per A fragment, done by
// BOTH waves that share the fragment's rows, 2.4 VALU per MFMA in all (profiles/r03_pmc.txt) -- in front of the MFMAs that need them.
// Here the loader waves own the DMA and, once a tile has landed, split its activation rows ONCE, IN PLACE, into the layout the
// pre-split weights already have ([hi 64 B | lo 64 B] per 128-byte row: item (row, q) reads float32 chunks q and 4+q -- the k's lane
// group q of a 16x16x32 MFMA consumes -- and writes the f16 hi halves back to chunk q, the lo halves to chunk 4+q).  The consumers
// then read ready-made f16 fragments for both operands (18 ds_read_b128 per 60 MFMAs) and do nothing but multiply.
//
//   stage life: landing (DMA) -> converting (loaders, first half of the next K step) -> ready -> consumed -> free.  Per K step kt:
//   loader:    convert tile kt+1 ; lgkmcnt(0) ; M(kt) ; issue tile kt+3 (stage of tile kt-1) ; vmcnt(tile kt+2 landed) ; B(kt+1)
//   consumer:  first half of the MFMAs on tile kt ; M(kt) ; reads of tile kt+1 -> second fragment set ; second half ; B(kt+1)
//   B(kt) guarantees: tile kt converted (M(kt-1)), tile kt+1 landed, every read of tile kt-1 returned (they were consumed).
// The MFMA order per accumulator is gemm_split_kernel's (lo x hi, hi x lo, hi x hi per K step, K steps in the same order): results are
// bit-identical to it.
// ------------------------------------------------------------------------------------------------
template <bool CONV, int TN>
__global__ __launch_bounds__(512, 2) void gemm_split_lc_kernel(const GemmParams p) {
    constexpr int WN = 2, NCONS = 4, LW = 4, NST = 4, TM = 4;
    constexpr int BM = 2 * TM * 16, BN = WN * TN * 16;
    constexpr int RPP = LW * 8;
    constexpr int NA = BM / RPP, NW = BN / RPP, NP = NA + NW;
    constexpr int NCV = BM * 4 / (LW * 64);          // conversion items (row, q) per loader lane and tile
    static_assert(BM % RPP == 0 && BN % RPP == 0 && (BM * 4) % (LW * 64) == 0, "tile rows must be whole staging passes");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int kStage = (BM + BN) * 128;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wuni = __builtin_amdgcn_readfirstlane(wid);
    const bool loader = wuni >= NCONS;
    int m0, n0;
    {
        const int nwg = gridDim.x, id = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, j = id >> 3;
        const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
        const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
        if (p.M >= p.N || CONV) {
            const int mt = L / tiles_n;
            m0 = mt * BM;
            n0 = (L - mt * tiles_n) * BN;
        } else {
            const int nt = L / tiles_m;
            n0 = nt * BN;
            m0 = (L - nt * tiles_m) * BM;
        }
    }
    const int z = 0;
    const int ks = p.ksplit > 1 ? blockIdx.z : 0;
    const int nk_total = p.K / BKS;
    const int per = (nk_total + p.ksplit - 1) / p.ksplit;
    const int kt_begin = ks * per;
    const int nk = (kt_begin + per <= nk_total ? per : nk_total - kt_begin);
    auto seg_barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    if (loader) {
        const int lw = wuni - NCONS;
        const int ltid = tid - NCONS * 64;
        const int srow = ltid >> 3;
        const int chunk = (ltid & 7) ^ ((srow >> 1) & 7);
        unsigned aoff[NA], woff[NW];
        int pb[NA], py[NA], px[NA];
        bool pv[NA];
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int m = m0 + srow + RPP * i;
            pv[i] = m < p.M;
            pb[i] = py[i] = px[i] = 0;
            if (CONV) {
                if (pv[i]) {
                    const int hw = p.Hout * p.Wout;
                    if (((hw & (hw - 1)) | (p.Wout & (p.Wout - 1))) == 0) {
                        const int sh = __builtin_ctz(hw), sw = __builtin_ctz(p.Wout);
                        pb[i] = m >> sh;
                        const int rem = m & (hw - 1);
                        py[i] = rem >> sw;
                        px[i] = rem & (p.Wout - 1);
                    } else {
                        pb[i] = m / hw;
                        const int rem = m - pb[i] * hw;
                        py[i] = rem / p.Wout;
                        px[i] = rem - py[i] * p.Wout;
                    }
                }
                aoff[i] = kOOB;
            } else {
                aoff[i] = pv[i] ? (unsigned)m * (unsigned)p.lda * 4u + (unsigned)chunk * 16u : kOOB;
            }
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int n = n0 + srow + RPP * i;
            woff[i] = n < p.N ? (unsigned)n * (unsigned)p.ldw * 4u + (unsigned)chunk * 16u : kOOB;
        }
        int tap = 0, c0 = 0, cb0 = 0;
        bool newtap = true;
        if (CONV) {
            const int sb = p.cblk / BKS, per_cb = 9 * sb;
            const int cbi = kt_begin / per_cb, rem = kt_begin - cbi * per_cb;
            tap = rem / sb;
            cb0 = cbi * p.cblk;
            c0 = cb0 + (rem - tap * sb) * BKS;
        }
        const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);
        u32x4 dA, dW;
        {
            const uint64_t ba = (uint64_t)p.A, bw = (uint64_t)p.W;
            dA = u32x4{(unsigned)__builtin_amdgcn_readfirstlane((unsigned)ba), (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(ba >> 32) & 0xffffu),
                       (unsigned)__builtin_amdgcn_readfirstlane(p.a_bytes), 0x00020000u};
            dW = u32x4{(unsigned)__builtin_amdgcn_readfirstlane((unsigned)bw), (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(bw >> 32) & 0xffffu),
                       (unsigned)__builtin_amdgcn_readfirstlane(p.w_bytes), 0x00020000u};
        }
        auto dma16 = [&](const u32x4& desc, unsigned lds_addr, unsigned voff, unsigned soff) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds"
                         :
                         : "v"(voff), "s"(lds_addr), "s"(desc), "s"(soff)
                         : "memory", "m0");
#pragma clang diagnostic pop
        };
        // pieces [LO, HI) of one tile (A passes first, then W passes); LO == 0 opens the tile (conv: new tap offsets), HI == NP closes it
        auto dma_part = [&](auto LO, auto HI, int kt, int stage_idx) {
            constexpr int lo = decltype(LO)::value, hi = decltype(HI)::value;
            unsigned kbytes = (unsigned)(kt_begin + kt) * (BKS * 4);
            unsigned abytes = kbytes;
            if (CONV) {
                if (lo == 0 && newtap) {
                    const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
                    for (int i = 0; i < NA; ++i) aoff[i] = conv_tap_offset_f32(p, pv[i], pb[i], py[i], px[i], ky, kx, chunk);
                    newtap = false;
                }
                abytes = (unsigned)c0 * 4u;
                kbytes = (unsigned)(tap * p.Cin + c0) * 4u;
            }
            const unsigned stage = lds_base + (unsigned)stage_idx * kStage + (unsigned)lw * (8 * 128);
#pragma unroll
            for (int i = 0; i < NA; ++i)
                if (i >= lo && i < hi) dma16(dA, stage + i * (RPP * 128), aoff[i], abytes);
#pragma unroll
            for (int i = 0; i < NW; ++i)
                if (NA + i >= lo && NA + i < hi) dma16(dW, stage + BM * 128 + i * (RPP * 128), woff[i], kbytes);
            if (CONV && hi == NP) {
                c0 += BKS;
                if (c0 >= cb0 + p.cblk) {
                    c0 = cb0;
                    ++tap;
                    newtap = true;
                    if (tap == 9) { tap = 0; cb0 += p.cblk; c0 = cb0; }
                }
            }
        };
        auto dma_tile = [&](int kt, int stage_idx) { dma_part(IntC<0>{}, IntC<NP>{}, kt, stage_idx); };
        // split the activation rows of one landed tile in place (this wave's share of the (row, q) items)
        auto convert_item = [&](int stage_idx, int j) {
            unsigned char* sA = smem + stage_idx * kStage;
            {
                const int id = ltid + (LW * 64) * j;
                const int row = id >> 2, q = id & 3;
                float4* p0 = reinterpret_cast<float4*>(sA + lds_off(row, q));
                float4* p1 = reinterpret_cast<float4*>(sA + lds_off(row, 4 + q));
                const float4 x0 = *p0, x1 = *p1;
                uint4 hi, lo;
                split8(x0, x1, hi, lo);
                *reinterpret_cast<uint4*>(p0) = hi;
                *reinterpret_cast<uint4*>(p1) = lo;
            }
        };
        auto convert_tile = [&](int stage_idx) {
#pragma unroll
            for (int j = 0; j < NCV; ++j) convert_item(stage_idx, j);
        };
        if (nk > 0) {  // block-uniform
#pragma unroll
            for (int t = 0; t < 3; ++t)
                if (t < nk) dma_tile(t, t);
            // tile 0 landed (at most tiles 1 and 2 outstanding); every loader's pieces: barrier; convert tile 0; barrier
            if (nk >= 3) wait_vmcnt<2 * NP>();
            else if (nk == 2) wait_vmcnt<NP>();
            else wait_vmcnt<0>();
            seg_barrier();          // P0: tile 0 has landed
            convert_tile(0);
            // tile 1 has landed too before B(0) (it is converted during K step 0)
            if (nk >= 3) wait_vmcnt<NP>();
            else wait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            seg_barrier();          // B(0): tile 0 ready, tile 1 landed
            int st_fill = 3;
            for (int kt = 0; kt < nk; ++kt) {
                const int st_next = (kt + 1) & 3;
                const bool more = kt + 3 < nk;
                // The A pieces of tile kt+3 (its stage was freed by B(kt)) go out BETWEEN the conversion items of tile kt+1: an LDS-DMA
                // issued into a full memory queue blocks the wave, a queue that is fed one piece per ~100 cycles does not, and the
                // split's VALU / LDS work fills the gaps (issued back to back in front of the conversion the K step took 2000 cycles)
                static_assert(NCV == 2 && NA == 4, "interleave below");
                if (more) dma_part(IntC<0>{}, IntC<1>{}, kt + 3, st_fill);
                if (kt + 1 < nk) convert_item(st_next, 0);
                if (more) dma_part(IntC<1>{}, IntC<3>{}, kt + 3, st_fill);
                if (kt + 1 < nk) convert_item(st_next, 1);
                if (more) dma_part(IntC<3>{}, IntC<4>{}, kt + 3, st_fill);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                seg_barrier();      // M(kt): tile kt+1 is ready
                if (more) dma_part(IntC<4>{}, IntC<NP>{}, kt + 3, st_fill);
                // tile kt+2 has landed: only tile kt+3 (just issued) may remain -- where it exists
                if (kt + 3 < nk) wait_vmcnt<NP>();
                else wait_vmcnt<0>();
                seg_barrier();      // B(kt+1)
                st_fill = (st_fill + 1) & 3;
            }
        }
        __syncthreads();
        return;
    }

    // -------------------------------------------------------------------- consumer waves
    const int wr = wid / WN, wc = wid % WN;
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fq = lane >> 4;
    uint4 ah0[TM], al0[TM], bh0[TN], bl0[TN], ah1[TM], al1[TM], bh1[TN], bl1[TN];
    auto frag_reads = [&](int stage_idx, uint4 (&ah)[TM], uint4 (&al)[TM], uint4 (&bh)[TN], uint4 (&bl)[TN]) {
        const unsigned char* sA = smem + stage_idx * kStage;
        const unsigned char* sW = sA + BM * 128;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = wr * (TM * 16) + i * 16 + frow;
            ah[i] = *reinterpret_cast<const uint4*>(sA + lds_off(row, fq));
            al[i] = *reinterpret_cast<const uint4*>(sA + lds_off(row, 4 + fq));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int row = wc * (TN * 16) + j * 16 + frow;
            bh[j] = *reinterpret_cast<const uint4*>(sW + lds_off(row, fq));
            bl[j] = *reinterpret_cast<const uint4*>(sW + lds_off(row, 4 + fq));
        }
    };
    // rows [I0, I1) of the wave tile: the same three products per accumulator, in the same order, as gemm_split_kernel
    auto mfmas = [&](auto I0, auto I1, const uint4 (&ah)[TM], const uint4 (&al)[TM], const uint4 (&bh)[TN], const uint4 (&bl)[TN]) {
#pragma unroll
        for (int i = decltype(I0)::value; i < decltype(I1)::value; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                acc[i][j] = mfma_h(bl[j], ah[i], acc[i][j]);
                acc[i][j] = mfma_h(bh[j], al[i], acc[i][j]);
                acc[i][j] = mfma_h(bh[j], ah[i], acc[i][j]);
            }
    };
    if (nk > 0) {
        seg_barrier();  // P0
        seg_barrier();  // B(0): tile 0 is ready
        frag_reads(0, ah0, al0, bh0, bl0);
        auto kstep = [&](int kt, uint4 (&ah)[TM], uint4 (&al)[TM], uint4 (&bh)[TN], uint4 (&bl)[TN], uint4 (&nah)[TM], uint4 (&nal)[TM],
                         uint4 (&nbh)[TN], uint4 (&nbl)[TN]) {
            mfmas(IntC<0>{}, IntC<TM / 2>{}, ah, al, bh, bl);
            seg_barrier();  // M(kt): tile kt+1 is ready
            frag_reads((kt + 1) & 3, nah, nal, nbh, nbl);  // (behind the last K step: a stale stage into registers nobody uses)
            mfmas(IntC<TM / 2>{}, IntC<TM>{}, ah, al, bh, bl);
            seg_barrier();  // B(kt+1)
        };
        int kt = 0;
        for (; kt + 1 < nk; kt += 2) {
            kstep(kt, ah0, al0, bh0, bl0, ah1, al1, bh1, bl1);
            kstep(kt + 1, ah1, al1, bh1, bl1, ah0, al0, bh0, bl0);
        }
        if (kt < nk) kstep(kt, ah0, al0, bh0, bl0, ah1, al1, bh1, bl1);
    }
    const int mw = m0 + wr * (TM * 16), nw = n0 + wc * (TN * 16);
    constexpr int kStrip = 32 * (TN * 16 + 4);
    const bool full = m0 + BM <= p.M && n0 + BN <= p.N && (p.N & 3) == 0;
    __syncthreads();  // every wave (loaders included) is done with the stages: the strips below overwrite them
    if constexpr (TN % 2 == 0) {
        if (p.act == GMD_ACT_GEGLU && full) {
            epilogue_rows_geglu_f32<TM, TN>(p, acc, reinterpret_cast<float*>(smem) + wid * kStrip, mw, nw, lane, z);
            return;
        }
    }
    if (p.ksplit > 1 && full) {
        epilogue_rows_f32<TM, TN, true>(p, acc, reinterpret_cast<float*>(smem) + wid * kStrip, mw, nw, lane, z, ks);
        return;
    }
    const bool rows_ok = p.ksplit <= 1 && full && (p.ldc & 3) == 0 && (p.sC & 3) == 0 &&
                         (p.residual == nullptr || ((p.ldr & 3) == 0 && (p.sR & 3) == 0)) &&
                         (p.rowbias == nullptr || ((p.ldrb & 3) == 0 && (reinterpret_cast<uintptr_t>(p.rowbias) & 15) == 0)) &&
                         (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0;
    if (rows_ok) epilogue_rows_f32<TM, TN, false>(p, acc, reinterpret_cast<float*>(smem) + wid * kStrip, mw, nw, lane, z, ks);
    else epilogue_regs_f32<TM, TN>(p, acc, mw, nw, frow, fq, z, ks);
}

// one launch of the pre-split weight layout: out[n][kb][plane][q][j], plane 0 = hi, 1 = lo; chunk q holds
// k = kb*32 + {4q..4q+3, 16+4q..16+4q+3}
__global__ __launch_bounds__(256) void split_weights_kernel(const float* __restrict__ w, unsigned* __restrict__ out, int64_t N, int64_t K, int64_t ldw) {
    const int64_t nblk = K / 32, total = N * nblk * 4;  // one thread per (row, k block, chunk q)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(i & 3);
        const int64_t blk = i >> 2, n = blk / nblk, kb = blk - n * nblk;
        const float* src = w + n * ldw + kb * 32 + 4 * q;
        const float4 x0 = *reinterpret_cast<const float4*>(src), x1 = *reinterpret_cast<const float4*>(src + 16);
        uint4 hi, lo;
        split8(x0, x1, hi, lo);
        unsigned* dst = out + blk * 32 + q * 4;  // 128 bytes = 32 words per (row, k block)
        *reinterpret_cast<uint4*>(dst) = hi;
        *reinterpret_cast<uint4*>(dst + 16) = lo;
    }
}

struct SplitPlan {
    int bm, bn, ksplit;
};
}  // namespace
int gmd_split_colstats_ok(int M, int N, int K, int batch, int64_t ws_bytes, int bucket);
namespace {

// Tile / split-K selection: the 16-bit heuristic of gemm.hip (make_plan) in units of 64-deep K steps; the tiles are
// 128 x 160 / 128 x 128 (two workgroups per CU) and 64 x 64 for launches that cannot put 256 large tiles on the chip.
SplitPlan make_split_plan(int M, int N, int K, int batch, int64_t ws_bytes) {
    SplitPlan pl{64, 64, 1};
    if (M >= 96 && N >= 96) {
        pl.bm = 128;
        pl.bn = (N % 160 == 0) ? 160 : 128;
    }
    const int64_t tiles = (int64_t)((M + pl.bm - 1) / pl.bm) * ((N + pl.bn - 1) / pl.bn) * batch;
    const int nk = K / 64;
    if (batch == 1 && tiles < 160 && nk >= 24) {
        int ks = (int)(((tiles >= 64 ? 512 : 256) + tiles / 2) / tiles);
        if (ks > nk / 8) ks = nk / 8;
        if (ks > 16) ks = 16;
        if (ks > 1 && (int64_t)ks * M * N * (int64_t)sizeof(float) <= ws_bytes) pl.ksplit = ks;
    }
    if (batch == 1 && pl.bm == 128 && tiles >= 224 && tiles <= 256 && nk >= 160 && 2 * (int64_t)M * N * (int64_t)sizeof(float) <= ws_bytes)
        pl.ksplit = 2;
    if (pl.ksplit == 1 && batch > 1 && M <= 640 && pl.bm == 128) { pl.bm = 64; pl.bn = 64; }
    if (pl.ksplit == 1 && tiles < 256 && pl.bm == 128) { pl.bm = 64; pl.bn = 64; }
    return pl;
}

template <bool CONV, bool WSPLIT, int WM, int WN, int TM, int TN, int NST, bool ASPLIT = false>
hipError_t launch_split(const GemmParams& p, int gz, hipStream_t s) {
    constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
    constexpr size_t smem = (size_t)NST * (BM + BN) * 128;
    if (smem > 64 * 1024) {
        hipError_t e = opt_in_lds(reinterpret_cast<const void*>(&gemm_split_kernel<CONV, WSPLIT, WM, WN, TM, TN, NST, ASPLIT>), (int)smem);
        if (e != hipSuccess) return e;
    }
    dim3 grid(((p.N + BN - 1) / BN) * ((p.M + BM - 1) / BM), 1, gz);
    gemm_split_kernel<CONV, WSPLIT, WM, WN, TM, TN, NST, ASPLIT><<<grid, WM * WN * 64, smem, s>>>(p);
    return hipGetLastError();
}

// 0 (default): never; -1: the loader / converter kernel where it fits (GMD_SPLIT_LC=1); 1: wherever it is instantiated
// (gmd_gemm_plan_override(.., pf = 244, ..); pf = 9: never).  NOT the default: alone on the chip it wins 3-9 % on one-round launches,
// but in the two-stream float32 pipeline the whole run is 2.3 % SLOWER with it (bench.py tolerance_path 1894 -> 1937 ms per batch,
// gpurun_out/s2): like the 16-bit loader-wave kernels it owns its CU (144 KB of LDS), and a workgroup of the other stream on the same
// CU was already hiding what the loader waves hide.  Kept for single-stream users and as the measured answer
// to "take the operand split out of the float32 main loop".
int g_split_lc_mode = [] { const char* e = getenv("GMD_SPLIT_LC"); return (e && e[0] == '1') ? -1 : 0; }();

template <bool CONV, int TN>
hipError_t launch_split_lc(const GemmParams& p, int gz, hipStream_t s) {
    constexpr int BM = 128, BN = 2 * TN * 16;
    constexpr size_t smem = (size_t)4 * (BM + BN) * 128;
    hipError_t e = opt_in_lds(reinterpret_cast<const void*>(&gemm_split_lc_kernel<CONV, TN>), (int)smem);
    if (e != hipSuccess) return e;
    dim3 grid(((p.N + BN - 1) / BN) * ((p.M + BM - 1) / BM), 1, gz);
    gemm_split_lc_kernel<CONV, TN><<<grid, 512, smem, s>>>(p);
    return hipGetLastError();
}

// the loader / converter kernel runs ONE workgroup per CU: taken when the launch's 128-row tiles come in (nearly) whole rounds of 256
bool split_lc_fits(const SplitPlan& pl, const GemmParams& p, int batch) {
    if (g_split_lc_mode == 0 || batch != 1 || pl.bm != 128 || p.K / BKS < 4) return false;
    if (g_split_lc_mode == 1) return true;
    // measured (tools/ab_split_lc.py, bit-identical results): +3...+9 % where the launch is ONE round of workgroups (conv 8x32x32
    // 640->640 208 -> 196 us, 4x64x64 320->320 110 -> 102 us, linear M=8192 N=640 K=2560 101 -> 95 us); launches of several rounds lose
    // 10-28 % to two co-resident workgroups of the ring kernel, which overlap one tile's epilogue with the next one's prologue
    const int64_t tiles = (int64_t)((p.M + 127) / 128) * ((p.N + pl.bn - 1) / pl.bn) * (pl.ksplit > 1 ? pl.ksplit : 1);
    return tiles >= 200 && tiles <= 256;
}

template <bool CONV, bool WSPLIT, bool ASPLIT = false>
int launch_split_any(GemmParams p, int batch, void* ws, int64_t ws_bytes, hipStream_t s, const char* name) {
    static_assert(!ASPLIT || WSPLIT, "a pre-split activation meets pre-split weights only (GMD_F32SA)");
    SplitPlan pl = make_split_plan(p.M, p.N, p.K, batch, ws ? ws_bytes : 0);
    if (p.act == GMD_ACT_GEGLU) {
        // value / gate tile pairs inside a wave: even TN (128x128 or 64x64 tiles), unsplit K, every tile full, row epilogue
        if (pl.bn == 160) pl.bn = 128;
        pl.ksplit = 1;
    }
    if (p.colstats) {
        const bool rows_ok = p.act != GMD_ACT_GEGLU && (p.ldc & 3) == 0 && (p.residual == nullptr || (p.ldr & 3) == 0) &&
                             (p.rowbias == nullptr || ((p.ldrb & 3) == 0 && (reinterpret_cast<uintptr_t>(p.rowbias) & 15) == 0));
        if (!rows_ok || !gmd_split_colstats_ok(p.M, p.N, p.K, batch, ws ? ws_bytes : 0, p.cs_bucket)) {
            gmd_set_error("%s: this float32 launch cannot emit column statistics (full-tile row epilogue of an unsplit 128-row launch: ask "
                          "gmd_gemm_colstats_plan first)", name);
            return GMD_ERR_UNSUPPORTED;
        }
    }
    if (p.vt_out && !(pl.bm == 128 && pl.ksplit == 1 && batch == 1 && p.M % 128 == 0 && p.N % pl.bn == 0 && p.vt_col0 % pl.bn == 0 &&
                      p.vt_tokens % 64 == 0 && (p.ldc & 3) == 0)) {
        gmd_set_error("%s: this float32 launch cannot write transposed V tiles (ask gmd_gemm_qkv_vt_ok first)", name);
        return GMD_ERR_UNSUPPORTED;
    }
    if (p.c_split) {  // pre-split output: only the full-tile row epilogues write it (gmd_split_out_ok + their alignment conditions)
        const bool rows_ok = pl.bm == 128 && pl.ksplit == 1 && batch == 1 && p.M % 128 == 0 && p.N % pl.bn == 0 && (p.ldc & 3) == 0 &&
                             (p.residual == nullptr || (p.ldr & 3) == 0) &&
                             (p.rowbias == nullptr || ((p.ldrb & 3) == 0 && (reinterpret_cast<uintptr_t>(p.rowbias) & 15) == 0));
        if (!rows_ok) {
            gmd_set_error("%s: this float32 launch cannot store its output pre-split (ask gmd_gemm_out_split_ok first)", name);
            return GMD_ERR_UNSUPPORTED;
        }
    }
    p.ksplit = pl.ksplit;
    p.ws = (float*)ws;
    const int gz = pl.ksplit > 1 ? pl.ksplit : batch;
    hipError_t e;
    if (WSPLIT && !ASPLIT && split_lc_fits(pl, p, batch)) e = pl.bn == 160 ? launch_split_lc<CONV, 5>(p, gz, s) : launch_split_lc<CONV, 4>(p, gz, s);
    else if (pl.bm == 128 && pl.bn == 160) e = launch_split<CONV, WSPLIT, 2, 2, 4, 5, 2, ASPLIT>(p, gz, s);
    else if (pl.bm == 128) e = launch_split<CONV, WSPLIT, 2, 2, 4, 4, 2, ASPLIT>(p, gz, s);
    else e = launch_split<CONV, WSPLIT, 2, 2, 2, 2, 2, ASPLIT>(p, gz, s);
    if (e == hipSuccess && pl.ksplit > 1 && !p.defer_reduce) {
        const int64_t total = (int64_t)p.M * ((p.N + 3) / 4);
        int64_t g = (total + 255) / 256;
        if (g > 4096) g = 4096;
        splitk_reduce_f32_kernel<<<(int)g, 256, 0, s>>>(p);
        e = hipGetLastError();
    }
    if (e != hipSuccess) {
        gmd_set_error("%s: launch failed: %s", name, hipGetErrorString(e));
        return GMD_ERR_LAUNCH;
    }
    return GMD_OK;
}

// channel block of the conv3x3 K order (conv_channel_block of gemm.hip with 4-byte elements and 32-channel steps)
int split_channel_block(int B, int Hin, int Win, int Cin, int Cout) {
    const int64_t rows_total = (int64_t)B * Hin * Win;
    const int tiles_n = (Cout + 159) / 160;
    const int64_t rows_resident = (int64_t)(64 / tiles_n > 0 ? 64 / tiles_n : 1) * 128;
    const int64_t rows = rows_total / 8 < rows_resident ? (rows_total + 7) / 8 : rows_resident;
    const int64_t budget = 3ll << 20;
    if (rows * Cin * 4 <= budget) return Cin;
    int best = 32;
    for (int d = 32; d < Cin; d += 32)
        if (Cin % d == 0 && rows * d * 4 <= budget) best = d;
    return best;
}

}  // namespace

// producer column statistics (GemmParams::colstats) come out of the full-tile row epilogue of an unsplit launch of the 128-row
// kernels, whose waves own 64 rows x (BN/2) columns -- a whole number of buckets: the float32 twin of colstats_plan_ok (gemm.hip)
int gmd_split_colstats_ok(int M, int N, int K, int batch, int64_t ws_bytes, int bucket) {
    if (M <= 0 || N <= 0 || K <= 0 || K % 32 || batch != 1 || bucket <= 0) return 0;
    const SplitPlan pl = make_split_plan(M, N, K, batch, ws_bytes);
    return (pl.bm == 128 && pl.ksplit == 1 && M % 128 == 0 && N % pl.bn == 0 && (pl.bn / 2) % bucket == 0) ? 1 : 0;
}

// the launch can store its result pre-split (GemmParams::c_split): an unsplit launch of full 128-row tiles, whose row epilogues write
// whole 4-element pieces of 32-element chunks (wave tiles start at multiples of 16 columns; GEGLU: of 8 output columns -- 64 / 80 wide)
int gmd_split_out_ok(int M, int N, int K, int geglu, int64_t ws_bytes) {
    if (M <= 0 || N <= 0 || K <= 0 || K % 32) return 0;
    SplitPlan pl = make_split_plan(M, N, K, 1, ws_bytes);
    if (geglu) { if (pl.bn == 160) pl.bn = 128; pl.ksplit = 1; }
    return (pl.bm == 128 && pl.ksplit == 1 && M % 128 == 0 && N % pl.bn == 0) ? 1 : 0;
}

// fused Q|K|V projection with transposed V tiles on the float32 matrix-core path (GemmParams::vt_out)
int gmd_split_qkv_vt_ok(int M, int N, int K, int vt_col0, int vt_tokens, int64_t ws_bytes) {
    if (M <= 0 || N <= 0 || K <= 0 || K % 32 || vt_col0 <= 0 || vt_col0 >= N || vt_tokens <= 0 || vt_tokens % 64 || M % vt_tokens) return 0;
    const SplitPlan pl = make_split_plan(M, N, K, 1, ws_bytes);
    return (pl.bm == 128 && pl.ksplit == 1 && M % 128 == 0 && N % pl.bn == 0 && vt_col0 % pl.bn == 0) ? 1 : 0;
}

// split-K factor launch_split_any will choose (no GEGLU): gmd_conv3x3_groupnorm / gmd_conv3x3_gn_fusable of gemm.hip
int gmd_split_plan_ksplit(int M, int N, int K, int64_t ws_bytes) { return make_split_plan(M, N, K, 1, ws_bytes).ksplit; }

void gmd_split_set_lc(int mode) { g_split_lc_mode = mode; }

// called by gmd_gemm_nt / gmd_conv3x3 (gemm.hip) for the two split dtype codes; the parameter block is validated there
// presplit: 0 = both operands plain float32 (GMD_F32S), 1 = W pre-split (GMD_F32SW), 2 = A and W pre-split (GMD_F32SA)
int gmd_launch_split_gemm(const void* params, int presplit, int batch, void* ws, int64_t ws_bytes, hipStream_t s, const char* name) {
    const GemmParams& p = *reinterpret_cast<const GemmParams*>(params);
    if (presplit == 2) return launch_split_any<false, true, true>(p, batch, ws, ws_bytes, s, name);
    return presplit ? launch_split_any<false, true>(p, batch, ws, ws_bytes, s, name) : launch_split_any<false, false>(p, batch, ws, ws_bytes, s, name);
}
int gmd_launch_split_conv(const void* params, int presplit, int B, void* ws, int64_t ws_bytes, hipStream_t s, const char* name) {
    GemmParams p = *reinterpret_cast<const GemmParams*>(params);
    p.cblk = split_channel_block(B, p.Hin, p.Win, p.Cin, p.N);
    if (presplit == 2) return launch_split_any<true, true, true>(p, 1, ws, ws_bytes, s, name);
    return presplit ? launch_split_any<true, true>(p, 1, ws, ws_bytes, s, name) : launch_split_any<true, false>(p, 1, ws, ws_bytes, s, name);
}

extern "C" int gmd_split_weights(const float* W, void* out, int64_t N, int64_t K, int64_t ldw, gmd_stream_t stream) {
    GMD_REQUIRE(N >= 0 && K >= 0 && K % 32 == 0 && ldw >= K && ldw % 4 == 0, "gmd_split_weights: K=%lld must be a multiple of 32, ldw >= K a multiple of 4",
                (long long)K);
    if (N == 0 || K == 0) return GMD_OK;
    GMD_REQUIRE(W && out && gmd_aligned16(W) && gmd_aligned16(out), "gmd_split_weights: null or unaligned pointer");
    const int64_t total = N * (K / 32) * 4;
    int64_t g = (total + 255) / 256;
    if (g > 8192) g = 8192;
    split_weights_kernel<<<(int)g, 256, 0, (hipStream_t)stream>>>(W, (unsigned*)out, N, K, ldw);
    GMD_CHECK_LAUNCH("gmd_split_weights");
    return GMD_OK;
}

// Dense contractions of the UNet / VAE: nn.Linear, conv1x1 (plain NT GEMM) and conv3x3 as an
// implicit GEMM over channels-last activations, with the bias / time-embedding / residual /
// activation epilogue fused.
//
//   C[m, n] = act(alpha * sum_k A[m, k] * W[n, k] + bias[n] + rowbias[m / rpg, n] + residual[m, n])
//
// Two kernels share one parameter block and one A-operand addressing scheme:
//   * gemm_bf16_kernel : bf16 inputs, fp32 accumulate on the matrix cores
//     (__builtin_amdgcn_mfma_f32_16x16x32_bf16), BMxBNx64 tiles, 4 waves (2x2), double-buffered
//     LDS filled through registers (global loads for tile k+1 are issued before the MFMAs of tile k
//     and written to LDS after them), XOR-swizzled 128-byte LDS rows so that the ds_read_b128
//     fragment reads are bank-conflict free, accumulators staged through LDS for a fully coalesced
//     (16 bytes/lane) fused epilogue.
//   * gemm_f32_kernel : exact float32 FMA path used for parity runs (64x64x16 tiles, 4x4 per thread).
//
// For conv3x3 the GEMM row m is the output pixel (b, oy, ox), k = (ky*3+kx)*Cin + c; the A tile is
// gathered on the fly (zero padding, stride 2, fused nearest-2x upsample) -- no im2col buffer.
#include "gemm_shared.h"
#include <stdlib.h>
#include <type_traits>

// gemm_split.hip: float32 on the matrix cores as three float16 products (GMD_F32S / GMD_F32SW)
int gmd_launch_split_gemm(const void* params, int presplit, int batch, void* ws, int64_t ws_bytes, hipStream_t s, const char* name);
int gmd_launch_split_conv(const void* params, int presplit, int B, void* ws, int64_t ws_bytes, hipStream_t s, const char* name);
int gmd_split_plan_ksplit(int M, int N, int K, int64_t ws_bytes);
int gmd_split_colstats_ok(int M, int N, int K, int batch, int64_t ws_bytes, int bucket);
int gmd_split_out_ok(int M, int N, int K, int geglu, int64_t ws_bytes);
int gmd_split_qkv_vt_ok(int M, int N, int K, int vt_col0, int vt_tokens, int64_t ws_bytes);
void gmd_split_set_lc(int mode);

namespace {

// Fused epilogue for 8 consecutive columns n..n+7 of row m (bf16 activations): alpha, bias, per-group row
// bias, residual, activation, then a 16-byte store (scalar stores on ragged / unaligned edges).
template <typename HT>
__device__ __forceinline__ void epilogue_store8(const GemmParams& p, int z, int m, int n, float (&v)[8]) {
    const bool vec_ok = (p.ldc % 8 == 0) && (p.sC % 8 == 0) && (p.residual == nullptr || (p.ldr % 8 == 0 && p.sR % 8 == 0));
    const int nvalid = p.N - n < 8 ? p.N - n : 8;
    const float* rb = p.rowbias ? p.rowbias + (int64_t)(m / p.rows_per_group) * p.ldrb : nullptr;
    const HT* res = p.residual ? (const HT*)p.residual + (int64_t)z * p.sR + (int64_t)m * p.ldr + n : nullptr;
    float rv[8];
    if (res) {
        if (nvalid == 8 && vec_ok) load_vec(res, rv);
        else
            for (int j = 0; j < 8; ++j) rv[j] = j < nvalid ? Elem<HT>::ld(res + j) : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (j < nvalid) {
            float x = v[j] * p.alpha;
            if (p.bias) x += p.bias[n + j];
            float add = 0.f;  // (acc*alpha + bias) + (residual + rowbias): the association of epilogue_regs / epilogue_rows
            if (res) add += rv[j];
            if (rb) add += rb[n + j];
            v[j] = apply_act(x + add, p.act);
        }
    }
    const int64_t coff = (int64_t)z * p.sC + (int64_t)m * p.ldc + n;
    if (p.out_f32) {
        float* o = (float*)p.C + coff;
        if (nvalid == 8 && p.ldc % 4 == 0 && p.sC % 4 == 0) {
            *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
            for (int j = 0; j < nvalid; ++j) o[j] = v[j];
        }
    } else {
        HT* o = (HT*)p.C + coff;
        if (nvalid == 8 && vec_ok) store_vec(o, v);
        else
            for (int j = 0; j < nvalid; ++j) Elem<HT>::st(o + j, v[j]);
    }
}

// split-K: sum the fp32 partial slabs ws[s][m][n] in a fixed order, then the fused epilogue
template <typename HT>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmParams p) {
    GMD_WG_TRACE_SCOPE(WGK_SPLITK_REDUCE);
    const int NC = (p.N + 7) / 8;
    const int64_t total = (int64_t)p.M * NC;
    const int64_t slab = (int64_t)p.M * p.N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / NC), n = (int)(i - (int64_t)m * NC) * 8;
        const int nvalid = p.N - n < 8 ? p.N - n : 8;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
        const float* src = p.ws + (int64_t)m * p.N + n;
        for (int s = 0; s < p.ksplit; ++s, src += slab) {
            if (nvalid == 8 && p.N % 4 == 0) {
                const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
                v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
            } else {
                for (int j = 0; j < nvalid; ++j) v[j] += src[j];
            }
        }
        epilogue_store8<HT>(p, 0, m, n, v);
    }
}

// ------------------------------------------------------------------------------------------------
// bf16 MFMA kernel
// ------------------------------------------------------------------------------------------------
// Epilogue straight from registers.  The MFMA operands are swapped (W fragment as A, activation fragment as B), so
// by the C/D layout (col = lane&15, row = 4*(lane>>4)+reg) a lane holds FOUR CONSECUTIVE output columns n of ONE
// row m: bias/row-bias/residual/activation are applied in registers and each lane stores 8 bytes (bf16) or 16 bytes
// (fp32 / split-K partials) -- no LDS round trip, no barrier.  mw/nw: first row/column of this wave's tile.
template <typename HT, int TM, int TN>
__device__ __forceinline__ void epilogue_regs(const GemmParams& p, const f32x4 (&acc)[TM][TN], int mw, int nw, int frow, int fq,
                                              int z, int ks) {
    const bool vec_c = (p.ldc % 4 == 0) && (p.sC % 4 == 0);
    const bool vec_r = p.residual != nullptr && (p.ldr % 4 == 0) && (p.sR % 4 == 0);
    const bool vec_n = (p.N % 4 == 0);
    const bool vec_rb = vec_n && (p.ldrb % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.rowbias) & 15) == 0);
    float bz[TN][4];  // per-column bias of this lane's TN column groups, loaded once (16-byte loads)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = nw + j * 16 + fq * 4;
        bz[j][0] = bz[j][1] = bz[j][2] = bz[j][3] = 0.f;
        if (p.bias && p.ksplit <= 1 && n < p.N) {
            if (n + 4 <= p.N && vec_n) {
                const float4 t = *reinterpret_cast<const float4*>(p.bias + n);
                bz[j][0] = t.x; bz[j][1] = t.y; bz[j][2] = t.z; bz[j][3] = t.w;
            } else {
                for (int e = 0; e < 4; ++e) if (n + e < p.N) bz[j][e] = p.bias[n + e];
            }
        }
    }
    if constexpr (TN % 2 == 0) {
        if (p.act == GMD_ACT_GEGLU) {
            // W rows are interleaved in 16-row groups [value | gate] (host re-layout): tile j holds 16 value columns and
            // tile j+1 the matching gate columns IN THE SAME LANE, so h * gelu_erf(g) is formed in registers and only the
            // [M, N/2] product is written (GEGLU.forward of diffusers; N % 32 == 0 checked on the host).
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = mw + i * 16 + frow;
                if (m >= p.M) continue;
#pragma unroll
                for (int j = 0; j < TN; j += 2) {
                    const int n = nw + j * 16 + fq * 4;  // interleaved column of the value group
                    if (n >= p.N) continue;
                    float o4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float h = acc[i][j][e] * p.alpha + bz[j][e];
                        const float g = acc[i][j + 1][e] * p.alpha + bz[j + 1][e];
                        o4[e] = h * (0.5f * g * (1.0f + fast_erf(g * 0.70710678118654752440f)));
                    }
                    const int no = (nw + j * 16) / 2 + fq * 4;  // output column in [0, N/2)
                    HT* o = (HT*)p.C + (int64_t)z * p.sC + (int64_t)m * p.ldc + no;
                    *reinterpret_cast<uint2*>(o) = make_uint2(Half<HT>::pack2(o4[0], o4[1]), Half<HT>::pack2(o4[2], o4[3]));
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
            if (p.ksplit > 1) {  // raw partial sums; the epilogue runs in splitk_reduce_kernel
                float* o = p.ws + ((int64_t)ks * p.M + m) * p.N + n;
                if (nvalid == 4 && vec_n) *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
                else
                    for (int e = 0; e < nvalid; ++e) o[e] = v[e];
                continue;
            }
            float rv[4] = {0.f, 0.f, 0.f, 0.f};
            if (p.residual) {
                const HT* res = (const HT*)p.residual + (int64_t)z * p.sR + (int64_t)m * p.ldr + n;
                if (nvalid == 4 && vec_r) {
                    const uint2 w = *reinterpret_cast<const uint2*>(res);
                    Half<HT>::unpack2(w.x, rv[0], rv[1]);
                    Half<HT>::unpack2(w.y, rv[2], rv[3]);
                } else {
                    for (int e = 0; e < nvalid; ++e) rv[e] = Elem<HT>::ld(res + e);
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
            for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e] * p.alpha + bz[j][e] + rv[e], p.act);
            const int64_t coff = (int64_t)z * p.sC + (int64_t)m * p.ldc + n;
            if (p.out_f32) {
                float* o = (float*)p.C + coff;
                if (nvalid == 4 && vec_c) *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
                else
                    for (int e = 0; e < nvalid; ++e) o[e] = v[e];
            } else {
                HT* o = (HT*)p.C + coff;
                if (nvalid == 4 && vec_c) *reinterpret_cast<uint2*>(o) = make_uint2(Half<HT>::pack2(v[0], v[1]), Half<HT>::pack2(v[2], v[3]));
                else
                    for (int e = 0; e < nvalid; ++e) Elem<HT>::st(o + e, v[e]);
            }
        }
    }
}

// Row-contiguous epilogue through LDS (full bf16 tiles without split-K / GEGLU).  The register epilogue above writes, per
// store instruction, 16 rows x 32 bytes; the store path is transaction-bound (a float32 output, twice the bytes in the same
// number of instructions, takes exactly the same time), which makes the small-K linear layers epilogue-bound.  Here each
// wave parks its raw accumulators in a private LDS strip (two halves of 32 rows; the K-loop stages are dead by then),
// reads them back row-major and issues 16-byte stores that cover whole 128/160-byte row segments; bias, row bias,
// residual (16-byte loads, same shape) and activation are applied on the way out in float32, bit-identical to the
// register epilogue.  Caller guarantees: block tile fully inside M x N, ldc / ldr / batch strides multiples of 8, and a
// __syncthreads() between the last LDS read of the K loop and this call.
// The V column tiles of a fused Q|K|V projection (GemmParams::vt_out): the wave tile leaves TRANSPOSED -- lane = column, 32 rows
// (tokens) per pass through the strip, packed to 64 contiguous bytes of vt_out[sample][column][token ...]: the layout the attention
// kernels consume, written by the projection itself instead of by a second (batched, transposed) GEMM launch per attention.
// alpha and bias as epilogue_rows; no residual / row bias / activation (host-checked).  Wave tiles never straddle a sample
// (vt_tokens is a multiple of the wave tile's rows: host-checked).
template <typename HT, int TM, int TN>
__device__ __forceinline__ void epilogue_cols_vt(const GemmParams& p, const f32x4 (&acc)[TM][TN], float* strip, int mw, int nw, int lane) {
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
            const int cc = lane + 64 * k < NCOL ? lane + 64 * k : NCOL - 1;  // every lane reads (the idle ones the last column): no EXEC change
            const float bz = p.bias ? p.bias[nw + cc] : 0.f;
            unsigned pk[16];
#pragma unroll
            for (int r = 0; r < 32; r += 2)
                pk[r / 2] = Half<HT>::pack2(strip[r * ROWF + cc] * p.alpha + bz, strip[(r + 1) * ROWF + cc] * p.alpha + bz);
            if (lane + 64 * k < NCOL) {
                HT* o = (HT*)p.vt_out + ((int64_t)b * (p.N - p.vt_col0) + (nw + cc - p.vt_col0)) * p.vt_ld + tok0 + h * 32;
#pragma unroll
                for (int q = 0; q < 4; ++q) *reinterpret_cast<uint4*>(o + 8 * q) = make_uint4(pk[4 * q], pk[4 * q + 1], pk[4 * q + 2], pk[4 * q + 3]);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// (PLAIN: the launch has no activation -- every projection and convolution of the UNet but the time-embedding MLP.  A compile-time
// fact for the item loop: with the activation chosen by a run-time switch per element, the SiLU and quick-GELU bodies (exp, IEEE
// division) sat in all ten unrolled items of a wave tile -- ~600 instructions, 3.6 KB of code per item -- and although their
// branches were never taken, FETCHING that code once per wave tile cost ~0.55 us per item: in-kernel stamps put 5.5 of the 7.2 us
// epilogue of a K = 320 projection there, with the stores, the bias and the residual all ruled out one by one (round 5).)
template <typename HT, int TM, int TN, bool PLAIN>
__device__ __forceinline__ void epilogue_rows_impl(const GemmParams& p, const f32x4 (&acc)[TM][TN], float* strip, int mw, int nw, int lane,
                                                   int z) {
    static_assert(TM % 2 == 0, "halves of two 16-row tiles");
    if (p.vt_out && nw >= p.vt_col0) {  // wave-uniform: a V column tile of a fused Q|K|V projection
        epilogue_cols_vt<HT, TM, TN>(p, acc, strip, mw, nw, lane);
        return;
    }
    constexpr int NCOL = TN * 16, ROWF = NCOL + 4, CH = NCOL / 8;
    constexpr int ITER = (32 * CH + 63) / 64;
    const int frow = lane & 15, fq = lane >> 4;
    // row-bias group (sample) of the wave tile's first row: ONE division per wave instead of one per stored item; a group
    // of at least 64 rows can change at most once inside the 64-row wave tile
    int g0 = 0, grem = 0;
    if (p.rowbias) {
        g0 = mw / p.rows_per_group;
        grem = mw - g0 * p.rows_per_group;
    }
    float cs[(NCOL + 63) / 64] = {}, cq[(NCOL + 63) / 64] = {};  // column sums of this wave tile (p.colstats only)
    // Round 5: ALL of the wave tile's residual rows are requested up front (TM / 2 x ITER 16-byte loads per lane: 40 registers at
    // TN = 5, free now that the fragments are dead), so their L2 / HBM latency runs under the staging of the accumulators instead of
    // being exposed once per 32-row half behind the wave barrier (the short-K projections are prologue / epilogue-bound).
#if defined(GMD_PP_DIAG) && defined(GMD_WG_TRACE)
    unsigned long long et[6] = {0, 0, 0, 0, 0, 0};
    int eti = 0;
#define EPI_STAMP() { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); if (eti < 6) et[eti++] = t_; }
#else
#define EPI_STAMP()
#endif
    EPI_STAMP()
    static_assert(NCOL / 4 <= 32, "one bias float4 per strip row");
    // Round 5: the residual rows (or, for a launch with a row bias and no residual, the row bias) of a 32-row half are requested BEFORE
    // that half's accumulators are staged, so their L2 / HBM latency runs under the LDS writes and the wave barrier instead of being
    // exposed item by item behind the previous item's store.  (One half at a time, two 16-byte values per item: both halves up front
    // spill in the 256 x 160 kernels once the item loop is a single basic block.)
    uint4 resv[2][ITER];
    // Round 5: the bias of this wave's columns goes into the strip's PAD columns once (row j holds float4 j of the bias: the four
    // floats behind the NCOL data columns of a row are never written by the staging below).  Loaded inside the item loop it cost
    // ONE exposed global-load round trip per item -- the loads cannot move above the previous item's store (they may alias for all
    // the compiler knows) -- ten serialised ~0.6 us trips per wave tile: in-kernel stamps put 6.2 of the 7.6 us epilogue of a
    // K = 320 projection there (tools/dbg/pp_phase_probe.py).
    if (lane < NCOL / 4)
        *reinterpret_cast<float4*>(strip + lane * ROWF + NCOL) = p.bias ? *reinterpret_cast<const float4*>(p.bias + nw + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const bool rb_pre = PLAIN && p.rowbias != nullptr && p.residual == nullptr;  // row bias alone (conv1 of a ResnetBlock2D): prefetched per half
    // (residual rows of half hh -> resv[hh & 1]: half 0 here, half h + 1 below, behind the staging of half h -- its accumulators are
    //  dead by then, so the 20 registers are free -- and in front of half h's items, which hide its L2 / HBM latency; requested at the
    //  top of its own half, as before, the second half's latency was exposed once more per wave tile)
    auto request_residual = [&](int hh) {
#pragma unroll
        for (int t = 0; t < ITER; ++t) {
            const int idx = lane + 64 * t;
            const int r = idx / CH, c = idx - r * CH;
            if (32 * CH % 64 != 0 && r >= 32) continue;
            resv[hh & 1][t] = *reinterpret_cast<const uint4*>((const HT*)p.residual + (int64_t)z * p.sR + (int64_t)(mw + hh * 32 + r) * p.ldr + nw + c * 8);
        }
    };
    const bool res_pre = PLAIN && p.residual != nullptr;  // (the activation instances -- time-embedding MLP, text encoder: tiny launches -- load in place)
    if (res_pre) request_residual(0);
#pragma unroll
    for (int h = 0; h < TM / 2; ++h) {
        if (rb_pre) {
#pragma unroll
            for (int t = 0; t < ITER; ++t) {
                const int idx = lane + 64 * t;
                const int r = idx / CH, c = idx - r * CH;
                if (32 * CH % 64 != 0 && r >= 32) continue;
                const int ro = h * 32 + r;
                const int grp = p.rows_per_group >= 64 ? g0 + (grem + ro >= p.rows_per_group ? 1 : 0) : (mw + ro) / p.rows_per_group;
                const float* rb = p.rowbias + (int64_t)grp * p.ldrb + nw + c * 8;
                resv[0][t] = __builtin_bit_cast(uint4, *reinterpret_cast<const float4*>(rb));
                resv[1][t] = __builtin_bit_cast(uint4, *reinterpret_cast<const float4*>(rb + 4));
            }
        }
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                *reinterpret_cast<float4*>(strip + (i2 * 16 + frow) * ROWF + j * 16 + fq * 4) =
                    make_float4(acc[2 * h + i2][j][0], acc[2 * h + i2][j][1], acc[2 * h + i2][j][2], acc[2 * h + i2][j][3]);
        __builtin_amdgcn_wave_barrier();  // same wave, in-order LDS: only the compiler must not reorder across this point
        EPI_STAMP()
        if (res_pre && h + 1 < TM / 2) request_residual(h + 1);
#pragma unroll
        for (int t = 0; t < ITER; ++t) {
            const int idx = lane + 64 * t;
            const int r = idx / CH, c = idx - r * CH;
            if (32 * CH % 64 != 0 && r >= 32) continue;
            const int m = mw + h * 32 + r, n = nw + c * 8;
            const float4 a0 = *reinterpret_cast<const float4*>(strip + r * ROWF + c * 8);
            const float4 a1 = *reinterpret_cast<const float4*>(strip + r * ROWF + c * 8 + 4);
            float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            float add[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (p.residual) {
                const uint4 w = PLAIN ? resv[h & 1][t] : *reinterpret_cast<const uint4*>((const HT*)p.residual + (int64_t)z * p.sR + (int64_t)m * p.ldr + n);
                const unsigned ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) Half<HT>::unpack2(ww[e], add[2 * e], add[2 * e + 1]);
            }
            if (rb_pre) {
                const float4 t0 = __builtin_bit_cast(float4, resv[0][t]), t1 = __builtin_bit_cast(float4, resv[1][t]);
                add[0] += t0.x; add[1] += t0.y; add[2] += t0.z; add[3] += t0.w; add[4] += t1.x; add[5] += t1.y; add[6] += t1.z; add[7] += t1.w;
            } else if (p.rowbias) {  // row bias AND residual (never in the UNet): loaded in place
                const int ro = h * 32 + r;  // row offset inside the wave tile
                const int grp = p.rows_per_group >= 64 ? g0 + (grem + ro >= p.rows_per_group ? 1 : 0) : m / p.rows_per_group;
                const float* rb = p.rowbias + (int64_t)grp * p.ldrb + n;
                const float4 t0 = *reinterpret_cast<const float4*>(rb), t1 = *reinterpret_cast<const float4*>(rb + 4);
                add[0] += t0.x; add[1] += t0.y; add[2] += t0.z; add[3] += t0.w; add[4] += t1.x; add[5] += t1.y; add[6] += t1.z; add[7] += t1.w;
            }
            const float4 b0 = *reinterpret_cast<const float4*>(strip + (2 * c) * ROWF + NCOL), b1 = *reinterpret_cast<const float4*>(strip + (2 * c + 1) * ROWF + NCOL);
            const float bz[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = apply_act(v[e] * p.alpha + bz[e] + add[e], PLAIN ? GMD_ACT_NONE : p.act);  // same association as epilogue_regs
            HT* o = (HT*)p.C + (int64_t)z * p.sC + (int64_t)m * p.ldc + n;
            const uint4 packed = make_uint4(Half<HT>::pack2(v[0], v[1]), Half<HT>::pack2(v[2], v[3]), Half<HT>::pack2(v[4], v[5]), Half<HT>::pack2(v[6], v[7]));
#if defined(GMD_EPI_NOSTORE)  // timing experiment only (tools/dbg): the epilogue without its global stores (results are wrong)
            if (packed.x == 0x12345678u && packed.y == 0x9abcdef0u) *reinterpret_cast<uint4*>(o) = packed;
#elif defined(GMD_EPI_NT)     // timing experiment only: non-temporal stores
            __builtin_nontemporal_store(__builtin_bit_cast(u32x4, packed), reinterpret_cast<u32x4*>(o));
#else
            *reinterpret_cast<uint4*>(o) = packed;
#endif
            // (one item at a time: without this fence the compact PLAIN body lets the scheduler pull every item's strip reads to the
            //  front -- 70 spilled registers in the 256 x 160 kernels, one workgroup per CU instead of two in the ring kernel)
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (p.colstats) {  // the values as stored (rounded) go back to the strip for the column pass below
                const unsigned pw[4] = {packed.x, packed.y, packed.z, packed.w};
                float rv[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) Half<HT>::unpack2(pw[e], rv[2 * e], rv[2 * e + 1]);
                *reinterpret_cast<float4*>(strip + r * ROWF + c * 8) = make_float4(rv[0], rv[1], rv[2], rv[3]);
                *reinterpret_cast<float4*>(strip + r * ROWF + c * 8 + 4) = make_float4(rv[4], rv[5], rv[6], rv[7]);
            }
        }
        __builtin_amdgcn_wave_barrier();
        EPI_STAMP()
        if (p.colstats) {  // lane -> column (and column 64 + lane): 32 conflict-free reads each, fixed order (gemm_shared.h)
            colstats_pass<NCOL, ROWF>(strip, lane, cs, cq);
            __builtin_amdgcn_wave_barrier();
        }
    }
#if defined(GMD_PP_DIAG) && defined(GMD_WG_TRACE)
    if (threadIdx.x == 0 && g_wg_trace != nullptr) {  // epilogue split of wave 0 (wid field 0xFE): start | staged h0 | items h0 | staged h1 | items h1
        GmdWgTraceHeader* hh = g_wg_trace;
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        const unsigned shard = ((((xcc & 7u) * 8u + ((hw >> 13) & 7u)) * 2u + ((hw >> 12) & 1u)) * 16u) + ((hw >> 8) & 15u);
        unsigned long long* counters = reinterpret_cast<unsigned long long*>(hh + 1);
        const unsigned long long i2 = __hip_atomic_fetch_add(counters + 16ull * shard, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (i2 < hh->capacity) {
            uint4* rec = reinterpret_cast<uint4*>(counters + 16ull * hh->shards) + 2ull * (shard * hh->capacity + i2);
            rec[0] = make_uint4((unsigned)(et[1] - et[0]), (unsigned)(et[2] - et[1]), (unsigned)(et[3] - et[2]), (unsigned)(et[4] - et[3]));
            rec[1] = make_uint4(hw, (xcc & 15u) | ((unsigned)(WGK_PP | 0x80) << 8) | (0xFEu << 16), (unsigned)TN << 16, p.residual ? 1u : 0u);
        }
    }
#endif
#undef EPI_STAMP
    // producer statistics for a following GroupNorm: {sum, sum of squares} of the STORED values over this wave tile's TM*16 = 64
    // rows (one row block of the statistics) and each bucket of cs_bucket adjacent columns
    if (p.colstats)
        colstats_store<NCOL, ROWF>(strip, lane, cs, cq, p.cs_bucket,
                                   p.colstats + ((int64_t)(mw / (TM * 16)) * (p.N / p.cs_bucket) + nw / p.cs_bucket) * 2);
}

template <typename HT, int TM, int TN>
__device__ __forceinline__ void epilogue_rows(const GemmParams& p, const f32x4 (&acc)[TM][TN], float* strip, int mw, int nw, int lane, int z) {
    if (__builtin_expect(p.act == GMD_ACT_NONE, 1)) epilogue_rows_impl<HT, TM, TN, true>(p, acc, strip, mw, nw, lane, z);  // (hot: laid out next to the K loop)
    else epilogue_rows_impl<HT, TM, TN, false>(p, acc, strip, mw, nw, lane, z);
}

// Split-K variant of epilogue_rows: the raw float32 partial sums of a full tile go to this slice's slab as whole row
// segments (16-byte stores, 64 * TN bytes contiguous per row) instead of 64-byte pieces of 16 rows per instruction.
template <int TM, int TN>
__device__ __forceinline__ void epilogue_rows_slab(const GemmParams& p, const f32x4 (&acc)[TM][TN], float* strip, int mw, int nw, int lane,
                                                   int ks) {
    static_assert(TM == 4, "two halves of two 16-row tiles");
    constexpr int NCOL = TN * 16, ROWF = NCOL + 4, CH = NCOL / 4;
    constexpr int ITER = (32 * CH + 63) / 64;
    const int frow = lane & 15, fq = lane >> 4;
    float* slab = p.ws + (int64_t)ks * p.M * p.N;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
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
            *reinterpret_cast<float4*>(slab + (int64_t)(mw + h * 32 + r) * p.N + nw + c * 4) =
                *reinterpret_cast<const float4*>(strip + r * ROWF + c * 4);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// In-kernel split-K reduction (round 5).  Grid z = K slice, and workgroups are dispatched in increasing linear id (x fastest, z
// slowest), so every workgroup of the slices 0 .. ksplit-2 of ANY tile has been placed on a CU before the first workgroup of the
// last slice is: the last slice -- the only one that waits -- can never hold a CU that a workgroup it waits for still needs, whatever
// another stream's kernels occupy (the same argument the vendor library's stream-K fix-up rests on).  The spin is bounded all the same.
//   slices 0 .. ksplit-2 ("producers"): every consumer wave stores its TM x TN accumulator fragments to the fragment area, lane-
//     linear (one 1 KB store instruction per fragment: whole 128-byte lines by one instruction of one wave), with sc1 stores;
//     s_waitcnt vmcnt(0); workgroup barrier; ONE lane adds 1 to the tile's arrival counter (agent scope); the workgroup ends.
//   slice ksplit-1 ("finisher"): ONE lane polls the counter with sc1 loads until ksplit-1 producers have arrived; workgroup barrier;
//     every wave reads the same fragments back with sc1 loads and forms (s_0 + s_1 + ... + s_{ksplit-2}) + own -- the order of
//     splitk_reduce_kernel, so the result is bit-identical to the slab path -- then the counter is put back to 0 and the caller goes
//     on to the fused epilogue of an unsplit launch.  (Hand-off protocol: MI355X_MICROARCH.md, "sc1 loads in place of the acquire",
//     first row: sc1 stores of 16 bytes, one signalling lane per storing workgroup behind its barrier, sc1 poll, barrier, sc1 loads.)
// Returns false in a producer (the kernel returns), true in the finisher.  NCONS consumer waves = waves 0 .. NCONS-1 of the workgroup;
// the loader waves have ended by now (s_barrier counts the surviving waves only).
template <int TM, int TN, int NCONS>
__device__ __forceinline__ bool splitk_fixup(const GemmParams& p, f32x4 (&acc)[TM][TN], int ks, int wid, int lane) {
    constexpr unsigned FR = TM * TN;  // fragments (16 bytes per lane) per wave
    const unsigned tile = blockIdx.x, ntiles = gridDim.x;
    const int last = p.ksplit - 1;
    unsigned* cnt = p.fix_cnt + tile;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.ws, 0, (int)p.fix_bytes, 0x00020000);
    const unsigned wave_off = ((tile * NCONS + (unsigned)wid) * FR) * 1024u + (unsigned)lane * 16u;
    const unsigned slice_bytes = ntiles * NCONS * FR * 1024u;
    if (ks != last) {
        const unsigned base = (unsigned)ks * slice_bytes + wave_off;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rs, base + (unsigned)(i * TN + j) * 1024u, 0, 16 /* sc1 */);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // every storing wave has waited for its stores
        if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
    }
    if (threadIdx.x == 0) {
        int spins = 0;  // bounded: a grid must drain whatever happens (2^21 polls of >= 0.3 us each; never reached in a healthy run)
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)last && ++spins < (1 << 21)) __builtin_amdgcn_s_sleep(8);
    }
    __syncthreads();
    // two fragment rows (2 x TN 16-byte loads per lane) in flight at a time: the sc1 loads come from the fabric (~1 us each way), so
    // the number of load -> wait round trips is what this costs -- TM / 2 of them per slice; all TM rows at once would spill
    static_assert(TM % 2 == 0, "fragment rows are fetched in pairs");
#pragma unroll
    for (int i = 0; i < TM; i += 2) {
        f32x4 t[2][TN];
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                t[ii][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, wave_off + (unsigned)((i + ii) * TN + j) * 1024u, 0, 16 /* sc1 */));
        for (int s = 1; s < last; ++s) {  // (3 or 4 slices: one row of the pair at a time -- TN more values in flight, not 2 TN)
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                f32x4 u[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    u[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)s * slice_bytes + wave_off + (unsigned)((i + ii) * TN + j) * 1024u, 0, 16));
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    t[ii][j] += u[j];
                    asm volatile("" : "+v"(t[ii][j])::"memory");
                }
            }
        }
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                acc[i + ii][j] = t[ii][j] + acc[i + ii][j];
                // (the sums of this pair of rows exist before any load of the next pair is issued: bounded registers)
                asm volatile("" : "+v"(acc[i + ii][j])::"memory");
            }
    }
    if (threadIdx.x == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // clean for the next launch
    return true;
}

// GEGLU variant of epilogue_rows: h * gelu(g) is formed in registers exactly as in epilogue_regs (value / gate tiles of a
// pair sit in the same lane), staged, and written as [M, N/2] rows with 16-byte stores.
template <typename HT, int TM, int TN>
__device__ __forceinline__ void epilogue_rows_geglu(const GemmParams& p, const f32x4 (&acc)[TM][TN], float* strip, int mw, int nw,
                                                    int lane, int z) {
    static_assert(TM == 4 && TN % 2 == 0, "two halves of two 16-row tiles; value/gate tile pairs");
    constexpr int NCOL = TN * 8, ROWF = NCOL + 4, CH = NCOL / 8;  // output columns of this wave
    constexpr int ITER = (32 * CH + 63) / 64;
    const int frow = lane & 15, fq = lane >> 4;
    float bz[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const float4 t = p.bias ? *reinterpret_cast<const float4*>(p.bias + nw + j * 16 + fq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        bz[j][0] = t.x; bz[j][1] = t.y; bz[j][2] = t.z; bz[j][3] = t.w;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int j = 0; j < TN; j += 2) {
                float o4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float hv = acc[2 * h + i2][j][e] * p.alpha + bz[j][e];
                    const float g = acc[2 * h + i2][j + 1][e] * p.alpha + bz[j + 1][e];
                    o4[e] = hv * (0.5f * g * (1.0f + fast_erf(g * 0.70710678118654752440f)));
                }
                *reinterpret_cast<float4*>(strip + (i2 * 16 + frow) * ROWF + (j / 2) * 16 + fq * 4) = make_float4(o4[0], o4[1], o4[2], o4[3]);
            }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < ITER; ++t) {
            const int idx = lane + 64 * t;
            const int r = idx / CH, c = idx - r * CH;
            if (32 * CH % 64 != 0 && r >= 32) continue;
            const float4 a0 = *reinterpret_cast<const float4*>(strip + r * ROWF + c * 8);
            const float4 a1 = *reinterpret_cast<const float4*>(strip + r * ROWF + c * 8 + 4);
            HT* o = (HT*)p.C + (int64_t)z * p.sC + (int64_t)(mw + h * 32 + r) * p.ldc + nw / 2 + c * 8;
            *reinterpret_cast<uint4*>(o) = make_uint4(Half<HT>::pack2(a0.x, a0.y), Half<HT>::pack2(a0.z, a0.w), Half<HT>::pack2(a1.x, a1.y), Half<HT>::pack2(a1.z, a1.w));
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// PF = register prefetch distance in K steps: the global loads of tile kt+PF are issued before the MFMAs of
// tile kt and are written to LDS at the end of iteration kt+PF-1, i.e. they have PF whole iterations to land.
template <typename HT, bool CONV, int BM, int BN, int PF>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const GemmParams p) {
    GMD_WG_TRACE_SCOPE(WGK_GEMM64 | (CONV ? WGK_CONV_BIT : 0));
    constexpr int NA = BM / 32, NW = BN / 32;  // staging slots per thread
    constexpr int TM = BM / 32, TN = BN / 32;  // 16x16 tiles per wave along M / N (wave tile = BM/2 x BN/2)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int kStage = (BM + BN) * 128;  // bytes per pipeline stage: A tile then W tile

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    // XCD-aware tile order (as in gemm_ring_kernel): workgroup ids with equal id % 8 share an XCD and get a CONTIGUOUS run
    // of tiles.  The fast axis is the one whose operand panel is shared by the run: n-fastest when A (M x K) is the larger
    // operand -- a run then covers whole row blocks and A is fetched once, W once per XCD -- m-fastest otherwise (small-M
    // launches of the 8x8 level: W, the larger operand, is fetched once).  rocprofv3 FETCH_SIZE on M=2048 N=1280 K=1280:
    // 3.9x the algorithmic reads on the plain 2-D grid (profiles/r02_pmc_conv_gemm_traffic.txt).
    int m0, n0;
    {
        const int nwg = gridDim.x, id = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, j = id >> 3;
        const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
        const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
        if (p.M >= p.N) {
            const int mt = L / tiles_n;
            m0 = mt * BM;
            n0 = (L - mt * tiles_n) * BN;
        } else {
            const int nt = L / tiles_m;
            n0 = nt * BN;
            m0 = (L - nt * tiles_m) * BM;
        }
    }
    // grid z = batch index (plain batched GEMM) or K-split index (p.ksplit > 1, batch == 1)
    const int z = p.ksplit > 1 ? 0 : blockIdx.z;
    const int ks = p.ksplit > 1 ? blockIdx.z : 0;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const bf16_t*)p.A + (int64_t)z * p.sA), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const bf16_t*)p.W + (int64_t)z * p.sW), 0, p.w_bytes, 0x00020000);

    const int srow = tid >> 3;  // staging: 8 chunks per row, 32 rows per pass
    // register staging writes logical chunk `tid&7` at its swizzled LDS position; the LDS-DMA path (PF == 0) can
    // only write lane-linear, so there the SOURCE chunk is swizzled instead (same involution, rows srow + 32 i
    // share (row>>1)&7)
    const int chunk = PF == 0 ? ((tid & 7) ^ ((srow >> 1) & 7)) : (tid & 7);
    unsigned aoff[NA], woff[NW];                  // byte offsets of (slot row, current tap, channel 0)
    int pb[NA], py[NA], px[NA];                   // conv: output pixel of each A slot
    bool pv[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int m = m0 + srow + 32 * i;
        pv[i] = m < p.M;
        pb[i] = py[i] = px[i] = 0;
        if (CONV) {
            if (pv[i]) {
                const int hw = p.Hout * p.Wout;
                if (((hw & (hw - 1)) | (p.Wout & (p.Wout - 1))) == 0) {  // power-of-two feature maps (every SD-1.5 level): shifts
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
            aoff[i] = pv[i] ? (unsigned)m * (unsigned)p.lda * 2u + (unsigned)chunk * 16u : kOOB;
        }
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int n = n0 + srow + 32 * i;
        woff[i] = n < p.N ? (unsigned)n * (unsigned)p.ldw * 2u + (unsigned)chunk * 16u : kOOB;
    }

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x4 rga[PF > 0 ? PF : 1][NA], rgw[PF > 0 ? PF : 1][NW];
    const int nk_total = p.K / BK;
    const int per = (nk_total + p.ksplit - 1) / p.ksplit;
    const int kt_begin = ks * per;
    const int nk = (kt_begin + per <= nk_total ? per : nk_total - kt_begin);  // K steps of this block (may be <= 0)
    int tap = 0, c0 = 0;     // conv: (tap, channel) of the K step being LOADED
    bool newtap = true;
    if (CONV) {
        const int kb = kt_begin * BK;
        tap = kb / p.Cin;
        c0 = kb - tap * p.Cin;
    }

    auto load_tile = [&](auto SET, int kt) {
        constexpr int S = decltype(SET)::value;
        const unsigned kbytes = (unsigned)(kt_begin + kt) * (BK * 2);
        unsigned abytes = kbytes;  // gemm: k offset; conv: channel offset inside the tap
        if (CONV) {
            if (newtap) {  // wave-uniform: once per tap
                const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
                for (int i = 0; i < NA; ++i) aoff[i] = conv_tap_offset<CONV>(p, pv[i], pb[i], py[i], px[i], ky, kx, chunk);
                newtap = false;
            }
            abytes = (unsigned)c0 * 2u;
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) rga[S][i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, aoff[i] + abytes, 0, 0);
#pragma unroll
        for (int i = 0; i < NW; ++i) rgw[S][i] = __builtin_amdgcn_raw_buffer_load_b128(rsW, woff[i] + kbytes, 0, 0);
        if (CONV) {
            c0 += BK;
            if (c0 >= p.Cin) { c0 = 0; ++tap; newtap = true; }
        }
    };
    auto store_tile = [&](auto SET, int buf) {
        constexpr int S = decltype(SET)::value;
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<u32x4*>(smem + buf * kStage + lds_off(srow + 32 * i, chunk)) = rga[S][i];
#pragma unroll
        for (int i = 0; i < NW; ++i) *reinterpret_cast<u32x4*>(smem + buf * kStage + BM * 128 + lds_off(srow + 32 * i, chunk)) = rgw[S][i];
    };
    const int frow = lane & 15, fq = lane >> 4;
    auto compute = [&](int cur) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            uint4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[i] = *reinterpret_cast<const uint4*>(smem + cur * kStage + lds_off(wr * (BM / 2) + i * 16 + frow, 4 * s + fq));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                b[j] = *reinterpret_cast<const uint4*>(smem + cur * kStage + BM * 128 + lds_off(wc * (BN / 2) + j * 16 + frow, 4 * s + fq));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = Half<HT>::mfma16(b[j], a[i], acc[i][j]);  // D[n][m]: lane = (m, 4 n's)
        }
    };

    // LDS-DMA: buffer_load ... lds writes 64 lanes x 16 B = 8 tile rows per wave instruction at the wave-uniform LDS
    // address in M0.  Issued through inline asm so that hipcc does not serialise it against the ds_reads of the
    // OTHER stage (with the builtin it waits vmcnt(0) before every LDS read); completion is awaited by the explicit
    // s_waitcnt vmcnt(0) in front of the stage-release barrier.
    const int wuni = __builtin_amdgcn_readfirstlane(wid);
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);
    u32x4 dA, dW;  // hand-built buffer descriptors (base, base_hi | stride 0, num_records, flags)
    {
        const uint64_t ba = (uint64_t)((const bf16_t*)p.A + (int64_t)z * p.sA), bw = (uint64_t)((const bf16_t*)p.W + (int64_t)z * p.sW);
        dA = u32x4{(unsigned)__builtin_amdgcn_readfirstlane((unsigned)ba), (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(ba >> 32) & 0xffffu),
                   (unsigned)__builtin_amdgcn_readfirstlane(p.a_bytes), 0x00020000u};
        dW = u32x4{(unsigned)__builtin_amdgcn_readfirstlane((unsigned)bw), (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(bw >> 32) & 0xffffu),
                   (unsigned)__builtin_amdgcn_readfirstlane(p.w_bytes), 0x00020000u};
    }
    // see gemm_ring_kernel::dma16 (M0 clobbered, K offset as soffset)
    auto dma16 = [&](const u32x4& desc, unsigned lds_addr, unsigned voff, unsigned soff) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds"
                     :
                     : "v"(voff), "s"(lds_addr), "s"(desc), "s"(soff)
                     : "memory", "m0");
#pragma clang diagnostic pop
    };
    auto dma_tile = [&](int kt, int buf) {
        const unsigned kbytes = (unsigned)(kt_begin + kt) * (BK * 2);
        unsigned abytes = kbytes;
        if (CONV) {
            if (newtap) {
                const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
                for (int i = 0; i < NA; ++i) aoff[i] = conv_tap_offset<CONV>(p, pv[i], pb[i], py[i], px[i], ky, kx, chunk);
                newtap = false;
            }
            abytes = (unsigned)c0 * 2u;
        }
        const unsigned stage = lds_base + (unsigned)buf * kStage + (unsigned)wuni * (8 * 128);
#pragma unroll
        for (int i = 0; i < NA; ++i) dma16(dA, stage + i * (32 * 128), aoff[i], abytes);
#pragma unroll
        for (int i = 0; i < NW; ++i) dma16(dW, stage + BM * 128 + i * (32 * 128), woff[i], kbytes);
        if (CONV) {
            c0 += BK;
            if (c0 >= p.Cin) { c0 = 0; ++tap; newtap = true; }
        }
    };
    auto dma_wait_and_release = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMAs have landed in LDS ...
        __syncthreads();                                  // ... and every wave's; all reads of the other stage are done
    };

    if (nk > 0) {
        if constexpr (PF == 0) {
            // two LDS stages; the DMA of tile kt+1 is in flight while tile kt is multiplied; __syncthreads() waits for
            // this wave's DMAs (vmcnt(0)) and then for every wave, which also releases stage kt&1 for the next DMA
            dma_tile(0, 0);
            dma_wait_and_release();
            for (int kt = 0; kt < nk; ++kt) {
                const int cur = kt & 1;
                if (kt + 1 < nk) dma_tile(kt + 1, cur ^ 1);
                compute(cur);
                dma_wait_and_release();
            }
        } else if constexpr (PF == 1) {
            load_tile(IntC<0>{}, 0);
            store_tile(IntC<0>{}, 0);
            __syncthreads();
            for (int kt = 0; kt < nk; ++kt) {
                const int cur = kt & 1;
                if (kt + 1 < nk) load_tile(IntC<0>{}, kt + 1);
                compute(cur);
                if (kt + 1 < nk) store_tile(IntC<0>{}, cur ^ 1);
                __syncthreads();
            }
        } else {
            // tile kt lives in LDS[kt&1]; tile kt+1 is in flight in register set (kt+1)&1; tile kt+2 is issued
            // into set kt&1 at the top of iteration kt (that set was drained to LDS at the end of iteration kt-1)
            load_tile(IntC<0>{}, 0);
            if (nk > 1) load_tile(IntC<1>{}, 1);
            store_tile(IntC<0>{}, 0);
            __syncthreads();
            auto iter = [&](auto SET, int kt) {
                constexpr int S = decltype(SET)::value;  // == kt & 1
                if (kt + 2 < nk) load_tile(IntC<S>{}, kt + 2);
                compute(S);
                if (kt + 1 < nk) store_tile(IntC<S ^ 1>{}, S ^ 1);
                __syncthreads();
            };
            int kt = 0;
            for (; kt + 1 < nk; kt += 2) {
                iter(IntC<0>{}, kt);
                iter(IntC<1>{}, kt + 1);
            }
            if (kt < nk) iter(IntC<0>{}, kt);
        }
    }

    // every path above ends on the stage-release barrier: the LDS stages are dead and can carry the epilogue strips
    if constexpr (TM % 2 == 0 && (size_t)4 * 32 * (TN * 16 + 4) * 4 <= (size_t)2 * (BM + BN) * 128) {
        const bool rows_ok = !p.out_f32 && p.ksplit <= 1 && p.act != GMD_ACT_GEGLU && m0 + BM <= p.M && n0 + BN <= p.N && (p.ldc & 7) == 0 &&
                             (p.sC & 7) == 0 && (p.residual == nullptr || ((p.ldr & 7) == 0 && (p.sR & 7) == 0)) &&
                             (p.rowbias == nullptr || ((p.ldrb & 3) == 0 && (reinterpret_cast<uintptr_t>(p.rowbias) & 15) == 0)) &&
                         (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0;
        if (rows_ok) {
            __syncthreads();
            epilogue_rows<HT, TM, TN>(p, acc, reinterpret_cast<float*>(smem) + wid * (32 * (TN * 16 + 4)), m0 + wr * (BM / 2), n0 + wc * (BN / 2),
                                  lane, z);
            return;
        }
    }
    epilogue_regs<HT, TM, TN>(p, acc, m0 + wr * (BM / 2), n0 + wc * (BN / 2), frow, fq, z, ks);
}

// ------------------------------------------------------------------------------------------------
// bf16 MFMA kernel, ring form: WM x WN waves (wave tile 64 x 16*TN), NST-stage LDS ring filled by LDS-DMA with
// COUNTED waits -- the DMAs of the next NST-2 tiles stay in flight across the per-tile barrier, so the ~1 us
// DMA latency is covered by NST-1 tiles of MFMA work instead of one.
//   iteration kt:  s_waitcnt vmcnt(groups still allowed in flight)   tile kt of THIS wave has landed
//                  barrier                                           ... of every wave; stage (kt-1)%NST is free
//                  issue DMA of tile kt+NST-1 into stage (kt-1)%NST
//                  MFMAs on stage kt%NST
// ------------------------------------------------------------------------------------------------

template <typename HT, bool CONV, int WM, int WN, int TN, int NST>
__global__ __launch_bounds__(WM* WN * 64, (WM * WN) >= 4 ? (WM * WN) / 4 : 1) void gemm_ring_kernel(const GemmParams p) {
    GMD_WG_TRACE_SCOPE(WGK_RING | (CONV ? WGK_CONV_BIT : 0));
    constexpr int NWAVES = WM * WN, TM = 4;
    constexpr int BM = WM * 64, BN = WN * TN * 16;
    constexpr int RPP = NWAVES * 8;                 // tile rows written per staging pass (8 rows per wave instruction)
    constexpr int NA = BM / RPP;                    // A staging slots per thread
    constexpr int NW = (BN + RPP - 1) / RPP;        // W staging slots per thread (the last may be invalid for some waves)
    constexpr bool W_RAGGED = (BN % RPP) != 0;
    constexpr int D = NST - 1;                      // prefetch distance in tiles
    static_assert(BM % RPP == 0 && BN % 8 == 0 && NST >= 2 && NST <= 4, "bad ring geometry");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int kStage = (BM + BN) * 128;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid / WN, wc = wid % WN;
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (private L2 each), so workgroup ids with
    // equal id % 8 get a CONTIGUOUS run of tiles (bijective remap, cdna guide T1).  Tiles are numbered n-fastest, so
    // the N-tiles of one row block and neighbouring row blocks (whose conv taps overlap) share one L2.
    int m0, n0;
    {
        const int nwg = gridDim.x, id = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, j = id >> 3;
        const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
        const int tiles_n = (p.N + BN - 1) / BN;
        const int mt = L / tiles_n;
        m0 = mt * BM;
        n0 = (L - mt * tiles_n) * BN;
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
                if (((hw & (hw - 1)) | (p.Wout & (p.Wout - 1))) == 0) {  // power-of-two feature maps (every SD-1.5 level): shifts
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
            aoff[i] = pv[i] ? (unsigned)m * (unsigned)p.lda * 2u + (unsigned)chunk * 16u : kOOB;
        }
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int rl = srow + RPP * i;  // row inside the W tile
        const int n = n0 + rl;
        woff[i] = (rl < BN && n < p.N) ? (unsigned)n * (unsigned)p.ldw * 2u + (unsigned)chunk * 16u : kOOB;
    }
    // does this wave own a valid row group in the last (ragged) W pass?  wave-uniform
    const bool w_last = !W_RAGGED || ((NW - 1) * RPP + wuni * 8 < BN);

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk_total = p.K / BK;
    const int per = (nk_total + p.ksplit - 1) / p.ksplit;
    const int kt_begin = ks * per;
    const int nk = (kt_begin + per <= nk_total ? per : nk_total - kt_begin);
    int tap = 0, c0 = 0, cb0 = 0;  // filter tap / channel of the K step being LOADED, first channel of its channel block
    bool newtap = true;
    if (CONV) {
        const int sb = p.cblk / BK, per_cb = 9 * sb;  // K steps per (block, tap) / per channel block
        const int cbi = kt_begin / per_cb, rem = kt_begin - cbi * per_cb;
        tap = rem / sb;
        cb0 = cbi * p.cblk;
        c0 = cb0 + (rem - tap * sb) * BK;
    }

    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);
    u32x4 dA, dW;
    {
        const uint64_t ba = (uint64_t)((const bf16_t*)p.A + (int64_t)z * p.sA), bw = (uint64_t)((const bf16_t*)p.W + (int64_t)z * p.sW);
        dA = u32x4{(unsigned)__builtin_amdgcn_readfirstlane((unsigned)ba), (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(ba >> 32) & 0xffffu),
                   (unsigned)__builtin_amdgcn_readfirstlane(p.a_bytes), 0x00020000u};
        dW = u32x4{(unsigned)__builtin_amdgcn_readfirstlane((unsigned)bw), (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(bw >> 32) & 0xffffu),
                   (unsigned)__builtin_amdgcn_readfirstlane(p.w_bytes), 0x00020000u};
    }
    // (M0 is a reserved register: nothing else in this kernel reads it -- gfx950 LDS instructions do not -- so it is listed as
    // clobbered and the compiler's "reserved register" diagnostic is silenced for this statement only)
    // one LDS-DMA piece: M0 = wave-uniform LDS destination, per-lane row offset in a VGPR, the K offset of the tile as the
    // scalar soffset (no VALU add per piece); M0 is declared clobbered instead of saved / restored around every piece
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
        unsigned kbytes = (unsigned)(kt_begin + kt) * (BK * 2);
        unsigned abytes = kbytes;
        if (CONV) {
            if (newtap) {
                const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
                for (int i = 0; i < NA; ++i) aoff[i] = conv_tap_offset<CONV>(p, pv[i], pb[i], py[i], px[i], ky, kx, chunk);
                newtap = false;
            }
            abytes = (unsigned)c0 * 2u;
            kbytes = (unsigned)(tap * p.Cin + c0) * 2u;  // weights are [Cout][tap][Cin]
        }
        const unsigned stage = lds_base + (unsigned)stage_idx * kStage + (unsigned)wuni * (8 * 128);
#pragma unroll
        for (int i = 0; i < NA; ++i) dma16(dA, stage + i * (RPP * 128), aoff[i], abytes);
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            if (i + 1 < NW || w_last) dma16(dW, stage + BM * 128 + i * (RPP * 128), woff[i], kbytes);
        }
        if (CONV) {
            c0 += BK;
            if (c0 >= cb0 + p.cblk) {  // next tap of this channel block, or the first tap of the next block
                c0 = cb0;
                ++tap;
                newtap = true;
                if (tap == 9) { tap = 0; cb0 += p.cblk; c0 = cb0; }
            }
        }
    };
    // wait until all but `ahead` of this wave's tile DMA groups have landed (a group = NA + NW or NA + NW - 1 loads)
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
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            uint4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const uint4*>(sA + lds_off(wr * 64 + i * 16 + frow, 4 * s2 + fq));
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const uint4*>(sW + lds_off(wc * (TN * 16) + j * 16 + frow, 4 * s2 + fq));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = Half<HT>::mfma16(b[j], a[i], acc[i][j]);
        }
    };

    if (nk > 0) {
#pragma unroll
        for (int t = 0; t < D; ++t)
            if (t < nk) dma_tile(t, t);
        int st_cur = 0, st_fill = D % NST;  // stage of tile kt / of tile kt+D
        for (int kt = 0; kt < nk; ++kt) {
            const int rem = nk - 1 - kt;
            wait_tiles(rem < D - 1 ? rem : D - 1);
            __syncthreads();  // the asm DMAs are invisible to hipcc, so this is a bare s_barrier (+ lgkmcnt for its own ds ops)
            if (kt + D < nk) dma_tile(kt + D, st_fill);
            compute(st_cur);
            st_cur = st_cur + 1 == NST ? 0 : st_cur + 1;
            st_fill = st_fill + 1 == NST ? 0 : st_fill + 1;
        }
    }
    // block-uniform choice: full bf16 tile, plain epilogue -> row-contiguous stores through LDS
    const bool rows_ok = !p.out_f32 && p.ksplit <= 1 && p.act != GMD_ACT_GEGLU && m0 + BM <= p.M && n0 + BN <= p.N && (p.ldc & 7) == 0 &&
                         (p.sC & 7) == 0 && (p.residual == nullptr || ((p.ldr & 7) == 0 && (p.sR & 7) == 0)) &&
                         (p.rowbias == nullptr || ((p.ldrb & 3) == 0 && (reinterpret_cast<uintptr_t>(p.rowbias) & 15) == 0)) &&
                         (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0;
    if constexpr (TN % 2 == 0) {
        if (p.act == GMD_ACT_GEGLU && !p.out_f32 && p.ksplit <= 1 && m0 + BM <= p.M && n0 + BN <= p.N && (p.ldc & 7) == 0 && (p.sC & 7) == 0 &&
            (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0) {
            __syncthreads();
            constexpr int kStripG = 32 * (TN * 8 + 4);
            epilogue_rows_geglu<HT, TM, TN>(p, acc, reinterpret_cast<float*>(smem) + wid * kStripG, m0 + wr * 64, n0 + wc * (TN * 16), lane, z);
            return;
        }
    }
    if (p.ksplit > 1 && m0 + BM <= p.M && n0 + BN <= p.N && (p.N & 3) == 0) {
        __syncthreads();
        constexpr int kStripS = 32 * (TN * 16 + 4);
        epilogue_rows_slab<TM, TN>(p, acc, reinterpret_cast<float*>(smem) + wid * kStripS, m0 + wr * 64, n0 + wc * (TN * 16), lane, ks);
        return;
    }
    if (__builtin_expect(rows_ok, 1)) {
        __syncthreads();  // every wave is done with the K-loop stages: the strips below overwrite them
        constexpr int kStrip = 32 * (TN * 16 + 4);  // floats per wave
        static_assert((size_t)NWAVES * kStrip * 4 <= (size_t)NST * kStage, "epilogue strips must fit in the ring stages");
        epilogue_rows<HT, TM, TN>(p, acc, reinterpret_cast<float*>(smem) + wid * kStrip, m0 + wr * 64, n0 + wc * (TN * 16), lane, z);
    } else {
        epilogue_regs<HT, TM, TN>(p, acc, m0 + wr * 64, n0 + wc * (TN * 16), frow, fq, z, ks);
    }
}

// ------------------------------------------------------------------------------------------------
// bf16 MFMA kernel, ping-pong form (round 4): ONE workgroup per CU on a 256 x (32 TN) tile -- 8 consumer waves (wave tile
// 64 x 16 TN, the two waves of every SIMD staggered by one barrier so that one multiplies while the other reads fragments)
// and LW loader waves that do nothing but issue the LDS-DMA of later tiles.
//
// Why (in-kernel stamps, tools/pp_diag.py, conv 8x64x64 640->320):
//  * the ring kernel above is bound by what a CU can pull from L2 into its LDS (~35 B/clk: 97 us with the MFMAs removed =
//    6.5 MB per CU at 67 GB/s); a 256 x 160 tile needs 52 KB per K step where two co-resident 128 x 160 tiles need 72 KB;
//  * an LDS-DMA instruction BLOCKS its wave while the CU's memory queue is full -- ~90 cycles per piece with four waves
//    issuing at once.  In the first version of this kernel the consumers issued the DMA in their read segments: 285 cycles of
//    blocked issue + 280 of fragment reads + the vmcnt wait = 700 cycles opposite a 380-cycle MFMA segment, i.e. the matrix
//    pipe idled half of every interval.  Loader waves take that blocking off the consumers' path.
//
//   consumer, per K step kt (stage kt % 3):   R0: 4 + TN ds_read_b128 (k 0..31) ; lgkmcnt(0) ; barrier
//                                             C0: 4 x TN MFMAs                   ; barrier
//                                             R1: reads (k 32..63) ; lgkmcnt(0)  ; barrier
//                                             C1: 4 x TN MFMAs                   ; barrier
//     waves 4-7 execute one extra barrier before the loop, waves 0-3 one after it: in every barrier interval one group reads
//     and the other multiplies (the 8-phase template of the programming guide, cut for 64 x 16 TN wave tiles);
//   loader, per K step kt:  four intervals, each: a quarter of tile kt+2's pieces ; [last: vmcnt(tile kt+1 landed)] ; barrier
//
// LDS hazards, by global barrier number (prologue barrier = #0; waves 0-3 and the loaders end their four segments of step kt
// at #4kt+1..#4kt+4, waves 4-7 one later): the last reads of tile kt-1 (waves 4-7, R1) have RETURNED (lgkmcnt(0)) before they
// arrive at #4kt; the DMA of tile kt+2 into that stage is issued after #4kt.  A loader's pieces of tile kt+1 have landed
// (counted vmcnt) before it arrives at #4kt+4, and the first read of tile kt+1 (waves 0-3, R0) is issued after #4kt+4.
// ------------------------------------------------------------------------------------------------
template <typename HT, bool CONV, int TN>
__global__ __launch_bounds__(768, 3) void gemm_pp_kernel(const GemmParams p) {
    GMD_WG_TRACE_SCOPE(WGK_PP | (CONV ? WGK_CONV_BIT : 0));
    constexpr int WN = 2, NCONS = 8, LW = 4, TM = 4, NST = 3;
    constexpr int BM = 256, BN = WN * TN * 16;
    constexpr int RPP = LW * 8;                     // 32 tile rows per staging pass (8 rows per loader-wave instruction)
    constexpr int NA = BM / RPP;                    // 8 A pieces per loader wave and tile
    constexpr int NW = BN / RPP;                    // 5 / 4 W pieces
    constexpr int NP = NA + NW;                     // pieces per loader wave and tile
    static_assert(BN % RPP == 0, "W tile rows must be whole staging passes");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int kStage = (BM + BN) * 128;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wuni = __builtin_amdgcn_readfirstlane(wid);
    const bool loader = wuni >= NCONS;
    const bool late = wuni >= 4 && !loader;         // the staggered consumer group (SIMD partners of waves 0-3)
    int m0, n0;
    {
        const int nwg = gridDim.x, id = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, j = id >> 3;
        const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
        const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
        int mt, nt;
        grouped_tile(L, tiles_m, tiles_n, p.tile_group, mt, nt);
        m0 = mt * BM;
        n0 = nt * BN;
    }
    const int z = p.ksplit > 1 ? 0 : blockIdx.z;
    const int ks = p.ksplit > 1 ? blockIdx.z : 0;
    const int nk_total = p.K / BK;
    const int per = (nk_total + p.ksplit - 1) / p.ksplit;
    const int kt_begin = ks * per;
    const int nk = (kt_begin + per <= nk_total ? per : nk_total - kt_begin);

    // end of a segment: nothing of it may sink below the barrier, nothing of the next may rise above it
    auto seg_barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
#ifdef GMD_PP_DIAG
    // diagnostic build only (tools/dbg/libgmd_ppdiag.so): where a wave's loop time goes.  Stamps go to the workspace, which no
    // other code of this launch reads.  Never timed as a kernel: the stamps' lgkmcnt(0) forbid overlaps the product has.
    unsigned long long dg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dprev = 0, dreal0 = 0, dtime0 = 0;
    auto stamp_now = [&]() {
        unsigned long long t;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        __builtin_amdgcn_sched_barrier(0);
        return t;
    };
    auto diag_begin = [&]() {
        dprev = stamp_now();
        dtime0 = dprev;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dreal0)::"memory");
    };
    auto diag_end = [&]() {
        unsigned long long dreal1;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dreal1)::"memory");
        dg[7] = dreal1 - dreal0;  // 100 MHz ticks
#ifdef GMD_WG_TRACE
        // with the occupancy trace: one DETAIL record per wave into the CU's trace array (kind | 0x80), so that the loop split is
        // available for the launches of a whole pipeline run, co-running with the other stream (tools/cu_occupancy.py --pp-detail)
        if (lane == 0 && g_wg_trace != nullptr) {
            GmdWgTraceHeader* h = g_wg_trace;
            const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
            const unsigned shard = ((((xcc & 7u) * 8u + ((hw >> 13) & 7u)) * 2u + ((hw >> 12) & 1u)) * 16u) + ((hw >> 8) & 15u);
            unsigned long long* counters = reinterpret_cast<unsigned long long*>(h + 1);
            const unsigned long long i = __hip_atomic_fetch_add(counters + 16ull * shard, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (i < h->capacity) {
                uint4* rec = reinterpret_cast<uint4*>(counters + 16ull * h->shards) + 2ull * (shard * h->capacity + i);
                // consumers: reads + wait | barrier after R | mfma issue | barrier after C;  loaders: dma issue | loader barrier | - | vmcnt wait
                const unsigned a0 = (unsigned)(loader ? dg[1] : dg[0]), a1 = (unsigned)dg[3], a2 = (unsigned)(loader ? 0 : dg[4]), a3 = (unsigned)(loader ? dg[6] : dg[5]);
                rec[0] = make_uint4(a0, a1, a2, a3);
                rec[1] = make_uint4(hw, (xcc & 15u) | ((unsigned)((WGK_PP | (CONV ? WGK_CONV_BIT : 0)) | 0x80) << 8) | ((unsigned)wid << 16), (unsigned)nk | ((unsigned)TN << 16),
                                    (unsigned)(dprev - dtime0));
            }
        }
#else
        if (p.ws && lane == 0) {
            unsigned long long* o = reinterpret_cast<unsigned long long*>(p.ws) + ((size_t)blockIdx.x * (NCONS + LW) + wid) * 10;
            for (int k = 0; k < 8; ++k) o[k] = dg[k];
            o[8] = dprev - dtime0;
            o[9] = (unsigned long long)nk;
        }
#endif
    };
#define PP_STAMP(k) { const unsigned long long t_ = stamp_now(); dg[k] += t_ - dprev; dprev = t_; }
#else
#define PP_STAMP(k)
#endif

    if (loader) {
        // ---------------------------------------------------------------- loader waves: LDS-DMA only
        const int lw = wuni - NCONS;                                  // 0 .. LW-1
        const int ltid = tid - NCONS * 64;
        const int srow = ltid >> 3;                                   // 0 .. 31
        const int chunk = (ltid & 7) ^ ((srow >> 1) & 7);             // swizzled SOURCE chunk (RPP is a multiple of 16)
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
                aoff[i] = pv[i] ? (unsigned)m * (unsigned)p.lda * 2u + (unsigned)chunk * 16u : kOOB;
            }
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int n = n0 + srow + RPP * i;
            woff[i] = n < p.N ? (unsigned)n * (unsigned)p.ldw * 2u + (unsigned)chunk * 16u : kOOB;
        }
        int tap = 0, c0 = 0, cb0 = 0;
        bool newtap = true;
        if (CONV) {
            const int sb = p.cblk / BK, per_cb = 9 * sb;
            const int cbi = kt_begin / per_cb, rem = kt_begin - cbi * per_cb;
            tap = rem / sb;
            cb0 = cbi * p.cblk;
            c0 = cb0 + (rem - tap * sb) * BK;
        }
        const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);
        u32x4 dA, dW;
        {
            const uint64_t ba = (uint64_t)((const bf16_t*)p.A + (int64_t)z * p.sA), bw = (uint64_t)((const bf16_t*)p.W + (int64_t)z * p.sW);
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
        // quarter Q of one tile's NP pieces (pieces are numbered A passes first, then W passes); Q == 4: the whole tile
        auto dma_part = [&](auto QC, int kt, int stage_idx) {
            constexpr int Q = decltype(QC)::value;
            constexpr int lo = Q == 4 ? 0 : Q * NP / 4, hi = Q == 4 ? NP : (Q + 1) * NP / 4;
            unsigned kbytes = (unsigned)(kt_begin + kt) * (BK * 2);
            unsigned abytes = kbytes;
            if (CONV) {
                if ((Q == 0 || Q == 4) && newtap) {
                    const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
                    for (int i = 0; i < NA; ++i) aoff[i] = conv_tap_offset<CONV>(p, pv[i], pb[i], py[i], px[i], ky, kx, chunk);
                    newtap = false;
                }
                abytes = (unsigned)c0 * 2u;
                kbytes = (unsigned)(tap * p.Cin + c0) * 2u;
            }
            const unsigned stage = lds_base + (unsigned)stage_idx * kStage + (unsigned)lw * (8 * 128);
#pragma unroll
            for (int i = 0; i < NA; ++i)
                if (i >= lo && i < hi) dma16(dA, stage + i * (RPP * 128), aoff[i], abytes);
#pragma unroll
            for (int i = 0; i < NW; ++i)
                if (NA + i >= lo && NA + i < hi) dma16(dW, stage + BM * 128 + i * (RPP * 128), woff[i], kbytes);
            if (CONV && (Q == 3 || Q == 4)) {
                c0 += BK;
                if (c0 >= cb0 + p.cblk) {
                    c0 = cb0;
                    ++tap;
                    newtap = true;
                    if (tap == 9) { tap = 0; cb0 += p.cblk; c0 = cb0; }
                }
            }
        };
        if (nk > 0) {  // block-uniform
            dma_part(IntC<4>{}, 0, 0);
            if (nk > 1) {
                dma_part(IntC<4>{}, 1, 1);
                wait_vmcnt<NP>();
            } else {
                wait_vmcnt<0>();
            }
            seg_barrier();  // #0: tile 0 is in LDS
            int st_fill = 2;
#ifdef GMD_PP_DIAG
            diag_begin();
#endif
            for (int kt = 0; kt < nk; ++kt) {
                const bool more = kt + 2 < nk;
                if (more) dma_part(IntC<0>{}, kt + 2, st_fill);
                PP_STAMP(1)
                seg_barrier();
                PP_STAMP(3)
                if (more) dma_part(IntC<1>{}, kt + 2, st_fill);
                PP_STAMP(1)
                seg_barrier();
                PP_STAMP(3)
                if (more) dma_part(IntC<2>{}, kt + 2, st_fill);
                PP_STAMP(1)
                seg_barrier();
                PP_STAMP(3)
                if (more) dma_part(IntC<3>{}, kt + 2, st_fill);
                PP_STAMP(1)
                if (more) wait_vmcnt<NP>();  // all but the NP pieces of tile kt+2: tile kt+1 has landed
                else wait_vmcnt<0>();
                PP_STAMP(6)
                seg_barrier();
                PP_STAMP(3)
                st_fill = st_fill == NST - 1 ? 0 : st_fill + 1;
            }
#ifdef GMD_PP_DIAG
            diag_end();
#endif
            seg_barrier();  // the trailing barrier of the early consumer group
        }
        __syncthreads();  // the consumers' barrier in front of their epilogue strips
        return;
    }

    // -------------------------------------------------------------------- consumer waves: fragment reads + MFMA
    const int wr = wid / WN, wc = wid % WN;
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fq = lane >> 4;
    uint4 fa[TM], fb[TN];
    auto frag_reads = [&](int stage_idx, int s2) {
        const unsigned char* sA = smem + stage_idx * kStage;
        const unsigned char* sW = sA + BM * 128;
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const uint4*>(sA + lds_off(wr * 64 + i * 16 + frow, 4 * s2 + fq));
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const uint4*>(sW + lds_off(wc * (TN * 16) + j * 16 + frow, 4 * s2 + fq));
    };
    auto mfmas = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = Half<HT>::mfma16(fb[j], fa[i], acc[i][j]);
    };
    auto lgkm0 = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };

    if (nk > 0) {  // block-uniform (split-K slices beyond the last K step skip the loop and its barriers together)
        seg_barrier();                  // #0: tile 0 is in LDS
        if (late) seg_barrier();        // the stagger
        int st = 0;
#ifdef GMD_PP_DIAG
        diag_begin();
#endif
        for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                frag_reads(st, s2);
                lgkm0();
                PP_STAMP(0)
                seg_barrier();
                PP_STAMP(3)
                __builtin_amdgcn_s_setprio(1);
                mfmas();
                __builtin_amdgcn_s_setprio(0);
                PP_STAMP(4)
                seg_barrier();
                PP_STAMP(5)
            }
            st = st == NST - 1 ? 0 : st + 1;
        }
#ifdef GMD_PP_DIAG
        diag_end();
#endif
        if (!late) seg_barrier();
    }
#if defined(GMD_PP_DIAG) && defined(GMD_WG_TRACE)
    // phase record of the workgroup (wave 0, lane 0), written where the kernel ends: 100 MHz ticks kernel entry -> K loop | K loop |
    // barrier + in-kernel reduction | epilogue (to the last store ISSUED) -- tools/cu_occupancy.py --pp-detail
    unsigned long long ph_sync = 0;
    auto real_now = [&]() { unsigned long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); return t; };
    auto phase_emit = [&]() {
        if (wid != 0 || lane != 0 || g_wg_trace == nullptr || nk <= 0) return;
        const unsigned long long t_end = real_now();
        GmdWgTraceHeader* h = g_wg_trace;
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        const unsigned shard = ((((xcc & 7u) * 8u + ((hw >> 13) & 7u)) * 2u + ((hw >> 12) & 1u)) * 16u) + ((hw >> 8) & 15u);
        unsigned long long* counters = reinterpret_cast<unsigned long long*>(h + 1);
        const unsigned long long i2 = __hip_atomic_fetch_add(counters + 16ull * shard, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (i2 < h->capacity) {
            uint4* rec = reinterpret_cast<uint4*>(counters + 16ull * h->shards) + 2ull * (shard * h->capacity + i2);
            rec[0] = make_uint4((unsigned)(dreal0 - gmd_wg_trace_scope_.t0), (unsigned)dg[7], (unsigned)(ph_sync - (dreal0 + dg[7])), (unsigned)(t_end - ph_sync));
            rec[1] = make_uint4(hw, (xcc & 15u) | ((unsigned)((WGK_PP | (CONV ? WGK_CONV_BIT : 0)) | 0x80) << 8) | (0xFFu << 16), (unsigned)nk | ((unsigned)TN << 16),
                                (gridDim.x * gridDim.z) | ((unsigned)p.ksplit << 20) | (p.residual ? (1u << 28) : 0u) | (p.fixup ? (1u << 29) : 0u));
        }
    };
#define PP_PHASE_SYNC() ph_sync = real_now()
#define PP_PHASE_EMIT() phase_emit()
#else
#define PP_PHASE_SYNC()
#define PP_PHASE_EMIT()
#endif
    __syncthreads();  // every wave (loaders included) is done with the K-loop stages: the strips below overwrite them
    GemmParams q = p;
    if (p.fixup) {  // in-kernel split-K reduction: the producer slices leave here, the last slice goes on as an unsplit launch
        if (!splitk_fixup<TM, TN, NCONS>(p, acc, ks, wid, lane)) return;
        q.ksplit = 1;
    }
    PP_PHASE_SYNC();
    const bool rows_ok = !q.out_f32 && q.ksplit <= 1 && q.act != GMD_ACT_GEGLU && m0 + BM <= q.M && n0 + BN <= q.N && (q.ldc & 7) == 0 &&
                         (q.sC & 7) == 0 && (q.residual == nullptr || ((q.ldr & 7) == 0 && (q.sR & 7) == 0)) &&
                         (q.rowbias == nullptr || ((q.ldrb & 3) == 0 && (reinterpret_cast<uintptr_t>(q.rowbias) & 15) == 0)) &&
                         (reinterpret_cast<uintptr_t>(q.bias) & 15) == 0;
    if constexpr (TN % 2 == 0) {
        if (q.act == GMD_ACT_GEGLU && !q.out_f32 && q.ksplit <= 1 && m0 + BM <= q.M && n0 + BN <= q.N && (q.ldc & 7) == 0 && (q.sC & 7) == 0 &&
            (reinterpret_cast<uintptr_t>(q.bias) & 15) == 0) {
            constexpr int kStripG = 32 * (TN * 8 + 4);
            epilogue_rows_geglu<HT, TM, TN>(q, acc, reinterpret_cast<float*>(smem) + wid * kStripG, m0 + wr * 64, n0 + wc * (TN * 16), lane, z);
            PP_PHASE_EMIT();
            return;
        }
    }
    if (__builtin_expect(q.ksplit > 1 && m0 + BM <= q.M && n0 + BN <= q.N && (q.N & 3) == 0, 0)) {
        constexpr int kStripS = 32 * (TN * 16 + 4);
        epilogue_rows_slab<TM, TN>(q, acc, reinterpret_cast<float*>(smem) + wid * kStripS, m0 + wr * 64, n0 + wc * (TN * 16), lane, ks);
        PP_PHASE_EMIT();
        return;
    }
    if (__builtin_expect(rows_ok, 1)) {
        constexpr int kStrip = 32 * (TN * 16 + 4);
        static_assert((size_t)NCONS * kStrip * 4 <= (size_t)NST * kStage, "epilogue strips must fit in the ring stages");
        epilogue_rows<HT, TM, TN>(q, acc, reinterpret_cast<float*>(smem) + wid * kStrip, m0 + wr * 64, n0 + wc * (TN * 16), lane, z);
    } else {
        epilogue_regs<HT, TM, TN>(q, acc, m0 + wr * 64, n0 + wc * (TN * 16), frow, fq, z, ks);
    }
    PP_PHASE_EMIT();
}
#undef PP_STAMP
#undef PP_PHASE_SYNC
#undef PP_PHASE_EMIT

// ------------------------------------------------------------------------------------------------
// bf16 MFMA kernel, loader / consumer form (round 4) for launches that have about ONE tile per CU (256 tiles of 128 x 160 or
// 64 x 160: the transformer linears and the convolutions of the 32x32 / 16x16 levels).  One workgroup of 4 consumer waves (2 x 2,
// one per SIMD, wave tile 16 TM x 16 TN) + 4 loader waves, a 4-stage LDS ring, ONE barrier per 64-deep K step.
//
// Why: such launches run the ring kernel with a single 4-wave workgroup per CU, where every wave serialises its own blocked
// LDS-DMA issue (~90 cycles per piece while the CU's memory queue is full), its fragment reads and its MFMAs -- the vendor
// library's kernels for exactly these shapes (profiles/r04_library_tensile_kernels.txt: MT160x128x64, MT160x64x64, MT96x64x64 ...,
// one wave per SIMD, 256 workgroups) are 7-20 % faster.  Here the loaders own the DMA (blocking costs the consumers nothing) and
// a consumer issues the fragment reads of the NEXT 32-deep half step (second register set) in front of the MFMAs of the current
// one, so its matrix pipe never waits for LDS.  These tiles are bound by what a CU can pull from L2 (36 KB per K step at
// ~36 B/clk = 1000 cycles against 640 MFMA cycles): the loaders are the critical path, the consumers have slack.
//
//   consumer:  B(0) ; reads (0, half 0) -> set 0 ;  per K step kt: { reads (kt, 1) -> set 1 ; MFMAs set 0 ;
//                                                                     reads (kt+1, 0) -> set 0 ; MFMAs set 1 ; B(kt+1) }
//   loader:    tiles 0..3 ; vmcnt(tiles 0, 1 landed) ; B(0) ;  per kt: { vmcnt(tile kt+2 landed) ; B(kt+1) ; issue tile kt+4 }
//
// LDS hazards: B(j) is passed by a loader only after ITS pieces of tile j+1 have landed (counted vmcnt), so every read of tile
// j+1 -- the first is issued in K step j, behind B(j) -- finds it.  The reads of tile kt are issued in K steps kt-1 and kt and
// have RETURNED when the MFMAs that consume them were issued, i.e. before B(kt+1); tile kt+4 is written into that stage behind
// B(kt+1).  Reads still in flight at a barrier target the NEXT tile's stage, which is not rewritten for three more K steps.
// ------------------------------------------------------------------------------------------------
template <typename HT, bool CONV, int TM, int TN>
__global__ __launch_bounds__(512, 2) void gemm_lc_kernel(const GemmParams p) {
    GMD_WG_TRACE_SCOPE(WGK_LC | (CONV ? WGK_CONV_BIT : 0));
    constexpr int WN = 2, NCONS = 4, LW = 4, NST = 4;
    constexpr int BM = 2 * TM * 16, BN = WN * TN * 16;
    constexpr int RPP = LW * 8;                     // 32 tile rows per staging pass
    constexpr int NA = BM / RPP, NW = BN / RPP, NP = NA + NW;
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be whole staging passes");
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
        int mt, nt;
        grouped_tile(L, tiles_m, tiles_n, p.tile_group, mt, nt);  // (tile_group = tiles_m: m fastest, for W-heavy launches)
        m0 = mt * BM;
        n0 = nt * BN;
    }
    const int z = p.ksplit > 1 ? 0 : blockIdx.z;
    const int ks = p.ksplit > 1 ? blockIdx.z : 0;
    const int nk_total = p.K / BK;
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
                aoff[i] = pv[i] ? (unsigned)m * (unsigned)p.lda * 2u + (unsigned)chunk * 16u : kOOB;
            }
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int n = n0 + srow + RPP * i;
            woff[i] = n < p.N ? (unsigned)n * (unsigned)p.ldw * 2u + (unsigned)chunk * 16u : kOOB;
        }
        int tap = 0, c0 = 0, cb0 = 0;
        bool newtap = true;
        if (CONV) {
            const int sb = p.cblk / BK, per_cb = 9 * sb;
            const int cbi = kt_begin / per_cb, rem = kt_begin - cbi * per_cb;
            tap = rem / sb;
            cb0 = cbi * p.cblk;
            c0 = cb0 + (rem - tap * sb) * BK;
        }
        const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);
        u32x4 dA, dW;
        {
            const uint64_t ba = (uint64_t)((const bf16_t*)p.A + (int64_t)z * p.sA), bw = (uint64_t)((const bf16_t*)p.W + (int64_t)z * p.sW);
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
            unsigned kbytes = (unsigned)(kt_begin + kt) * (BK * 2);
            unsigned abytes = kbytes;
            if (CONV) {
                if (newtap) {
                    const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
                    for (int i = 0; i < NA; ++i) aoff[i] = conv_tap_offset<CONV>(p, pv[i], pb[i], py[i], px[i], ky, kx, chunk);
                    newtap = false;
                }
                abytes = (unsigned)c0 * 2u;
                kbytes = (unsigned)(tap * p.Cin + c0) * 2u;
            }
            const unsigned stage = lds_base + (unsigned)stage_idx * kStage + (unsigned)lw * (8 * 128);
#pragma unroll
            for (int i = 0; i < NA; ++i) dma16(dA, stage + i * (RPP * 128), aoff[i], abytes);
#pragma unroll
            for (int i = 0; i < NW; ++i) dma16(dW, stage + BM * 128 + i * (RPP * 128), woff[i], kbytes);
            if (CONV) {
                c0 += BK;
                if (c0 >= cb0 + p.cblk) {
                    c0 = cb0;
                    ++tap;
                    newtap = true;
                    if (tap == 9) { tap = 0; cb0 += p.cblk; c0 = cb0; }
                }
            }
        };
        if (nk > 0) {  // block-uniform
#pragma unroll
            for (int t = 0; t < NST; ++t)
                if (t < nk) dma_tile(t, t);
            // tiles 0 and 1 have landed: at most the pieces of tiles 2 and 3 may remain
            if (nk >= 4) wait_vmcnt<2 * NP>();
            else if (nk == 3) wait_vmcnt<NP>();
            else wait_vmcnt<0>();
            seg_barrier();  // B(0)
            int st_fill = 0;
            for (int kt = 0; kt < nk; ++kt) {
                // tile kt+2 has landed: only tile kt+3 (issued behind B(kt)) may remain -- where it exists
                if (kt + 3 < nk) wait_vmcnt<NP>();
                else wait_vmcnt<0>();
                seg_barrier();  // B(kt+1): publishes tile kt+2, frees the stage of tile kt
                if (kt + NST < nk) dma_tile(kt + NST, st_fill);
                st_fill = st_fill == NST - 1 ? 0 : st_fill + 1;
            }
        }
        __syncthreads();  // the consumers' barrier in front of their epilogue strips
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
    uint4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
    auto frag_reads = [&](int stage_idx, int s2, uint4 (&fa)[TM], uint4 (&fb)[TN]) {
        const unsigned char* sA = smem + stage_idx * kStage;
        const unsigned char* sW = sA + BM * 128;
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const uint4*>(sA + lds_off(wr * (TM * 16) + i * 16 + frow, 4 * s2 + fq));
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const uint4*>(sW + lds_off(wc * (TN * 16) + j * 16 + frow, 4 * s2 + fq));
    };
    auto mfmas = [&](const uint4 (&fa)[TM], const uint4 (&fb)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = Half<HT>::mfma16(fb[j], fa[i], acc[i][j]);
    };
    if (nk > 0) {
        seg_barrier();  // B(0): tiles 0 and 1 are in LDS
        frag_reads(0, 0, fa0, fb0);
        int st = 0;
        for (int kt = 0; kt < nk; ++kt) {
            const int st_next = st == NST - 1 ? 0 : st + 1;
            frag_reads(st, 1, fa1, fb1);
            mfmas(fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);  // keep the half steps apart: set 0 is rewritten below
            // tile kt+1: published by B(kt).  Unconditional -- behind the last K step it reads a stale stage into registers nobody
            // uses: a branch here makes hipcc wait lgkmcnt(0) in front of the MFMAs above instead of counting the reads
            frag_reads(st_next, 0, fa0, fb0);
            mfmas(fa1, fb1);
            seg_barrier();  // B(kt+1)
            st = st_next;
        }
        // keep the last (unused) prefetch alive: otherwise hipcc sinks it under "is there a next K step", and that branch
        // turns the counted lgkmcnt in front of the first MFMA block into lgkmcnt(0)
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(__builtin_bit_cast(u32x4, fa0[i])));
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(__builtin_bit_cast(u32x4, fb0[j])));
    }
    __syncthreads();  // every wave (loaders included) is done with the K-loop stages: the strips below overwrite them
    GemmParams q = p;
    if (p.fixup) {  // in-kernel split-K reduction: the producer slices leave here, the last slice goes on as an unsplit launch
        if (!splitk_fixup<TM, TN, NCONS>(p, acc, ks, wid, lane)) return;
        q.ksplit = 1;
    }
    const bool rows_ok = !q.out_f32 && q.ksplit <= 1 && q.act != GMD_ACT_GEGLU && m0 + BM <= q.M && n0 + BN <= q.N && (q.ldc & 7) == 0 &&
                         (q.sC & 7) == 0 && (q.residual == nullptr || ((q.ldr & 7) == 0 && (q.sR & 7) == 0)) &&
                         (q.rowbias == nullptr || ((q.ldrb & 3) == 0 && (reinterpret_cast<uintptr_t>(q.rowbias) & 15) == 0)) &&
                         (reinterpret_cast<uintptr_t>(q.bias) & 15) == 0;
    if constexpr (TM == 4 && TN % 2 == 0) {
        if (q.act == GMD_ACT_GEGLU && !q.out_f32 && q.ksplit <= 1 && m0 + BM <= q.M && n0 + BN <= q.N && (q.ldc & 7) == 0 && (q.sC & 7) == 0 &&
            (reinterpret_cast<uintptr_t>(q.bias) & 15) == 0) {
            constexpr int kStripG = 32 * (TN * 8 + 4);
            epilogue_rows_geglu<HT, TM, TN>(q, acc, reinterpret_cast<float*>(smem) + wid * kStripG, m0 + wr * (TM * 16), n0 + wc * (TN * 16), lane, z);
            return;
        }
    }
    if constexpr (TM == 4) {
        if (__builtin_expect(q.ksplit > 1 && m0 + BM <= q.M && n0 + BN <= q.N && (q.N & 3) == 0, 0)) {
            constexpr int kStripS = 32 * (TN * 16 + 4);
            epilogue_rows_slab<TM, TN>(q, acc, reinterpret_cast<float*>(smem) + wid * kStripS, m0 + wr * (TM * 16), n0 + wc * (TN * 16), lane, ks);
            return;
        }
    }
    if (__builtin_expect(rows_ok, 1)) {
        constexpr int kStrip = 32 * (TN * 16 + 4);
        static_assert((size_t)NCONS * kStrip * 4 <= (size_t)NST * kStage, "epilogue strips must fit in the ring stages");
        epilogue_rows<HT, TM, TN>(q, acc, reinterpret_cast<float*>(smem) + wid * kStrip, m0 + wr * (TM * 16), n0 + wc * (TN * 16), lane, z);
    } else {
        epilogue_regs<HT, TM, TN>(q, acc, m0 + wr * (TM * 16), n0 + wc * (TN * 16), frow, fq, z, ks);
    }
}

// ------------------------------------------------------------------------------------------------
// conv3x3 (stride 1, pad 1) with the INPUT PATCH resident in LDS (round 4): the ping-pong structure of gemm_pp_kernel -- 8 consumer
// + 4 loader waves on a tile of 256 output pixels x (32 TN) channels -- but the activation operand is not re-fetched per filter tap.
//
// Why: the implicit GEMM above pulls every input pixel through the CU nine times (once per tap): 32 KB of A + 20 KB of W per 64-deep
// K step.  The two-stream pipeline is bound by what the CUs can ingest from L2 (~36 B/clk each, make_plan), so the bytes are what
// counts.  Here a tile's input patch -- its rows of pixels plus a one-pixel halo, 64 channels deep: (R+2) x (W+2) pixels x 128 B,
// at most 50 KB -- is loaded ONCE per 64-channel block into a double-buffered LDS image and all nine taps read their A fragments
// from it at shifted rows (a tap is an LDS address offset); only the weights stream per K step (20 KB): 25.6 KB per K step instead of 52.
//
//   K order: 64-channel block (outer), tap (inner): K step kt = 9 cb + tap.   LDS: [patch 0][patch 1][W ring, 3 stages].
//   patch pixel index of (image i of the tile, patch row pr, patch column pc) = (i (R+2) + pr)(W+2) + pc  <->  input (r0 + pr - 1, pc - 1);
//   out-of-image pixels are read through an out-of-range buffer offset (zeros).  Output pixel (i, y, x) and tap (ky, kx) read patch
//   pixel base(i, y, x) + ky (W+2) + kx with base = (i (R+2) + y)(W+2) + x.  A patch pixel is one 128-byte LDS row, chunk-swizzled
//   by its row index like every other tile of this file.
//   loader, per K step: the W tile of step kt+2 (ring) + a share of the NEXT block's patch (taps 0..7; tap 8 issues none, so its
//   counted vmcnt leaves only W pieces outstanding and the whole patch has landed when the block ends).
//
// Barrier / hazard structure: exactly gemm_pp_kernel's (4 barriers per K step, waves 4-7 staggered by one); the patch buffer of
// block cb+1 was last read in block cb-1, whose last reads had returned before the barrier that opens block cb.
// Tiles: 256 consecutive output pixels = whole image rows (W <= 64, 256 % W == 0) of one image, or whole images when H W < 256.
// ------------------------------------------------------------------------------------------------
// Chunk swizzle of the resident input patch (round 5).  A tap reads its A fragments at SHIFTED patch rows: 16 consecutive rows starting
// at any row r0, not at a multiple of 16.  ds_read_b128 serves a wave in four groups of 16 lanes that are NOT contiguous -- {0-3, 12-15,
// 20-27}, {4-11, 16-19, 28-31} and the same +32 (MI355X_MICROARCH.md, LDS table): a group holds fragment rows {0-3, 12-15} of one 16-byte
// chunk column c and rows {4-11} of column c ^ 1, and its 16 lanes must fall on the 16 slots of a 256-byte bank line (slot = (row & 1) * 8
// + physical chunk).  Per row parity that is eight rows u0 .. u0+7 (u = row >> 1), the middle four with the low chunk bit flipped.  The
// tile swizzle of lds_off(), chunk ^ (u & 7), is conflict-free only for u0 even (r0 = 0 mod 4: every GEMM tile, one tap position in
// four here) -- rocprofv3: SQ_LDS_BANK_CONFLICT 2.16 M of 9.29 M LDS cycles on conv 8x64x64 320->320 against 0.20 M of 7.32 M for the
// per-tap kernel.  chunk ^ (2 * (u & 3)) is conflict-free for EVERY u0: rows u and u + 4 share the two upper chunk bits, and exactly one
// of them sits in the flipped middle four.
__device__ __forceinline__ int patch_swz(int row) { return ((row >> 1) & 3) << 1; }

template <typename HT, int TN, int NPP>
__global__ __launch_bounds__(768, 3) void conv_patch_kernel(const GemmParams p) {
    GMD_WG_TRACE_SCOPE(WGK_PATCH | WGK_CONV_BIT);
    constexpr int WN = 2, NCONS = 8, LW = 4, TM = 4, NWS = 3;
    constexpr int BM = 256, BN = WN * TN * 16;
    constexpr int NWP = BN / (LW * 8);               // W pieces per loader wave and K step (5 / 4)
    constexpr int kWStage = BN * 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wuni = __builtin_amdgcn_readfirstlane(wid);
    const bool loader = wuni >= NCONS;
    const bool late = wuni >= 4 && !loader;
    int m0, n0;
    {
        const int nwg = gridDim.x, id = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, j = id >> 3;
        const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
        const int tiles_n = (p.N + BN - 1) / BN;
        const int mt = L / tiles_n;
        m0 = mt * BM;
        n0 = (L - mt * tiles_n) * BN;
    }
    const int ks = p.ksplit > 1 ? blockIdx.z : 0;
    // geometry (host-checked: stride 1, pad 1, W a power of two <= 64, 256 % W == 0, H W a multiple or a divisor of 256)
    const int H = p.Hin, W = p.Win, HW = H * W;
    const int R = HW >= BM ? BM / W : H;              // image rows per image of the tile
    const int PW = W + 2, PR = R + 2;
    const int nimg = BM / (R * W);
    const int patch_px = nimg * PR * PW;
    const int npieces = (patch_px + 7) >> 3;
    const int PB = npieces * 1024;                    // bytes of one patch buffer
    const int b0 = m0 / HW, r0 = HW >= BM ? (m0 - b0 * HW) / W : 0;
    // K range of this slice, in 64-channel blocks
    const int nblk_total = p.Cin / BK;
    const int per = (nblk_total + p.ksplit - 1) / p.ksplit;
    const int cb_begin = ks * per;
    const int ncb = (cb_begin + per <= nblk_total ? per : nblk_total - cb_begin);  // blocks of this slice (may be <= 0)
    const int nk = ncb * 9;

    auto seg_barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);
    const unsigned w_ring = 2u * (unsigned)PB;        // byte offset of the W ring behind the two patch buffers

    if (loader) {
        const int lw = wuni - NCONS;
        const int ltid = tid - NCONS * 64;
        // patch pieces of this wave: piece q = j * 4 + lw; a piece beyond the patch re-loads an early piece (same bytes to the same
        // place: harmless) so that every wave issues exactly NPP pieces per block and the counted waits below hold
        unsigned poff[NPP];
#pragma unroll
        for (int j = 0; j < NPP; ++j) {
            int q = j * LW + lw;
            if (q >= npieces) q -= npieces;
            const int pix = q * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ patch_swz(pix);
            const int i = pix / (PR * PW), rem = pix - i * (PR * PW);
            const int pr = rem / PW, pc = rem - pr * PW;
            const int iy = r0 + pr - 1, ix = pc - 1, b = b0 + i;
            const bool ok = pix < patch_px && b < p.M / HW && iy >= 0 && iy < H && ix >= 0 && ix < W;
            poff[j] = ok ? (unsigned)(((b * H + iy) * W + ix) * p.Cin) * 2u + (unsigned)chunk * 16u : kOOB;
        }
        unsigned woff[NWP];
        {
            const int srow = ltid >> 3;
            const int chunk = (ltid & 7) ^ ((srow >> 1) & 7);
#pragma unroll
            for (int i = 0; i < NWP; ++i) {
                const int n = n0 + srow + (LW * 8) * i;
                woff[i] = n < p.N ? (unsigned)n * (unsigned)p.ldw * 2u + (unsigned)chunk * 16u : kOOB;
            }
        }
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
        // patch pieces [lo, hi) of this wave for 64-channel block cb into patch buffer `buf`
        auto dma_patch = [&](auto LO, auto HI, int cb, int buf) {
            constexpr int lo = decltype(LO)::value, hi = decltype(HI)::value;
            const unsigned soff = (unsigned)(cb_begin + cb) * (BK * 2);
#pragma unroll
            for (int j = 0; j < NPP; ++j)
                if (j >= lo && j < hi) {
                    int q = j * LW + lw;
                    if (q >= npieces) q -= npieces;
                    dma16(dA, lds_base + (unsigned)buf * (unsigned)PB + (unsigned)q * 1024u, poff[j], soff);
                }
        };
        // W pieces [lo, hi) of K step (cb, tap) into ring stage `st`
        auto dma_w = [&](auto LO, auto HI, int cb, int tap, int st) {
            constexpr int lo = decltype(LO)::value, hi = decltype(HI)::value;
            const unsigned soff = (unsigned)(tap * p.Cin + (cb_begin + cb) * BK) * 2u;
            const unsigned stage = lds_base + w_ring + (unsigned)st * kWStage + (unsigned)lw * (8 * 128);
#pragma unroll
            for (int i = 0; i < NWP; ++i)
                if (i >= lo && i < hi) dma16(dW, stage + i * (LW * 8 * 128), woff[i], soff);
        };
        // patch pieces issued at tap t of a block (for the NEXT block): 2,2,2,2,2,1,1,1,0 (NPP = 13) / 2,2,2,1,1,1,1,1,0 (NPP = 11)
        if (nk > 0) {
            dma_patch(IntC<0>{}, IntC<NPP>{}, 0, 0);
            dma_w(IntC<0>{}, IntC<NWP>{}, 0, 0, 0);
            dma_w(IntC<0>{}, IntC<NWP>{}, 0, 1, 1);
            wait_vmcnt<NWP>();  // the patch of block 0 and the W tile of step 0 have landed
            seg_barrier();      // #0
            int st_fill = 2;
            for (int cb = 0; cb < ncb; ++cb) {
                const bool next_blk = cb + 1 < ncb;
                const int nbuf = (cb + 1) & 1;
                auto kstep = [&](auto TAP) {
                    constexpr int t = decltype(TAP)::value;
                    constexpr int n2 = NPP == 13 ? 5 : 3;                 // taps that issue two patch pieces
                    constexpr int pc = t < n2 ? 2 : (t < 8 ? 1 : 0);      // patch pieces issued at this tap
                    constexpr int pj = t < n2 ? 2 * t : n2 + t;           // index of the first of them
                    const int kt = cb * 9 + t;
                    // the W tile of step kt+2: (cb, t+2) or the next block's (t+2-9)
                    const bool more = kt + 2 < nk;
                    const int wcb = t + 2 < 9 ? cb : cb + 1, wtap = t + 2 < 9 ? t + 2 : t + 2 - 9;
                    if (more) dma_w(IntC<0>{}, IntC<(NWP + 3) / 4>{}, wcb, wtap, st_fill);
                    if (next_blk && pc >= 1) dma_patch(IntC<pj>{}, IntC<pj + 1>{}, cb + 1, nbuf);
                    seg_barrier();
                    if (more) dma_w(IntC<(NWP + 3) / 4>{}, IntC<(NWP + 3) / 4 + (NWP + 2) / 4>{}, wcb, wtap, st_fill);
                    if (next_blk && pc >= 2) dma_patch(IntC<pj + 1>{}, IntC<pj + 2>{}, cb + 1, nbuf);
                    seg_barrier();
                    if (more) dma_w(IntC<(NWP + 3) / 4 + (NWP + 2) / 4>{}, IntC<(NWP + 3) / 4 + (NWP + 2) / 4 + (NWP + 1) / 4>{}, wcb, wtap, st_fill);
                    seg_barrier();
                    if (more) dma_w(IntC<(NWP + 3) / 4 + (NWP + 2) / 4 + (NWP + 1) / 4>{}, IntC<NWP>{}, wcb, wtap, st_fill);
                    // everything issued BEFORE this K step has landed (the W tile of step kt+1; at tap 8, pc == 0: the whole patch
                    // of the next block): only this step's own pieces may remain
                    if (more && next_blk) wait_vmcnt<NWP + pc>();
                    else if (more) wait_vmcnt<NWP>();
                    else wait_vmcnt<0>();
                    seg_barrier();
                    st_fill = st_fill == NWS - 1 ? 0 : st_fill + 1;
                };
                kstep(IntC<0>{}); kstep(IntC<1>{}); kstep(IntC<2>{}); kstep(IntC<3>{}); kstep(IntC<4>{});
                kstep(IntC<5>{}); kstep(IntC<6>{}); kstep(IntC<7>{}); kstep(IntC<8>{});
            }
            seg_barrier();  // the trailing barrier of the early consumer group
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
    int prow[TM];  // patch pixel of this lane's output pixel of 16-row tile i at tap (0, 0)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int ml = wr * 64 + i * 16 + frow;
        const int img = ml / (R * W), rem = ml - img * (R * W);
        const int y = rem / W, x = rem - y * W;
        prow[i] = (img * PR + y) * PW + x;
    }
    uint4 fa[TM], fb[TN];
    auto frag_reads = [&](int pbuf, int tapoff, int wst, int s2) {
        const unsigned char* sP = smem + pbuf * PB;
        const unsigned char* sW = smem + w_ring + wst * kWStage;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = prow[i] + tapoff;
            fa[i] = *reinterpret_cast<const uint4*>(sP + row * 128 + (((4 * s2 + fq) ^ patch_swz(row)) << 4));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const uint4*>(sW + lds_off(wc * (TN * 16) + j * 16 + frow, 4 * s2 + fq));
    };
    auto mfmas = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = Half<HT>::mfma16(fb[j], fa[i], acc[i][j]);
    };
    auto lgkm0 = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
    if (nk > 0) {
        seg_barrier();                  // #0
        if (late) seg_barrier();        // the stagger
        int wst = 0, pbuf = 0, tap = 0, kx = 0, tapoff = 0;
        for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                frag_reads(pbuf, tapoff, wst, s2);
                lgkm0();
                seg_barrier();
                __builtin_amdgcn_s_setprio(1);
                mfmas();
                __builtin_amdgcn_s_setprio(0);
                seg_barrier();
            }
            wst = wst == NWS - 1 ? 0 : wst + 1;
            ++tap;
            if (tap == 9) { tap = 0; kx = 0; tapoff = 0; pbuf ^= 1; }
            else if (kx == 2) { kx = 0; tapoff += PW - 2; }
            else { ++kx; ++tapoff; }
        }
        if (!late) seg_barrier();
    }
    const int z = 0;
    __syncthreads();
    GemmParams q = p;
    if (p.fixup) {  // in-kernel split-K reduction: the producer slices leave here, the last slice goes on as an unsplit launch
        if (!splitk_fixup<TM, TN, NCONS>(p, acc, ks, wid, lane)) return;
        q.ksplit = 1;
    }
    const bool rows_ok = !q.out_f32 && q.ksplit <= 1 && m0 + BM <= q.M && n0 + BN <= q.N && (q.ldc & 7) == 0 &&
                         (q.residual == nullptr || (q.ldr & 7) == 0) &&
                         (q.rowbias == nullptr || ((q.ldrb & 3) == 0 && (reinterpret_cast<uintptr_t>(q.rowbias) & 15) == 0)) &&
                         (reinterpret_cast<uintptr_t>(q.bias) & 15) == 0;
    if (__builtin_expect(q.ksplit > 1 && m0 + BM <= q.M && n0 + BN <= q.N && (q.N & 3) == 0, 0)) {
        constexpr int kStripS = 32 * (TN * 16 + 4);
        epilogue_rows_slab<TM, TN>(q, acc, reinterpret_cast<float*>(smem) + wid * kStripS, m0 + wr * 64, n0 + wc * (TN * 16), lane, ks);
        return;
    }
    if (__builtin_expect(rows_ok, 1)) {
        constexpr int kStrip = 32 * (TN * 16 + 4);
        epilogue_rows<HT, TM, TN>(q, acc, reinterpret_cast<float*>(smem) + wid * kStrip, m0 + wr * 64, n0 + wc * (TN * 16), lane, z);
    } else {
        epilogue_regs<HT, TM, TN>(q, acc, m0 + wr * 64, n0 + wc * (TN * 16), frow, fq, z, ks);
    }
}

// Same kernel with CONTINUOUS consumers: every consumer wave keeps both 32-deep half steps of a K step in registers (two fragment
// sets), issues all their reads up front and multiplies; ONE barrier per K step, no stagger.  (The ping-pong form above spends
// ~130 cycles of barrier skew in each of its four intervals per K step: tools/pp_diag.py.)
template <typename HT, int TN, int NPP>
__global__ __launch_bounds__(768, 3) void conv_patch_cont_kernel(const GemmParams p) {
    GMD_WG_TRACE_SCOPE(WGK_PATCH_CONT | WGK_CONV_BIT);
    constexpr int WN = 2, NCONS = 8, LW = 4, TM = 4, NWS = 3;
    constexpr int BM = 256, BN = WN * TN * 16;
    constexpr int NWP = BN / (LW * 8);               // W pieces per loader wave and K step (5 / 4)
    constexpr int kWStage = BN * 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wuni = __builtin_amdgcn_readfirstlane(wid);
    const bool loader = wuni >= NCONS;
    int m0, n0;
    {
        const int nwg = gridDim.x, id = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, j = id >> 3;
        const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
        const int tiles_n = (p.N + BN - 1) / BN;
        const int mt = L / tiles_n;
        m0 = mt * BM;
        n0 = (L - mt * tiles_n) * BN;
    }
    const int ks = p.ksplit > 1 ? blockIdx.z : 0;
    // geometry (host-checked: stride 1, pad 1, W a power of two <= 64, 256 % W == 0, H W a multiple or a divisor of 256)
    const int H = p.Hin, W = p.Win, HW = H * W;
    const int R = HW >= BM ? BM / W : H;              // image rows per image of the tile
    const int PW = W + 2, PR = R + 2;
    const int nimg = BM / (R * W);
    const int patch_px = nimg * PR * PW;
    const int npieces = (patch_px + 7) >> 3;
    const int PB = npieces * 1024;                    // bytes of one patch buffer
    const int b0 = m0 / HW, r0 = HW >= BM ? (m0 - b0 * HW) / W : 0;
    // K range of this slice, in 64-channel blocks
    const int nblk_total = p.Cin / BK;
    const int per = (nblk_total + p.ksplit - 1) / p.ksplit;
    const int cb_begin = ks * per;
    const int ncb = (cb_begin + per <= nblk_total ? per : nblk_total - cb_begin);  // blocks of this slice (may be <= 0)
    const int nk = ncb * 9;

    auto seg_barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);
    const unsigned w_ring = 2u * (unsigned)PB;        // byte offset of the W ring behind the two patch buffers

    if (loader) {
        const int lw = wuni - NCONS;
        const int ltid = tid - NCONS * 64;
        // patch pieces of this wave: piece q = j * 4 + lw; a piece beyond the patch re-loads an early piece (same bytes to the same
        // place: harmless) so that every wave issues exactly NPP pieces per block and the counted waits below hold
        unsigned poff[NPP];
#pragma unroll
        for (int j = 0; j < NPP; ++j) {
            int q = j * LW + lw;
            if (q >= npieces) q -= npieces;
            const int pix = q * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ patch_swz(pix);
            const int i = pix / (PR * PW), rem = pix - i * (PR * PW);
            const int pr = rem / PW, pc = rem - pr * PW;
            const int iy = r0 + pr - 1, ix = pc - 1, b = b0 + i;
            const bool ok = pix < patch_px && b < p.M / HW && iy >= 0 && iy < H && ix >= 0 && ix < W;
            poff[j] = ok ? (unsigned)(((b * H + iy) * W + ix) * p.Cin) * 2u + (unsigned)chunk * 16u : kOOB;
        }
        unsigned woff[NWP];
        {
            const int srow = ltid >> 3;
            const int chunk = (ltid & 7) ^ ((srow >> 1) & 7);
#pragma unroll
            for (int i = 0; i < NWP; ++i) {
                const int n = n0 + srow + (LW * 8) * i;
                woff[i] = n < p.N ? (unsigned)n * (unsigned)p.ldw * 2u + (unsigned)chunk * 16u : kOOB;
            }
        }
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
        // patch pieces [lo, hi) of this wave for 64-channel block cb into patch buffer `buf`
        auto dma_patch = [&](auto LO, auto HI, int cb, int buf) {
            constexpr int lo = decltype(LO)::value, hi = decltype(HI)::value;
            const unsigned soff = (unsigned)(cb_begin + cb) * (BK * 2);
#pragma unroll
            for (int j = 0; j < NPP; ++j)
                if (j >= lo && j < hi) {
                    int q = j * LW + lw;
                    if (q >= npieces) q -= npieces;
                    dma16(dA, lds_base + (unsigned)buf * (unsigned)PB + (unsigned)q * 1024u, poff[j], soff);
                }
        };
        // W pieces [lo, hi) of K step (cb, tap) into ring stage `st`
        auto dma_w = [&](auto LO, auto HI, int cb, int tap, int st) {
            constexpr int lo = decltype(LO)::value, hi = decltype(HI)::value;
            const unsigned soff = (unsigned)(tap * p.Cin + (cb_begin + cb) * BK) * 2u;
            const unsigned stage = lds_base + w_ring + (unsigned)st * kWStage + (unsigned)lw * (8 * 128);
#pragma unroll
            for (int i = 0; i < NWP; ++i)
                if (i >= lo && i < hi) dma16(dW, stage + i * (LW * 8 * 128), woff[i], soff);
        };
        // patch pieces issued at tap t of a block (for the NEXT block): 2,2,2,2,2,1,1,1,0 (NPP = 13) / 2,2,2,1,1,1,1,1,0 (NPP = 11)
        if (nk > 0) {
            dma_patch(IntC<0>{}, IntC<NPP>{}, 0, 0);
            dma_w(IntC<0>{}, IntC<NWP>{}, 0, 0, 0);
            dma_w(IntC<0>{}, IntC<NWP>{}, 0, 1, 1);
            wait_vmcnt<NWP>();  // the patch of block 0 and the W tile of step 0 have landed
            seg_barrier();      // #0
            int st_fill = 2;
            for (int cb = 0; cb < ncb; ++cb) {
                const bool next_blk = cb + 1 < ncb;
                const int nbuf = (cb + 1) & 1;
                auto kstep = [&](auto TAP) {
                    constexpr int t = decltype(TAP)::value;
                    constexpr int n2 = NPP == 13 ? 5 : 3;                 // taps that issue two patch pieces
                    constexpr int pc = t < n2 ? 2 : (t < 8 ? 1 : 0);      // patch pieces issued at this tap
                    constexpr int pj = t < n2 ? 2 * t : n2 + t;           // index of the first of them
                    const int kt = cb * 9 + t;
                    // the W tile of step kt+2: (cb, t+2) or the next block's (t+2-9)
                    const bool more = kt + 2 < nk;
                    const int wcb = t + 2 < 9 ? cb : cb + 1, wtap = t + 2 < 9 ? t + 2 : t + 2 - 9;
                    if (more) dma_w(IntC<0>{}, IntC<NWP>{}, wcb, wtap, st_fill);
                    if (next_blk && pc >= 1) dma_patch(IntC<pj>{}, IntC<pj + pc>{}, cb + 1, nbuf);
                    // everything issued BEFORE this K step has landed (the W tile of step kt+1; at tap 8, pc == 0: the whole patch
                    // of the next block): only this step's own pieces may remain
                    if (more && next_blk) wait_vmcnt<NWP + pc>();
                    else if (more) wait_vmcnt<NWP>();
                    else wait_vmcnt<0>();
                    seg_barrier();  // B(kt+1): publishes W tile kt+1, frees the stage of tile kt
                    st_fill = st_fill == NWS - 1 ? 0 : st_fill + 1;
                };
                kstep(IntC<0>{}); kstep(IntC<1>{}); kstep(IntC<2>{}); kstep(IntC<3>{}); kstep(IntC<4>{});
                kstep(IntC<5>{}); kstep(IntC<6>{}); kstep(IntC<7>{}); kstep(IntC<8>{});
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
    int prow[TM];  // patch pixel of this lane's output pixel of 16-row tile i at tap (0, 0)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int ml = wr * 64 + i * 16 + frow;
        const int img = ml / (R * W), rem = ml - img * (R * W);
        const int y = rem / W, x = rem - y * W;
        prow[i] = (img * PR + y) * PW + x;
    }
    uint4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
    auto frag_reads = [&](int pbuf, int tapoff, int wst, int s2, uint4 (&fa)[TM], uint4 (&fb)[TN]) {
        const unsigned char* sP = smem + pbuf * PB;
        const unsigned char* sW = smem + w_ring + wst * kWStage;
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const uint4*>(sW + lds_off(wc * (TN * 16) + j * 16 + frow, 4 * s2 + fq));
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = prow[i] + tapoff;
            fa[i] = *reinterpret_cast<const uint4*>(sP + row * 128 + (((4 * s2 + fq) ^ patch_swz(row)) << 4));
        }
    };
    auto mfmas = [&](const uint4 (&fa)[TM], const uint4 (&fb)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = Half<HT>::mfma16(fb[j], fa[i], acc[i][j]);
    };
    if (nk > 0) {
        seg_barrier();                  // B(0): the patch of block 0 and W tile 0 are in LDS
        int wst = 0, pbuf = 0, tap = 0, kx = 0, tapoff = 0;
        for (int kt = 0; kt < nk; ++kt) {
            frag_reads(pbuf, tapoff, wst, 0, fa0, fb0);
            frag_reads(pbuf, tapoff, wst, 1, fa1, fb1);   // the second half step's reads return under the first one's MFMAs
            mfmas(fa0, fb0);
            mfmas(fa1, fb1);
            seg_barrier();              // B(kt+1)
            wst = wst == NWS - 1 ? 0 : wst + 1;
            ++tap;
            if (tap == 9) { tap = 0; kx = 0; tapoff = 0; pbuf ^= 1; }
            else if (kx == 2) { kx = 0; tapoff += PW - 2; }
            else { ++kx; ++tapoff; }
        }
    }
    const int z = 0;
    __syncthreads();
    GemmParams q = p;
    if (p.fixup) {  // in-kernel split-K reduction: the producer slices leave here, the last slice goes on as an unsplit launch
        if (!splitk_fixup<TM, TN, NCONS>(p, acc, ks, wid, lane)) return;
        q.ksplit = 1;
    }
    const bool rows_ok = !q.out_f32 && q.ksplit <= 1 && m0 + BM <= q.M && n0 + BN <= q.N && (q.ldc & 7) == 0 &&
                         (q.residual == nullptr || (q.ldr & 7) == 0) &&
                         (q.rowbias == nullptr || ((q.ldrb & 3) == 0 && (reinterpret_cast<uintptr_t>(q.rowbias) & 15) == 0)) &&
                         (reinterpret_cast<uintptr_t>(q.bias) & 15) == 0;
    if (__builtin_expect(q.ksplit > 1 && m0 + BM <= q.M && n0 + BN <= q.N && (q.N & 3) == 0, 0)) {
        constexpr int kStripS = 32 * (TN * 16 + 4);
        epilogue_rows_slab<TM, TN>(q, acc, reinterpret_cast<float*>(smem) + wid * kStripS, m0 + wr * 64, n0 + wc * (TN * 16), lane, ks);
        return;
    }
    if (__builtin_expect(rows_ok, 1)) {
        constexpr int kStrip = 32 * (TN * 16 + 4);
        epilogue_rows<HT, TM, TN>(q, acc, reinterpret_cast<float*>(smem) + wid * kStrip, m0 + wr * 64, n0 + wc * (TN * 16), lane, z);
    } else {
        epilogue_regs<HT, TM, TN>(q, acc, m0 + wr * 64, n0 + wc * (TN * 16), frow, fq, z, ks);
    }
}

// ------------------------------------------------------------------------------------------------
// float32 FMA kernel (parity path): 64x64x16 tiles, 4x4 outputs per thread
// ------------------------------------------------------------------------------------------------
template <bool CONV>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmParams p) {
    GMD_WG_TRACE_SCOPE(WGK_GEMM_F32);
    constexpr int BM = 64, BN = 64, FK = 16, LD = 68;
    __shared__ __attribute__((aligned(16))) float As[FK][LD];
    __shared__ __attribute__((aligned(16))) float Ws[FK][LD];
    const int tid = threadIdx.x;
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM, z = blockIdx.z;
    const float* A = (const float*)p.A + (int64_t)z * p.sA;
    const float* W = (const float*)p.W + (int64_t)z * p.sW;
    const int srow = tid >> 2, chunk = tid & 3;
    const RowCtx ra = make_row<CONV>(p, m0 + srow);
    const int wn = n0 + srow;
    const int64_t wbase = wn < p.N ? (int64_t)wn * p.ldw : -1;
    const int ty = tid >> 4, tx = tid & 15;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    int tap = 0, c0 = 0;
    for (int k0 = 0; k0 < p.K; k0 += FK) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const int64_t off = a_offset<CONV>(p, ra, k0, ky, kx, c0);
        const bool kin = k0 + chunk * 4 < p.K;  // K only needs to be a multiple of 4: the tail chunks read as zero
        const float4 av = (off >= 0 && kin) ? *reinterpret_cast<const float4*>(A + off + chunk * 4) : make_float4(0, 0, 0, 0);
        const float4 wv = (wbase >= 0 && kin) ? *reinterpret_cast<const float4*>(W + wbase + k0 + chunk * 4) : make_float4(0, 0, 0, 0);
        if (CONV) {
            c0 += FK;
            if (c0 >= p.Cin) { c0 = 0; ++tap; }
        }
        __syncthreads();  // previous tile fully consumed
        As[chunk * 4 + 0][srow] = av.x; As[chunk * 4 + 1][srow] = av.y; As[chunk * 4 + 2][srow] = av.z; As[chunk * 4 + 3][srow] = av.w;
        Ws[chunk * 4 + 0][srow] = wv.x; Ws[chunk * 4 + 1][srow] = wv.y; Ws[chunk * 4 + 2][srow] = wv.z; Ws[chunk * 4 + 3][srow] = wv.w;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < FK; ++k) {
            const float4 a4 = *reinterpret_cast<const float4*>(&As[k][ty * 4]);
            const float4 b4 = *reinterpret_cast<const float4*>(&Ws[k][tx * 4]);
            const float a[4] = {a4.x, a4.y, a4.z, a4.w}, b[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty * 4 + i;
        if (m >= p.M) continue;
        const float* rb = p.rowbias ? p.rowbias + (int64_t)(m / p.rows_per_group) * p.ldrb : nullptr;
        const float* res = p.residual ? (const float*)p.residual + (int64_t)z * p.sR + (int64_t)m * p.ldr : nullptr;
        float* o = (float*)p.C + (int64_t)z * p.sC + (int64_t)m * p.ldc;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n >= p.N) continue;
            float x = acc[i][j] * p.alpha;
            if (p.bias) x += p.bias[n];
            if (rb) x += rb[n];
            if (res) x += res[n];
            o[n] = apply_act(x, p.act);
        }
    }
}

struct Plan {
    int bm, bn, pf, ksplit;
};

// Channel block of the ring kernel's conv3x3 K order.  The tiles resident on one XCD (64 = 32 CUs x 2 workgroups, a
// contiguous run of the n-fastest tile order) read the same input rows once per filter tap; walking ALL channels of a tap
// before the next tap makes the re-read distance rows x Cin x 2 bytes, which for Cin >= 640 at 64x64 (5.2 MB) no longer
// fits the XCD's 4 MiB L2 (rocprofv3 FETCH_SIZE: 9.1x the algorithmic reads on 8x64x64 640->320,
// profiles/r01_pmc_conv_attention_current.txt).  Blocks of `cblk` channels bring the distance back under 3 MB
// (rocprofv3 after: 1.97x, 125.5 -> 118.8 us; profiles/r02_pmc_conv_gemm_traffic.txt).  Shapes whose rows already fit keep
// the tap-major order (cblk == Cin): at 2.6 MB (32x32, Cin 1280) the blocked order measured slower, not faster.
bool tuning_enabled();
int conv_channel_block(int B, int Hin, int Win, int Cin, int Cout, int dtype) {
    if (dtype == GMD_F32) return Cin;
    if (tuning_enabled()) {  // experiments only (GMD_TUNING=1): GMD_CONV_CBLK=<multiple of 64 dividing Cin>
        const char* e = getenv("GMD_CONV_CBLK");
        const int v = e ? atoi(e) : 0;
        if (v >= 64 && v % 64 == 0 && Cin % v == 0) return v;
    }
    const int64_t rows_total = (int64_t)B * Hin * Win;
    const int tiles_n = (Cout + 159) / 160;
    const int64_t rows_resident = (int64_t)(64 / tiles_n > 0 ? 64 / tiles_n : 1) * 128;  // input rows under one XCD's resident tiles
    const int64_t rows = rows_total / 8 < rows_resident ? (rows_total + 7) / 8 : rows_resident;
    const int64_t budget = 3ll << 20;
    if (rows * Cin * 2 <= budget) return Cin;
    int best = 64;
    for (int d = 64; d < Cin; d += 64)
        if (Cin % d == 0 && rows * d * 2 <= budget) best = d;
    return best;
}

// Kernel-tuning override (debug only: tools/, never the product path).  It takes effect only in a process that has
// GMD_TUNING=1 in its environment -- both the GMD_GEMM_FORCE seed read when the library is loaded and the in-process
// gmd_gemm_plan_override() -- and every forced plan still passes plan_unsupported() below, the ONE place that refuses a
// kernel lacking the epilogue a launch asks for (round 2: five memory-access faults of tools/bench_graph_ops.py under forced
// odd-TN ring tiles, whose plain epilogue stored [M, N] into the [M, N/2] output of the fused-GEGLU projection).
bool tuning_enabled() {
    const char* t = getenv("GMD_TUNING");
    return t && t[0] == '1';
}
struct Force {
    int bm = 0, bn = 0, pf = 0, ks = 0;
    Force() {
        if (!tuning_enabled()) return;
        if (const char* f = getenv("GMD_GEMM_FORCE")) sscanf(f, "%d,%d,%d,%d", &bm, &bn, &pf, &ks);
    }
};
Force g_force;  // read once when the library is loaded
// GMD_PP=0 keeps the round-3 plans (A/B measurements of whole runs; read once when the library is loaded)
const bool g_pp_enabled = [] { const char* e = getenv("GMD_PP"); return !(e && e[0] == '0'); }();
// How stride-1 convolutions on 256-row ping-pong tiles fetch their activations: 0 (default since the end of round 5) = per-tap implicit
// GEMM (gemm_pp_kernel<CONV>); 2 = input patch resident in LDS, continuous consumers (conv_patch_cont_kernel); 1 = patch resident,
// ping-pong consumers (conv_patch_kernel).  GMD_CONV_PATCH seeds it when the library is loaded; gmd_conv_patch_override() changes it
// in-process for A/B runs and tests (GMD_TUNING=1 only).  Whole-run A/Bs: round 4 (launch-by-launch plans) 838.7 / 840.6 / 837.5 ms for
// 0 / 1 / 2 -- level, and the patch forms pull half the bytes from L2 (25.6 instead of 52 KB per K step), so 2 became the default; at
// the end of round 5 (co-running plan family, in-kernel reduction, the shorter epilogue) the per-tap kernel is 1.0 % FASTER on the wall
// of the two-stream pipeline (764.1 -> 756.4 ms, 4 of 4 interleaved rounds; 765.1 -> 756.8, 2 of 2), level with the streams serialised
// (922.1 / 921.0) and at batch 8 (1266.6 / 1265.2), and level or ahead launch by launch (8x64x64 320->320 61.0 -> 58.4 us, 640->320
// 113.7 -> 111.4): profiles/r05_ab_conv_patch_mode.txt.  The patch kernels stay in the library (tests, A/B).
int g_conv_patch_mode = [] { const char* e = getenv("GMD_CONV_PATCH"); return (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 0; }();
// Plan family (round 5, see make_plan): 0 = the plan that is fastest launch by launch when the launch has the chip to itself (direct
// calls, single-stream pipelines, the VAE); 1 = the co-running family -- 256-row tiles everywhere, filled up with K slices -- for
// launches that share the chip with a second stream's kernels (the dual-UNet pipeline's two forwards).  The calling thread selects
// it with gmd_gemm_plan_family(); GMD_PP=b / GMD_PP=1 pin family 1 / 0 for the whole process (A/B runs), read once at load time.
const int g_family_pin = [] { const char* e = getenv("GMD_PP"); return (e && e[0] == 'b') ? 1 : ((e && e[0] == '1') ? 0 : -1); }();
thread_local int t_plan_family = 0;
// K slices of the co-running family: until about g_f1_target workgroups, at least g_f1_min_steps K steps each (GMD_F1_TARGET /
// GMD_F1_MIN_STEPS: whole-run A/B only).  The target was 256 -- one workgroup per CU -- until the end of round 5; HALF the chip is
// better: 256 -> 128: 761.5 -> 744.5 ms per batch (-2.2 %, 3 of 3 interleaved rounds), 1320.4 -> 1307.3 at batch 8 (-1.0 %); 160 / 144:
// 747.7 / 746.3; 112: 766.8 (the 80-tile projections lose their second slice); 96 / 64 / 32: 768.9 / 780.3 / 813.8; 320: +4.4 %
// (profiles/r05_ab_f1_slice_target.txt).  Fewer slices = fewer prologues, fragment round trips and finisher waits per product: CU-time,
// which is what two forwards sharing the chip pay for; the other stream fills the CUs a launch leaves free.
const int g_f1_target = [] { const char* e = getenv("GMD_F1_TARGET"); return e ? atoi(e) : 128; }();
const int g_f1_min_steps = [] { const char* e = getenv("GMD_F1_MIN_STEPS"); return e ? atoi(e) : 8; }();
const bool g_f1_bn128 = [] { const char* e = getenv("GMD_F1_BN128"); return !(e && e[0] == '0'); }();  // GMD_F1_BN128=0: A/B
// In-kernel split-K reduction up to this many K slices (0 = off: slabs + reduction launch everywhere).  GMD_SPLITK_FIXUP=<n>, read once.
int g_fixup_max = [] { const char* e = getenv("GMD_SPLITK_FIXUP"); return e ? atoi(e) : 4; }();  // gmd_splitk_fixup_max() changes it in-process
inline bool big_tiles() { return (g_family_pin >= 0 ? g_family_pin : t_plan_family) == 1; }

// Tile / split-K selection.  All SD-1.5 channel widths (320, 640, 1280, 2560, 5120, 10240) are multiples of
// 160, the VAE widths (128, 256, 512) of 128.  Launches that would leave most of the 256 CUs idle and have a
// deep K (the 8x8 / 16x16 UNet levels: K up to 23040) are split along K.
Plan make_plan(int M, int N, int K, int batch, int64_t ws_bytes, bool pair_tiles, bool want_cs = false) {
    Plan pl{64, 64, 0, 1};  // pf 0 = LDS-DMA pipeline (fastest measured); 1/2 = register-staged fallbacks
    if (M >= 96 && N >= 96) {
        pl.bm = 128;
        pl.bn = (N % 160 == 0 && !pair_tiles) ? 160 : 128;  // GEGLU needs an even number of 16-column tiles per wave
    }
    // GMD_GEMM_FORCE="bm,bn,pf,ksplit" (0 = keep heuristic): tuning experiments only (tools/bench_gemm.py).  Parsed ONCE per
    // process
    // -- the launch path itself never touches the environment; gmd_gemm_plan_override() changes it in-process for A/B runs.
    const int fbm = g_force.bm, fbn = g_force.bn, fpf = g_force.pf, fks = g_force.ks;
    if (fbm && fbn) { pl.bm = fbm; pl.bn = fbn; }
    if (fpf) pl.pf = fpf == 9 ? 0 : fpf;  // 9 selects the LDS-DMA pipeline (pf 0)
    int64_t tiles = (int64_t)((M + pl.bm - 1) / pl.bm) * ((N + pl.bn - 1) / pl.bn) * batch;
    const int nk = K / BK;
    if (batch == 1 && tiles < 160 && nk >= 24 && !pair_tiles) {  // K >= 1536: the slab reduction (a second launch) must pay for itself
        // whole waves of blocks: 512 (two per CU) when the tile grid is at least a quarter of the chip, else 256 (the
        // fp32 slab traffic of more slices costs more than the second resident block buys) -- measured, tools/bench_gemm.py
        int ks = (int)(((tiles >= 64 ? 512 : 256) + tiles / 2) / tiles);
        if (ks > nk / 8) ks = nk / 8;  // at least 8 K steps (512 channels) per slice
        if (ks > 16) ks = 16;
        if (ks > 1 && (int64_t)ks * M * N * (int64_t)sizeof(float) <= ws_bytes) pl.ksplit = ks;
    }
    // one workgroup per CU (224..256 large tiles) leaves every SIMD with a single wave; with a very deep K (>= 10240: the
    // 32x32 up-block convolutions over concatenated inputs) two slices -- two workgroups per CU -- pay for the slab reduction
    // (tools/bench_graph_ops.py: 1280->640 149.9 -> 137.0 us, 1920->640 208.6 -> 181.1 us; 640->640, K = 5760, loses)
    if (batch == 1 && pl.bm == 128 && tiles >= 224 && tiles <= 256 && nk >= 160 && !pair_tiles &&
        2 * (int64_t)M * N * (int64_t)sizeof(float) <= ws_bytes)
        pl.ksplit = 2;
    // batched, operand-swapped projections (V^T[b] = W_v x_b^T: M = channels <= 640, N = tokens): 128-row tiles leave a
    // ragged third row block at M = 320 and lose to 64x64 tiles even at M = 640 (tools/bench_vt.py: 27.9 -> 18.4 us, 17.2 -> 15.7 us)
    if (pl.ksplit == 1 && batch > 1 && M <= 640 && pl.bm == 128 && !pair_tiles && !(fbm && fbn)) {
        pl.bm = 64;
        pl.bn = 64;
    }
    if (pl.ksplit == 1 && tiles < 256 && pl.bm == 128 && !(fbm && fbn)) {  // cannot fill the chip: 4-5x more, smaller tiles
        pl.bm = 64;
        pl.bn = 64;
    }
    if (fks) pl.ksplit = ((int64_t)fks * M * N * (int64_t)sizeof(float) <= ws_bytes && batch == 1) ? fks : 1;
    // Round-4 kernels: one workgroup per CU with dedicated LDS-DMA loader waves (gemm_pp_kernel, code 283: 256-row tiles;
    // gemm_lc_kernel, code 244: 128- / 64-row tiles for launches with about one tile per CU).
    //
    // DEFAULT POLICY -- the plan that is fastest launch by launch (tools/sweep_pp.py --round3, device time inside a HIP graph: -11 % /
    // -15 % summed over the UNet's linear + convolution launches at batch 8 / 4 against the round-3 plans; every row of the vendor-
    // library yardstick, tools/vs_library_gemm.py):
    //   * ping-pong kernel where there are >= 256 tiles of 256 x 160 (every level-0 linear / convolution at batch 8), or of 256 x 128
    //     where N is not a multiple of 160 (VAE decoder); the GEGLU projection (value | gate pairs) from K = 1280 up, or K = 640 with
    //     at least 8192 rows;
    //   * loader / consumer kernel where 128- or 64-row tiles give 200...256 workgroups (with K slices where K is deep): conv 32x32
    //     640->640 at batch 8 72.8 -> 55.5 us, 64x64 320->320 at batch 4 41.3 -> 33.0 us, linear M=2048 N=1280 K=5120 49.4 -> 36.8 us.
    //     More than 256 such tiles would run in two rounds of one workgroup per CU, fewer than ~200 leave CUs idle: both keep the
    //     plans above.
    // CO-RUNNING FAMILY (gmd_gemm_plan_family(1); what the dual-UNet pipeline selects for its two overlapped forwards) -- the largest
    // tile, filled up with K slices: slower launch by launch, faster on the wall of the two-stream pipeline in every interleaved
    // whole-run A/B (profiles/r05_ab_plan_default.txt, five rounds on one box: 867.7 -> 849.0 ms per batch at batch 4 (-2.2 %, 5 of 5),
    // 1454.3 -> 1436.9 at batch 8 (-1.2 %, 3 of 3); round 4, three boxes: -0.4 ... -1.3 %).  With two streams in flight a second
    // workgroup is always there to hide a kernel's own latencies, so L2 -> LDS bytes per product -- (1/BM + 1/BN) x 2 B: 0.020 for a
    // 256 x 160 tile, 0.028 for 128 x 160, 0.044 for 64 x 160 -- weigh more than the launch's time alone on the chip.  With the
    // streams serialised the same family LOSES 8.6 % (1018.3 -> 1106.3 ms), so it is never the choice of a launch that runs alone:
    // family 0 stays the default of the C ABI.  GMD_PP=0: the round-3 plans.
    if (g_pp_enabled && batch == 1 && !(fbm && fbn) && !fpf && !fks && M >= 64) {
        const int64_t mt256 = (M + 255) / 256;
        const int bn = N % 160 == 0 ? 160 : (N % 128 == 0 ? 128 : 0);
        if (big_tiles()) {
            if (pair_tiles) {
                if (M >= 256 && N % 128 == 0) pl = Plan{256, 128, 283, 1};  // GEGLU pairs value / gate tiles: no K slices
            } else if (bn && M >= 256) {
                auto slices = [&](int bnc) {
                    const int64_t t = mt256 * (N / bnc);
                    int ks = t >= g_f1_target ? 1 : (int)((g_f1_target + t / 2) / t);
                    if (ks > 8) ks = 8;
                    while (ks > 1 && (nk / ks < g_f1_min_steps || (int64_t)ks * M * N * (int64_t)sizeof(float) > ws_bytes)) --ks;
                    return ks;
                };
                int bnc = bn;
                // 128-column tiles where they still fit ONE round of workgroups: their row segments are whole 128-byte lines (a
                // 160-column tile shares every third line of a row with its neighbour) and there are a quarter more of them
                // (not for a launch that is to emit GroupNorm statistics: their 10-channel buckets need 80-column wave tiles)
                if (g_f1_bn128 && !want_cs && bn == 160 && N % 128 == 0 && mt256 * (N / 128) * slices(128) <= 256) bnc = 128;
                pl = Plan{256, bnc, 283, slices(bnc)};
            }
        } else if (pair_tiles) {
            if (M >= 256 && N % 128 == 0 && ((nk >= 20 && M >= 512) || (nk >= 10 && M >= 8192))) pl = Plan{256, 128, 283, 1};
        } else if (bn && M >= 256 && mt256 * (N / bn) >= 256) {
            pl = Plan{256, bn, 283, 1};
        } else if (bn) {
            bool found = false;
            for (int bm = 128; bm >= 64 && !found; bm >>= 1) {  // unsplit first: the larger tile wins when both fill the chip
                const int64_t t = (int64_t)((M + bm - 1) / bm) * (N / bn);
                if (t >= 200 && t <= 256) { pl = Plan{bm, bn, 244, 1}; found = true; }
            }
            // 64-row tiles pull 28 KB per K step through the CU for half the products of a 128-row tile (36 KB): with a deep K two
            // slices of 128-row tiles beat them (tools/sweep_lc.py, conv 16x16 1280->1280 at batch 8: 78.2 -> 66.9 us, 2560->1280:
            // 155 -> 121 us; at K = 5760 the slab reduction costs more than it buys: 39.4 vs 42.1 us)
            if (found && pl.bm == 64 && nk >= 144 && 2 * (int64_t)M * N * (int64_t)sizeof(float) <= ws_bytes) {
                const int64_t t = (int64_t)((M + 127) / 128) * (N / bn) * 2;
                if (t >= 200 && t <= 256) pl = Plan{128, bn, 244, 2};
            }
            for (int bm = 128; bm >= 64 && !found; bm >>= 1) {  // K slices: at least 20 K steps (K = 1280) each
                const int64_t t = (int64_t)((M + bm - 1) / bm) * (N / bn);
                for (int ks = 2; ks <= 8 && !found; ++ks)
                    if (t * ks >= 200 && t * ks <= 256 && nk / ks >= 20 && (int64_t)ks * M * N * (int64_t)sizeof(float) <= ws_bytes) {
                        pl = Plan{bm, bn, 244, ks};
                        found = true;
                    }
            }
            // still nothing (the GM UNet's 16x16 projections at batch 4, M = 1024 N = K = 1280: 128 tiles of 64 x 160, K too short to
            // slice): 64 x 128 loader / consumer tiles, a quarter more workgroups than 64 x 160 (tools/dbg/sweep_rows.py: 13.7 us on the
            // legacy 64 x 64 kernel -> 11.3 us; the vendor library 11.5)
            if (!found && !want_cs && N % 128 == 0 && nk >= 8) {  // (validated for K >= 512 only)
                const int64_t t = (int64_t)((M + 63) / 64) * (N / 128);
                if (t >= 144 && t <= 256) pl = Plan{64, 128, 244, 1};
            }
        }
    }
    return pl;
}

template <typename HT, bool CONV, int BM, int BN, int PF>
hipError_t launch_bf16(const GemmParams& p, int gz, hipStream_t s) {
    constexpr size_t smem = 2 * (BM + BN) * 128;
    if (smem > 64 * 1024) {  // > 64 KiB of dynamic LDS must be opted into once per (kernel, device)
        hipError_t e = opt_in_lds(reinterpret_cast<const void*>(&gemm_bf16_kernel<HT, CONV, BM, BN, PF>), (int)smem);
        if (e != hipSuccess) return e;
    }
    dim3 grid(((p.N + BN - 1) / BN) * ((p.M + BM - 1) / BM), 1, gz);  // 1-D tile index, remapped per XCD in the kernel
    gemm_bf16_kernel<HT, CONV, BM, BN, PF><<<grid, 256, smem, s>>>(p);
    return hipGetLastError();
}

template <typename HT, bool CONV, int WM, int WN, int TN, int NST>
hipError_t launch_ring(const GemmParams& p, int gz, hipStream_t s) {
    constexpr int BM = WM * 64, BN = WN * TN * 16;
    constexpr size_t smem = (size_t)NST * (BM + BN) * 128;
    if (smem > 64 * 1024) {
        hipError_t e = opt_in_lds(reinterpret_cast<const void*>(&gemm_ring_kernel<HT, CONV, WM, WN, TN, NST>), (int)smem);
        if (e != hipSuccess) return e;
    }
    dim3 grid(((p.N + BN - 1) / BN) * ((p.M + BM - 1) / BM), 1, gz);  // 1-D tile index, remapped per XCD in the kernel
    gemm_ring_kernel<HT, CONV, WM, WN, TN, NST><<<grid, WM * WN * 64, smem, s>>>(p);
    return hipGetLastError();
}

template <typename HT, bool CONV, int TM, int TN>
hipError_t launch_lc(const GemmParams& p, int gz, hipStream_t s) {
    constexpr int BM = 2 * TM * 16, BN = 2 * TN * 16;
    constexpr size_t smem = (size_t)4 * (BM + BN) * 128;
    hipError_t e = opt_in_lds(reinterpret_cast<const void*>(&gemm_lc_kernel<HT, CONV, TM, TN>), (int)smem);
    if (e != hipSuccess) return e;
    dim3 grid(((p.N + BN - 1) / BN) * ((p.M + BM - 1) / BM), 1, gz);
    gemm_lc_kernel<HT, CONV, TM, TN><<<grid, 512, smem, s>>>(p);
    return hipGetLastError();
}

// conv_patch_kernel: stride-1 / pad-1 convolutions whose 256-pixel tiles are whole image rows (or whole images) -- every UNet level
bool conv_patch_ok(const GemmParams& p) {
    const int H = p.Hin, W = p.Win, HW = H * W;
    if (p.stride != 1 || p.upsample || p.pad_lo != 1 || p.Hout != H || p.Wout != W) return false;
    if (W < 8 || W > 64 || (W & (W - 1)) || p.Cin % BK || p.M % HW) return false;
    if (HW >= 256 ? (HW % 256 != 0) : (256 % HW != 0)) return false;
    const int R = HW >= 256 ? 256 / W : H, nimg = 256 / (R * W);
    return nimg * (R + 2) * (W + 2) <= 400;
}

template <typename HT, int TN>
hipError_t launch_conv_patch(const GemmParams& p, int gz, hipStream_t s) {
    constexpr int BM = 256, BN = 2 * TN * 16;
    const int H = p.Hin, W = p.Win, HW = H * W;
    const int R = HW >= 256 ? 256 / W : H, nimg = 256 / (R * W);
    const int npieces = (nimg * (R + 2) * (W + 2) + 7) / 8;
    const size_t smem = (size_t)2 * npieces * 1024 + (size_t)3 * BN * 128;
    const size_t strips = (size_t)8 * 32 * (TN * 16 + 4) * 4;  // the epilogue strips reuse the K-loop image
    const size_t need = smem > strips ? smem : strips;
    dim3 grid(((p.N + BN - 1) / BN) * ((p.M + BM - 1) / BM), 1, gz);
    hipError_t e;
    if (g_conv_patch_mode == 2) {
        if (npieces > 44) {
            e = opt_in_lds(reinterpret_cast<const void*>(&conv_patch_cont_kernel<HT, TN, 13>), 163840);
            if (e != hipSuccess) return e;
            conv_patch_cont_kernel<HT, TN, 13><<<grid, 768, need, s>>>(p);
        } else {
            e = opt_in_lds(reinterpret_cast<const void*>(&conv_patch_cont_kernel<HT, TN, 11>), 163840);
            if (e != hipSuccess) return e;
            conv_patch_cont_kernel<HT, TN, 11><<<grid, 768, need, s>>>(p);
        }
        return hipGetLastError();
    }
    if (npieces > 44) {
        e = opt_in_lds(reinterpret_cast<const void*>(&conv_patch_kernel<HT, TN, 13>), 163840);
        if (e != hipSuccess) return e;
        conv_patch_kernel<HT, TN, 13><<<grid, 768, need, s>>>(p);
    } else {
        e = opt_in_lds(reinterpret_cast<const void*>(&conv_patch_kernel<HT, TN, 11>), 163840);
        if (e != hipSuccess) return e;
        conv_patch_kernel<HT, TN, 11><<<grid, 768, need, s>>>(p);
    }
    return hipGetLastError();
}

template <typename HT, bool CONV, int TN>
hipError_t launch_pp(const GemmParams& p, int gz, hipStream_t s) {
    constexpr int BM = 256, BN = 2 * TN * 16;
    constexpr size_t smem = (size_t)3 * (BM + BN) * 128;
    hipError_t e = opt_in_lds(reinterpret_cast<const void*>(&gemm_pp_kernel<HT, CONV, TN>), (int)smem);
    if (e != hipSuccess) return e;
    dim3 grid(((p.N + BN - 1) / BN) * ((p.M + BM - 1) / BM), 1, gz);
    gemm_pp_kernel<HT, CONV, TN><<<grid, 768, smem, s>>>(p);
    return hipGetLastError();
}

// One 16-bit element type (bf16_t or f16_t): plan, kernel choice, split-K reduction.  float16 instantiates the kernels the
// heuristic actually picks; the register-staged and deeper-ring tuning variants exist for bfloat16 only (plan overrides).
// Column statistics (GemmParams::colstats) come out of the row epilogue of the default ring kernels only: every tile must be a
// full tile of a single, unsplit launch whose waves own 64 rows x (BN/2) columns, a whole number of buckets.
// A split-K launch reduces inside the kernel (splitk_fixup) when this holds; `ws_bytes` = usable workspace bytes.  The one predicate
// of the launch itself (launch_half) and of the plan queries that depend on it (column statistics).
bool fixup_plan_ok(const Plan& pl, int M, int N, int64_t ws_bytes, bool defer_reduce) {
    if (pl.ksplit <= 1 || pl.ksplit > g_fixup_max || defer_reduce || ws_bytes <= 0 || !(pl.pf == 283 || pl.pf == 244)) return false;
    const int64_t tiles = (int64_t)((N + pl.bn - 1) / pl.bn) * ((M + pl.bm - 1) / pl.bm);
    const int64_t frag_bytes = (int64_t)(pl.ksplit - 1) * tiles * pl.bm * pl.bn * 4;
    return tiles <= kFixupCounters && frag_bytes <= ws_bytes && frag_bytes < 0xFFFF0000LL;
}

// (split launches: the finisher of the in-kernel reduction runs the row epilogue of an unsplit launch, statistics included)
bool colstats_plan_ok(const Plan& pl, int M, int N, int batch, int bucket, int64_t ws_bytes) {
    // all three: waves own 64 rows x (BN/2) columns
    const bool ring = pl.pf == 0 && pl.bm == 128, pp = pl.pf == 283 && pl.bm == 256, lc = pl.pf == 244 && pl.bm == 128;
    return (ring || pp || lc) && (pl.bn == 160 || pl.bn == 128) && (pl.ksplit == 1 || fixup_plan_ok(pl, M, N, ws_bytes, false)) && batch == 1 &&
           bucket > 0 && M % pl.bm == 0 && N % pl.bn == 0 && (pl.bn / 2) % bucket == 0;
}

// The one place that refuses a plan (heuristic or forced) whose kernel lacks an epilogue the launch asks for; nullptr = fine.
//   * GEGLU pairs value / gate tiles of 16 columns inside a wave: only kernels with an EVEN number of column tiles per wave
//     implement it (128x128 and 64x64 register / DMA kernels, ring tiles with TN = 2 or 4) and never with split-K -- any
//     other kernel's plain epilogue would store [M, N] into the [M, N/2] output;
//   * column statistics come out of the full-tile row epilogue of the two default 128-row ring kernels only.
const char* plan_unsupported(const Plan& pl, const GemmParams& p, int batch, int64_t ws_bytes) {
    if (p.act == GMD_ACT_GEGLU) {
        const bool odd_tn = pl.bn == 160 || (pl.pf >= 100 && pl.pf != 283 && pl.bm == 64 && pl.bn == 64);  // TN = 5 / ring<1,4,1,.>: TN = 1
        if (odd_tn || pl.ksplit > 1 || p.out_f32 || (pl.pf == 244 && pl.bm != 128)) return "has no GEGLU epilogue";
    }
    if (p.colstats) {
        const bool rows_ok = !p.out_f32 && p.act != GMD_ACT_GEGLU && (p.ldc & 7) == 0 && (p.residual == nullptr || (p.ldr & 7) == 0) &&
                             (p.rowbias == nullptr || ((p.ldrb & 3) == 0 && (reinterpret_cast<uintptr_t>(p.rowbias) & 15) == 0));
        if (!rows_ok || p.defer_reduce || !colstats_plan_ok(pl, p.M, p.N, batch, p.cs_bucket, ws_bytes))
            return "cannot emit column statistics (they need the full-tile row epilogue of an unsplit 128-row ring launch: ask "
                   "gmd_gemm_colstats_plan first)";
    }
    return nullptr;
}

// Tile order of the ping-pong / loader-consumer kernels (GemmParams::tile_group).  Workgroups with equal id % 8 share an XCD and get
// a contiguous run of tiles; with n fastest an XCD walks whole M-panels, so between two uses of a weight tile lie all the others:
//   * weights larger than the activations (N > M: the GEGLU projections of the 16x16 / 8x8 levels, W up to 26 MB): m fastest --
//     each weight tile then goes to ONE XCD instead of all eight (rocprofv3 FETCH_SIZE, M=2048 N=10240 K=1280: 226 MB = 7.2x the
//     algorithmic reads with n fastest);
//   * activations larger, but the weights do not fit an XCD's 4 MiB L2 beside them (M=8192 N=5120 K=640: W = 6.5 MB, 231 MB fetched
//     = 13.6x): the XCD's M-panels are walked together, one N tile at a time, so every weight tile is fetched once per XCD.
// Everything else keeps n fastest.
int pick_tile_group(const Plan& pl, int M, int N, int K) {
    if (pl.pf != 283 && pl.pf != 244) return 1;
    const int tiles_m = (M + pl.bm - 1) / pl.bm;
    const int64_t w_bytes = (int64_t)N * K * 2;
    int g;
    if (N > M) g = tiles_m;
    else if (w_bytes <= (3ll << 20)) return 1;
    else g = (tiles_m + 7) / 8;  // the XCD's share of M-panels
    // Round 5: at most FOUR M-panels per group.  An XCD walks its run of tiles one W panel at a time over the group's A panels: the
    // A panels must survive in its 4 MiB L2 beside the W panels in flight and the output lines passing through, the W panels are
    // streamed once.  rocprofv3 FETCH_SIZE against the group (profiles/r05_pmc_tile_group.txt, MB read, algorithmic in brackets):
    // M=4096 N=5120 K=640 [11.8]: 16 (the N > M rule) 126, 8: 50, 4: 37.5, 2: 59;  M=2048 N=10240 K=1280 [31.5]: 8 (N > M) 145, 4: 110,
    // 2: 130;  M=8192 N=5120 K=640 [17]: 4 (the share rule) 66, 8: 84, 2: 121;  M=1024 N=10240 K=1280: 4 = all its panels, 63.
    if (g > 4) g = 4;
    return g < 1 ? 1 : g;
}

// fused Q|K|V projection with transposed V tiles (GemmParams::vt_out, epilogue_cols_vt): every tile full and through the row epilogue,
// the V columns starting on a tile boundary, wave tiles (64 or 32 rows) inside one sample
bool qkv_vt_plan_ok(const Plan& pl, int M, int N, int batch, int vt_col0, int vt_tokens) {
    return batch == 1 && pl.ksplit == 1 && M % pl.bm == 0 && N % pl.bn == 0 && vt_col0 > 0 && vt_col0 < N && vt_col0 % pl.bn == 0 &&
           vt_tokens > 0 && vt_tokens % 64 == 0 && M % vt_tokens == 0;
}

template <typename HT, bool CONV>
int launch_half(GemmParams p, int batch, void* ws, int64_t ws_bytes, hipStream_t s, const char* name) {
    constexpr bool kTune = std::is_same<HT, bf16_t>::value;
    hipError_t e = hipSuccess;
    const Plan pl = make_plan(p.M, p.N, p.K, batch, ws ? ws_bytes : 0, p.act == GMD_ACT_GEGLU, p.colstats != nullptr);
    if (p.vt_out && !qkv_vt_plan_ok(pl, p.M, p.N, batch, p.vt_col0, p.vt_tokens)) {
        gmd_set_error("%s: plan %dx%d ksplit=%d cannot write transposed V tiles (ask gmd_gemm_qkv_vt_ok first)", name, pl.bm, pl.bn, pl.ksplit);
        return GMD_ERR_UNSUPPORTED;
    }
    if (const char* why = plan_unsupported(pl, p, batch, ws ? ws_bytes : 0)) {
        gmd_set_error("%s: plan %dx%d pf=%d ksplit=%d (M=%d N=%d bucket=%d) %s", name, pl.bm, pl.bn, pl.pf, pl.ksplit, p.M, p.N, p.cs_bucket, why);
        return GMD_ERR_UNSUPPORTED;
    }
    p.ksplit = pl.ksplit;
    p.ws = (float*)ws;
    p.tile_group = CONV ? 1 : pick_tile_group(pl, p.M, p.N, p.K);  // (convolutions: neighbouring M-panels share their halo rows)
    const int gz = pl.ksplit > 1 ? pl.ksplit : batch;
    // In-kernel split-K reduction (splitk_fixup) instead of slabs + splitk_reduce_kernel: the round-4 kernels (one workgroup per CU,
    // wave tiles in registers), up to g_fixup_max slices (the finisher reads the other slices' fragments one after the other), the
    // consumer of the slabs not being a fused GroupNorm (defer_reduce).  Bit-identical to the slab path (same order of additions).
    p.fixup = 0;
    if (fixup_plan_ok(pl, p.M, p.N, ws ? ws_bytes : 0, p.defer_reduce != 0)) {
        const int64_t tiles = (int64_t)((p.N + pl.bn - 1) / pl.bn) * ((p.M + pl.bm - 1) / pl.bm);
        p.fixup = 1;
        p.fix_bytes = (unsigned)((int64_t)(pl.ksplit - 1) * tiles * pl.bm * pl.bn * 4);
        p.fix_cnt = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws) + ws_bytes);  // the tail behind the usable bytes
    }
    bool done = false;
    if (pl.pf == 244) {  // loader / consumer kernel: 4 consumer + 4 loader waves, 4-stage ring, one workgroup per CU
        if (pl.bm == 128 && pl.bn == 160) e = launch_lc<HT, CONV, 4, 5>(p, gz, s);
        else if (pl.bm == 128 && pl.bn == 128) e = launch_lc<HT, CONV, 4, 4>(p, gz, s);
        else if (pl.bm == 64 && pl.bn == 160) e = launch_lc<HT, CONV, 2, 5>(p, gz, s);
        else if (pl.bm == 64 && pl.bn == 128) e = launch_lc<HT, CONV, 2, 4>(p, gz, s);
        else { gmd_set_error("%s: loader/consumer tile %dx%d is not instantiated", name, pl.bm, pl.bn); return GMD_ERR_UNSUPPORTED; }
        done = true;
    } else if (CONV && pl.pf == 283 && g_conv_patch_mode != 0 && conv_patch_ok(p) && (int64_t)pl.ksplit <= p.Cin / BK) {
        // ping-pong structure with the input patch resident in LDS (conv_patch_kernel): same tiles, same epilogues, same plan code
        if (pl.bn == 160) e = launch_conv_patch<HT, 5>(p, gz, s);
        else if (pl.bn == 128) e = launch_conv_patch<HT, 4>(p, gz, s);
        else { gmd_set_error("%s: ping-pong tile %dx%d is not instantiated", name, pl.bm, pl.bn); return GMD_ERR_UNSUPPORTED; }
        done = true;
    } else if (pl.pf == 283) {  // ping-pong kernel: 8 consumer + 4 loader waves on a 256-row tile
        if (pl.bm == 256 && pl.bn == 160) e = launch_pp<HT, CONV, 5>(p, gz, s);
        else if (pl.bm == 256 && pl.bn == 128) e = launch_pp<HT, CONV, 4>(p, gz, s);
        else { gmd_set_error("%s: ping-pong tile %dx%d is not instantiated", name, pl.bm, pl.bn); return GMD_ERR_UNSUPPORTED; }
        done = true;
    } else if constexpr (kTune) {
        done = true;
        // pf 1xx selects a ring kernel (experiments): 1WS with W = waves-in-M (2|4), S = stages
        if (pl.pf == 143 && pl.bn == 160) e = launch_ring<HT, CONV, 4, 2, 5, 3>(p, gz, s);
        else if (pl.pf == 143 && pl.bn == 128) e = launch_ring<HT, CONV, 4, 2, 4, 3>(p, gz, s);
        else if (pl.pf == 123 && pl.bn == 160) e = launch_ring<HT, CONV, 2, 2, 5, 3>(p, gz, s);
        else if (pl.pf == 124 && pl.bn == 160) e = launch_ring<HT, CONV, 2, 2, 5, 4>(p, gz, s);
        else if (pl.pf == 122 && pl.bn == 160) e = launch_ring<HT, CONV, 2, 2, 5, 2>(p, gz, s);
        else if (pl.pf == 122 && pl.bn == 128) e = launch_ring<HT, CONV, 2, 2, 4, 2>(p, gz, s);
        else if (pl.pf == 123 && pl.bn == 128) e = launch_ring<HT, CONV, 2, 2, 4, 3>(p, gz, s);
        else if (pl.pf == 124 && pl.bn == 128) e = launch_ring<HT, CONV, 2, 2, 4, 4>(p, gz, s);
        else if (pl.pf == 103 && pl.bm == 64 && pl.bn == 64) e = launch_ring<HT, CONV, 1, 4, 1, 3>(p, gz, s);
        else if (pl.pf == 104 && pl.bm == 64 && pl.bn == 64) e = launch_ring<HT, CONV, 1, 4, 1, 4>(p, gz, s);
        else if (pl.pf == 103 && pl.bm == 64 && pl.bn == 128) e = launch_ring<HT, CONV, 1, 4, 2, 3>(p, gz, s);
        else if (pl.pf == 104 && pl.bm == 64 && pl.bn == 128) e = launch_ring<HT, CONV, 1, 4, 2, 4>(p, gz, s);
        else if (pl.pf >= 100) { gmd_set_error("%s: ring variant %d not instantiated for BN=%d", name, pl.pf, pl.bn); return GMD_ERR_UNSUPPORTED; }
        else if (pl.pf != 0 && pl.bm == 128 && pl.bn == 160)
            e = pl.pf == 1 ? launch_bf16<HT, CONV, 128, 160, 1>(p, gz, s) : launch_bf16<HT, CONV, 128, 160, 2>(p, gz, s);
        else if (pl.pf != 0 && pl.bm == 128 && pl.bn == 128)
            e = pl.pf == 1 ? launch_bf16<HT, CONV, 128, 128, 1>(p, gz, s) : launch_bf16<HT, CONV, 128, 128, 2>(p, gz, s);
        else if (pl.pf != 0 && !(pl.bm == 128))
            e = pl.pf == 1 ? launch_bf16<HT, CONV, 64, 64, 1>(p, gz, s) : launch_bf16<HT, CONV, 64, 64, 2>(p, gz, s);
        else done = false;
    } else if (pl.pf != 0 && pl.pf != 283 && pl.pf != 244) {
        gmd_set_error("%s: plan override pf=%d is instantiated for bfloat16 only", name, pl.pf);
        return GMD_ERR_UNSUPPORTED;
    }
    if (!done) {
        // default: two-stage LDS-DMA ring, 4 waves, two workgroups per CU (fastest of all variants measured on MI355X);
        // 64x64 LDS-DMA tiles for launches that cannot put 256 of the large tiles on the chip
        if (pl.bm == 128 && pl.bn == 160) e = launch_ring<HT, CONV, 2, 2, 5, 2>(p, gz, s);
        else if (pl.bm == 128 && pl.bn == 128) e = launch_ring<HT, CONV, 2, 2, 4, 2>(p, gz, s);
        else if (pl.bm == 64 && pl.bn == 64) e = launch_bf16<HT, CONV, 64, 64, 0>(p, gz, s);
        else { gmd_set_error("%s: tile %dx%d is not instantiated", name, pl.bm, pl.bn); return GMD_ERR_UNSUPPORTED; }
    }
    if (e == hipSuccess && pl.ksplit > 1 && !p.defer_reduce && !p.fixup) {
        const int64_t total = (int64_t)p.M * ((p.N + 7) / 8);
        int64_t g = (total + 255) / 256;
        if (g > 4096) g = 4096;
        splitk_reduce_kernel<HT><<<(int)g, 256, 0, s>>>(p);
        e = hipGetLastError();
    }
    if (e != hipSuccess) {
        gmd_set_error("%s: launch failed: %s", name, hipGetErrorString(e));
        return GMD_ERR_LAUNCH;
    }
    return GMD_OK;
}

template <bool CONV>
int launch(GemmParams p, int dtype, int batch, void* ws, int64_t ws_bytes, hipStream_t s, const char* name) {
    if (dtype == GMD_BF16) return launch_half<bf16_t, CONV>(p, batch, ws, ws_bytes, s, name);
    if (dtype == GMD_F16) return launch_half<f16_t, CONV>(p, batch, ws, ws_bytes, s, name);
    if (p.colstats) {
        gmd_set_error("%s: column statistics are implemented for the 16-bit types only", name);
        return GMD_ERR_UNSUPPORTED;
    }
    p.ksplit = 1;
    p.ws = nullptr;
    dim3 grid((p.N + 63) / 64, (p.M + 63) / 64, batch);
    gemm_f32_kernel<CONV><<<grid, 256, 0, s>>>(p);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        gmd_set_error("%s: launch failed: %s", name, hipGetErrorString(e));
        return GMD_ERR_LAUNCH;
    }
    return GMD_OK;
}

}  // namespace

extern "C" {

int gmd_gemm_plan_override(int bm, int bn, int pf, int ksplit) {
    GMD_REQUIRE(bm >= 0 && bn >= 0 && pf >= 0 && ksplit >= 0, "gmd_gemm_plan_override: negative value");
    if (!tuning_enabled() && (bm | bn | pf | ksplit) != 0) {
        gmd_set_error("gmd_gemm_plan_override: kernel-tuning overrides are a debug facility; set GMD_TUNING=1 in the environment to use them");
        return GMD_ERR_UNSUPPORTED;
    }
    g_force.bm = bm; g_force.bn = bn; g_force.pf = pf; g_force.ks = ksplit;
    // the float32 matrix-core path has its own planner (gemm_split.hip); of an override it takes the kernel family only:
    // pf 244 = its loader / converter kernel wherever instantiated, anything else = its default (the in-register split)
    gmd_split_set_lc(pf == 244 ? 1 : 0);
    return GMD_OK;
}

int gmd_gemm_plan_family(int family) {
    const int prev = t_plan_family;
    if (family == 0 || family == 1) t_plan_family = family;  // anything else: query only
    return prev;
}

int gmd_splitk_fixup_max(int max_slices) {
    const int prev = g_fixup_max;
    if (max_slices >= 0) g_fixup_max = max_slices > 16 ? 16 : max_slices;
    return prev;
}

int gmd_conv_patch_override(int mode) {
    GMD_REQUIRE(mode >= 0 && mode <= 2, "gmd_conv_patch_override: mode 0, 1 or 2");
    if (!tuning_enabled()) {
        gmd_set_error("gmd_conv_patch_override: kernel-tuning overrides are a debug facility; set GMD_TUNING=1 in the environment to use them");
        return GMD_ERR_UNSUPPORTED;
    }
    g_conv_patch_mode = mode;
    return GMD_OK;
}

int gmd_gemm_colstats_plan(int dtype, int M, int N, int K, int batch, int64_t workspace_bytes, int bucket) {
    workspace_bytes = gmd_ws_usable_bytes(workspace_bytes);  // the tail of the workspace holds the split-K arrival counters
    if (gmd_is_split(dtype)) return gmd_split_colstats_ok(M, N, K, batch, workspace_bytes, bucket);  // round 4
    if (!gmd_is_half(dtype) || M <= 0 || N <= 0 || K <= 0 || K % BK != 0) return 0;
    return colstats_plan_ok(make_plan(M, N, K, batch, workspace_bytes, false, true), M, N, batch, bucket, workspace_bytes) ? 1 : 0;
}

int gmd_gemm_plan_info(int dtype, int M, int N, int K, int batch, int64_t workspace_bytes, int geglu, int* out4) {
    workspace_bytes = gmd_ws_usable_bytes(workspace_bytes);  // the tail of the workspace holds the split-K arrival counters
    GMD_REQUIRE(gmd_is_half(dtype) && M > 0 && N > 0 && K > 0 && K % BK == 0 && batch > 0 && out4, "gmd_gemm_plan_info: 16-bit launches only");
    const Plan pl = make_plan(M, N, K, batch, workspace_bytes, geglu != 0);
    out4[0] = pl.bm; out4[1] = pl.bn; out4[2] = pl.pf; out4[3] = pl.ksplit;
    return GMD_OK;
}

// 1 when a float32-split gmd_gemm_nt launch of these dimensions can take out_dtype = GMD_F32SA (store its result pre-split)
int gmd_gemm_out_split_ok(int M, int N, int K, int geglu, int64_t workspace_bytes) { return gmd_split_out_ok(M, N, K, geglu, gmd_ws_usable_bytes(workspace_bytes)); }

// (ws_bytes: the USABLE bytes, i.e. without the counter tail)
static int qkv_vt_ok_usable(int dtype, int M, int N, int K, int vt_col0, int vt_tokens, int64_t ws_bytes) {
    if (dtype == GMD_F32SW || dtype == GMD_F32SA) return gmd_split_qkv_vt_ok(M, N, K, vt_col0, vt_tokens, ws_bytes);
    if (!gmd_is_half(dtype) || M <= 0 || N <= 0 || K <= 0 || K % 64) return 0;
    return qkv_vt_plan_ok(make_plan(M, N, K, 1, ws_bytes, false), M, N, 1, vt_col0, vt_tokens) ? 1 : 0;
}

int gmd_gemm_qkv_vt_ok(int dtype, int M, int N, int K, int vt_col0, int vt_tokens, int64_t workspace_bytes) {
    return qkv_vt_ok_usable(dtype, M, N, K, vt_col0, vt_tokens, gmd_ws_usable_bytes(workspace_bytes));
}

int gmd_gemm_qkv_vt(const void* A, const void* W, void* C, void* Vt, int dtype, int M, int N, int K, int64_t ldc, int vt_col0, int vt_tokens,
                    int64_t vt_ld, float alpha, void* workspace, int64_t workspace_bytes, gmd_stream_t stream) {
    workspace_bytes = gmd_ws_usable_bytes(workspace_bytes);  // the tail of the workspace holds the split-K arrival counters
    const bool split = dtype == GMD_F32SW || dtype == GMD_F32SA;  // float32 tensors on the matrix cores, pre-split weights
    GMD_REQUIRE(gmd_is_half(dtype) || split, "gmd_gemm_qkv_vt: the 16-bit types and GMD_F32SW / GMD_F32SA only (dtype %d)", dtype);
    GMD_REQUIRE(M > 0 && N > 0 && K > 0 && K % (split ? 32 : 64) == 0, "gmd_gemm_qkv_vt: bad shape M=%d N=%d K=%d", M, N, K);
    GMD_REQUIRE(A && W && C && Vt && gmd_aligned16(A) && gmd_aligned16(W) && gmd_aligned16(C) && gmd_aligned16(Vt), "gmd_gemm_qkv_vt: null or unaligned pointer");
    GMD_REQUIRE(vt_col0 > 0 && vt_col0 < N && ldc >= vt_col0 && ldc % 8 == 0 && vt_tokens > 0 && M % vt_tokens == 0 && vt_ld >= vt_tokens && vt_ld % 8 == 0,
                "gmd_gemm_qkv_vt: bad V geometry (vt_col0=%d ldc=%lld tokens=%d vt_ld=%lld)", vt_col0, (long long)ldc, vt_tokens, (long long)vt_ld);
    // pre-split operands are [hi 64 B | lo 64 B] per 32-element chunk of a row (rows are K elements here): whole 128-byte chunks only
    GMD_REQUIRE(!split || ((reinterpret_cast<uintptr_t>(W) & 127) == 0 && (dtype != GMD_F32SA || (reinterpret_cast<uintptr_t>(A) & 127) == 0)),
                "gmd_gemm_qkv_vt: a pre-split operand must be 128-byte aligned (32-element chunks of [hi | lo])");
    GMD_REQUIRE(qkv_vt_ok_usable(dtype, M, N, K, vt_col0, vt_tokens, workspace ? workspace_bytes : 0), "gmd_gemm_qkv_vt: this launch cannot write transposed V tiles (ask gmd_gemm_qkv_vt_ok)");
    GemmParams p{};
    p.A = A; p.W = W; p.C = C; p.M = M; p.N = N; p.K = K;
    p.lda = K; p.ldw = K; p.ldc = ldc;
    {
        const int es = split ? 4 : 2;
        const int64_t ab = (int64_t)M * K * es, wb = (int64_t)N * K * es;
        GMD_REQUIRE(ab < 0xFFFF0000LL && wb < 0xFFFF0000LL, "gmd_gemm_qkv_vt: operand slab larger than 4 GiB");
        p.a_bytes = (unsigned)ab; p.w_bytes = (unsigned)wb;
    }
    p.rows_per_group = 1; p.ldrb = N; p.alpha = alpha; p.act = GMD_ACT_NONE; p.cblk = K;
    p.vt_out = Vt; p.vt_col0 = vt_col0; p.vt_tokens = vt_tokens; p.vt_ld = vt_ld;
    if (split) {
        p.out_f32 = 1;
        return gmd_launch_split_gemm(&p, dtype == GMD_F32SA ? 2 : 1, 1, workspace, workspace_bytes, (hipStream_t)stream, "gmd_gemm_qkv_vt");
    }
    return launch<false>(p, dtype, 1, workspace, workspace_bytes, (hipStream_t)stream, "gmd_gemm_qkv_vt");
}

int gmd_gemm_nt(const void* A, const void* W, void* C, int dtype, int out_dtype, int M, int N, int K, int64_t lda,
                int64_t ldw, int64_t ldc, int batch, int64_t strideA, int64_t strideW, int64_t strideC, const float* bias,
                const float* rowbias, int rows_per_group, int64_t ldrb, const void* residual, int64_t ldr, int64_t strideR, float alpha,
                int act, float* colstats, int colstats_bucket, void* workspace, int64_t workspace_bytes, gmd_stream_t stream) {
    workspace_bytes = gmd_ws_usable_bytes(workspace_bytes);  // the tail of the workspace holds the split-K arrival counters
    const bool split = gmd_is_split(dtype);  // float32 tensors, three float16 MFMA passes
    GMD_REQUIRE(dtype == GMD_BF16 || dtype == GMD_F16 || dtype == GMD_F32 || split, "gmd_gemm_nt: bad dtype %d", dtype);
    const bool is16 = gmd_is_half(dtype);
    const bool c_split = split && out_dtype == GMD_F32SA;  // float32 result stored pre-split for the next contraction
    GMD_REQUIRE(out_dtype == GMD_F32 || (!split && out_dtype == dtype) || c_split, "gmd_gemm_nt: out_dtype must be F32 or the input dtype");
    GMD_REQUIRE(M >= 0 && N >= 0 && K > 0 && batch >= 0, "gmd_gemm_nt: bad shape M=%d N=%d K=%d batch=%d", M, N, K, batch);
    if (c_split) {  // full 128-row tiles through the row epilogues only: the caller asks gmd_gemm_out_split_ok first
        const int nout = act == GMD_ACT_GEGLU ? N / 2 : N;
        GMD_REQUIRE(batch == 1 && nout % 32 == 0 && ldc == nout && !colstats && gmd_split_out_ok(M, N, K, act == GMD_ACT_GEGLU, workspace ? workspace_bytes : 0),
                    "gmd_gemm_nt: this launch cannot store its output pre-split (ask gmd_gemm_out_split_ok; ldc must equal the row length)");
    }
    if (M == 0 || N == 0 || batch == 0) return GMD_OK;
    const int kmul = is16 ? 64 : split ? 32 : 4, vec = is16 ? 8 : 4;
    GMD_REQUIRE(K % kmul == 0, "gmd_gemm_nt: K=%d must be a multiple of %d", K, kmul);
    GMD_REQUIRE(lda >= K && ldw >= K && (ldc >= N || act == GMD_ACT_GEGLU), "gmd_gemm_nt: leading dimension too small");
    GMD_REQUIRE(lda % vec == 0 && ldw % vec == 0 && strideA % vec == 0 && strideW % vec == 0,
                "gmd_gemm_nt: lda/ldw/strides must be multiples of %d elements", vec);
    GMD_REQUIRE(A && W && C && gmd_aligned16(A) && gmd_aligned16(W) && gmd_aligned16(C), "gmd_gemm_nt: null or unaligned pointer");
    // A pre-split operand is [hi 64 B | lo 64 B] per 32-element chunk of a ROW: a leading dimension / batch stride that is not a
    // whole number of chunks, or a view that starts inside a chunk, would be read as the wrong halves without any fault
    if (dtype == GMD_F32SW || dtype == GMD_F32SA)
        GMD_REQUIRE(ldw % 32 == 0 && strideW % 32 == 0 && (reinterpret_cast<uintptr_t>(W) & 127) == 0,
                    "gmd_gemm_nt: a pre-split W operand needs ldw and strideW multiples of 32 elements and a 128-byte aligned base");
    if (dtype == GMD_F32SA)
        GMD_REQUIRE(lda % 32 == 0 && strideA % 32 == 0 && (reinterpret_cast<uintptr_t>(A) & 127) == 0,
                    "gmd_gemm_nt: a pre-split A operand (GMD_F32SA) needs lda and strideA multiples of 32 elements and a 128-byte aligned base");
    GMD_REQUIRE(bias == nullptr || gmd_aligned16(bias), "gmd_gemm_nt: bias must be 16-byte aligned (it is read with float4 loads)");
    GMD_REQUIRE(rowbias == nullptr || rows_per_group > 0, "gmd_gemm_nt: rows_per_group must be positive");
    GMD_REQUIRE(residual == nullptr || (ldr >= N && gmd_aligned16(residual)), "gmd_gemm_nt: bad residual");
    GMD_REQUIRE(residual == nullptr || out_dtype == dtype || !is16, "gmd_gemm_nt: residual needs out_dtype == dtype");
    GMD_REQUIRE(act == GMD_ACT_NONE || act == GMD_ACT_SILU || act == GMD_ACT_GEGLU || act == GMD_ACT_QUICK_GELU, "gmd_gemm_nt: bad act %d", act);
    if (act == GMD_ACT_GEGLU) {
        GMD_REQUIRE((is16 && out_dtype == dtype) || split, "gmd_gemm_nt: GEGLU epilogue is implemented for the 16-bit types and the float32 split types");
        GMD_REQUIRE(N % 32 == 0 && ldc >= N / 2 && ldc % 4 == 0 && strideC % 4 == 0, "gmd_gemm_nt: GEGLU needs N %% 32 == 0 and ldc >= N/2");
        GMD_REQUIRE(!residual && !rowbias, "gmd_gemm_nt: GEGLU epilogue takes no residual / rowbias");
    }
    GMD_REQUIRE(batch <= 65535, "gmd_gemm_nt: batch too large");
    GemmParams p{};
    p.A = A; p.W = W; p.C = C; p.M = M; p.N = N; p.K = K;
    p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.sA = strideA; p.sW = strideW; p.sC = strideC;
    {
        const int es = is16 ? 2 : 4;
        const int64_t ab = ((int64_t)(M - 1) * lda + K) * es, wb = ((int64_t)(N - 1) * ldw + K) * es;
        GMD_REQUIRE(!(is16 || split) || (ab < 0xFFFF0000LL && wb < 0xFFFF0000LL), "gmd_gemm_nt: operand slab larger than 4 GiB");
        p.a_bytes = (unsigned)ab; p.w_bytes = (unsigned)wb;
    }
    p.bias = bias; p.rowbias = rowbias; p.rows_per_group = rows_per_group > 0 ? rows_per_group : 1; p.ldrb = ldrb > 0 ? ldrb : N;
    p.residual = residual; p.ldr = ldr; p.sR = strideR; p.alpha = alpha; p.act = act;
    p.out_f32 = out_dtype == GMD_F32 || c_split;
    p.c_split = c_split ? 1 : 0;
    p.cblk = K;
    p.colstats = colstats; p.cs_bucket = colstats_bucket;
    if (split) {
        return gmd_launch_split_gemm(&p, dtype == GMD_F32SA ? 2 : dtype == GMD_F32SW ? 1 : 0, batch, batch == 1 ? workspace : nullptr, workspace_bytes,
                                     (hipStream_t)stream, "gmd_gemm_nt");
    }
    return launch<false>(p, dtype, batch, workspace, workspace_bytes, (hipStream_t)stream, "gmd_gemm_nt");
}

}  // extern "C"

namespace {

// GroupNorm (+SiLU) of the convolution's output, fused behind a split-K plan (gmd_conv3x3_groupnorm)
struct GnTail {
    void* Ynorm;
    int G;
    float eps;
    const float* gamma;
    const float* beta;
    int silu;
};

void conv_out_shape(int Hin, int Win, int stride, int upsample, int pad_mode, int& Hout, int& Wout, int& pad_lo) {
    if (upsample) { Hout = 2 * Hin; Wout = 2 * Win; pad_lo = 1; }
    else if (pad_mode == 1) { Hout = (Hin + 1 - 3) / 2 + 1; Wout = (Win + 1 - 3) / 2 + 1; pad_lo = 0; }
    else { Hout = (Hin + 2 - 3) / stride + 1; Wout = (Win + 2 - 3) / stride + 1; pad_lo = 1; }
}

// split-K factor the conv launch will use (1 = unsplit); the same planners the launch itself calls
int conv_plan_ksplit(int dtype, int64_t M, int Cin, int Cout, int64_t ws_bytes) {
    if (gmd_is_split(dtype)) return gmd_split_plan_ksplit((int)M, Cout, 9 * Cin, ws_bytes);
    if (gmd_is_half(dtype)) return make_plan((int)M, Cout, 9 * Cin, 1, ws_bytes, false).ksplit;
    return 1;
}

int conv3x3_impl(const void* X, const void* Wt, void* Y, int dtype, int out_dtype, int B, int Hin, int Win, int Cin, int Cout,
                 int stride, int upsample, int pad_mode, const float* bias, const float* rowbias, int64_t ldrb, const void* residual,
                 float alpha, float* colstats, int colstats_bucket, void* workspace, int64_t workspace_bytes, gmd_stream_t stream,
                 const GnTail* gn) {
    const bool split = gmd_is_split(dtype);
    GMD_REQUIRE(dtype == GMD_BF16 || dtype == GMD_F16 || dtype == GMD_F32 || split, "gmd_conv3x3: bad dtype %d", dtype);
    const bool is16 = gmd_is_half(dtype);
    GMD_REQUIRE(out_dtype == GMD_F32 || (!split && out_dtype == dtype), "gmd_conv3x3: out_dtype must be F32 or the input dtype");
    GMD_REQUIRE(B >= 0 && Hin > 0 && Win > 0 && Cin > 0 && Cout > 0, "gmd_conv3x3: bad shape");
    if (B == 0) return GMD_OK;  // empty batch
    GMD_REQUIRE(stride == 1 || stride == 2, "gmd_conv3x3: stride must be 1 or 2");
    GMD_REQUIRE(!(upsample && stride != 1), "gmd_conv3x3: upsample requires stride 1");
    GMD_REQUIRE(pad_mode == 0 || (pad_mode == 1 && stride == 2 && !upsample), "gmd_conv3x3: pad_mode 1 requires stride 2");
    const int kmul = is16 ? 64 : split ? 32 : 16;
    GMD_REQUIRE(Cin % kmul == 0, "gmd_conv3x3: Cin=%d must be a multiple of %d (pad the channels)", Cin, kmul);
    GMD_REQUIRE(X && Wt && (Y || gn) && gmd_aligned16(X) && gmd_aligned16(Wt) && gmd_aligned16(Y), "gmd_conv3x3: null or unaligned pointer");
    GMD_REQUIRE(bias == nullptr || gmd_aligned16(bias), "gmd_conv3x3: bias must be 16-byte aligned (it is read with float4 loads)");
    GMD_REQUIRE(residual == nullptr || gmd_aligned16(residual), "gmd_conv3x3: unaligned residual");
    GMD_REQUIRE(residual == nullptr || out_dtype == dtype || !is16, "gmd_conv3x3: residual needs out_dtype == dtype");
    int Hout, Wout, pad_lo;
    conv_out_shape(Hin, Win, stride, upsample, pad_mode, Hout, Wout, pad_lo);
    const int64_t M = (int64_t)B * Hout * Wout;
    GMD_REQUIRE(M < (1LL << 31), "gmd_conv3x3: too many output pixels");
    GemmParams p{};
    p.A = X; p.W = Wt; p.C = Y; p.M = (int)M; p.N = Cout; p.K = 9 * Cin;
    p.lda = Cin; p.ldw = 9 * (int64_t)Cin; p.ldc = Cout;
    {
        const int es = is16 ? 2 : 4;
        const int64_t ab = (int64_t)B * Hin * Win * Cin * es, wb = (int64_t)Cout * 9 * Cin * es;
        GMD_REQUIRE(!(is16 || split) || (ab < 0xFFFF0000LL && wb < 0xFFFF0000LL), "gmd_conv3x3: tensor larger than 4 GiB");
        p.a_bytes = (unsigned)ab; p.w_bytes = (unsigned)wb;
    }
    p.bias = bias; p.rowbias = rowbias; p.rows_per_group = Hout * Wout; p.ldrb = ldrb > 0 ? ldrb : Cout;
    p.residual = residual; p.ldr = Cout; p.alpha = alpha; p.act = GMD_ACT_NONE;
    p.out_f32 = out_dtype == GMD_F32;
    p.Hin = Hin; p.Win = Win; p.Cin = Cin; p.Hout = Hout; p.Wout = Wout; p.stride = stride; p.upsample = upsample; p.pad_lo = pad_lo;
    p.cblk = conv_channel_block(B, Hin, Win, Cin, Cout, dtype);
    p.colstats = colstats; p.cs_bucket = colstats_bucket;
    int ks = 1;
    if (gn) {  // the split-K slabs stay in the workspace; the GroupNorm kernel sums them (no reduce launch, no raw tensor unless Y)
        const int act_dtype = split ? GMD_F32 : dtype;
        ks = workspace ? conv_plan_ksplit(dtype, M, Cin, Cout, workspace_bytes) : 1;
        if (ks <= 1 || colstats || out_dtype != act_dtype || !gmd_gn_from_slabs_ok(act_dtype, B, (int64_t)Hout * Wout, Cout, gn->G)) {
            gmd_set_error("gmd_conv3x3_groupnorm: this launch does not fuse (split-K factor %d; ask gmd_conv3x3_gn_fusable first)", ks);
            return GMD_ERR_UNSUPPORTED;
        }
        p.defer_reduce = 1;
    }
    const int rc = split ? gmd_launch_split_conv(&p, dtype == GMD_F32SA ? 2 : dtype == GMD_F32SW ? 1 : 0, B, workspace, workspace_bytes, (hipStream_t)stream, "gmd_conv3x3")
                         : launch<true>(p, dtype, 1, workspace, workspace_bytes, (hipStream_t)stream, "gmd_conv3x3");
    if (rc != GMD_OK || !gn) return rc;
    return gmd_launch_gn_from_slabs((const float*)workspace, ks, alpha, bias, rowbias, p.ldrb, residual, Y, gn->Ynorm, split ? GMD_F32 : dtype, B,
                                    (int64_t)Hout * Wout, Cout, gn->G, gn->eps, gn->gamma, gn->beta, gn->silu, (hipStream_t)stream);
}

}  // namespace

extern "C" {

int gmd_conv3x3(const void* X, const void* Wt, void* Y, int dtype, int out_dtype, int B, int Hin, int Win, int Cin, int Cout,
                int stride, int upsample, int pad_mode, const float* bias, const float* rowbias, int64_t ldrb, const void* residual,
                float alpha, float* colstats, int colstats_bucket, void* workspace, int64_t workspace_bytes, gmd_stream_t stream) {
    return conv3x3_impl(X, Wt, Y, dtype, out_dtype, B, Hin, Win, Cin, Cout, stride, upsample, pad_mode, bias, rowbias, ldrb, residual, alpha,
                        colstats, colstats_bucket, workspace, gmd_ws_usable_bytes(workspace_bytes), stream, nullptr);
}

int gmd_conv3x3_gn_fusable(int dtype, int B, int Hin, int Win, int Cin, int Cout, int stride, int upsample, int pad_mode, int groups,
                           int64_t workspace_bytes) {
    workspace_bytes = gmd_ws_usable_bytes(workspace_bytes);  // the tail of the workspace holds the split-K arrival counters
    const bool split = gmd_is_split(dtype);
    if (!(gmd_is_half(dtype) || split) || B <= 0 || Hin <= 0 || Win <= 0 || Cin <= 0 || Cout <= 0 || groups <= 0) return 0;
    if (!(stride == 1 || stride == 2) || (upsample && stride != 1) || !(pad_mode == 0 || (pad_mode == 1 && stride == 2 && !upsample))) return 0;
    int Hout, Wout, pad_lo;
    conv_out_shape(Hin, Win, stride, upsample, pad_mode, Hout, Wout, pad_lo);
    const int64_t M = (int64_t)B * Hout * Wout;
    if (M >= (1LL << 31) || Cin % (split ? 32 : 64)) return 0;
    // one workgroup per (sample, group) walks all slabs of its slice: below one workgroup per CU the separate, fully parallel
    // reduction wins (tools/bench_conv_gn.py: batch 4 x 32 groups loses 3-5 us per site, batch 8 wins 2-5 us)
    if ((int64_t)B * groups < 256) return 0;
    return conv_plan_ksplit(dtype, M, Cin, Cout, workspace_bytes) > 1 &&
           gmd_gn_from_slabs_ok(split ? GMD_F32 : dtype, B, (int64_t)Hout * Wout, Cout, groups) ? 1 : 0;
}

int gmd_conv3x3_groupnorm(const void* X, const void* Wt, void* Yraw, void* Ynorm, int dtype, int B, int Hin, int Win, int Cin, int Cout,
                          int stride, int upsample, int pad_mode, const float* bias, const float* rowbias, int64_t ldrb,
                          const void* residual, float alpha, int groups, float eps, const float* gamma, const float* beta, int silu,
                          void* workspace, int64_t workspace_bytes, gmd_stream_t stream) {
    workspace_bytes = gmd_ws_usable_bytes(workspace_bytes);  // the tail of the workspace holds the split-K arrival counters
    GMD_REQUIRE(Ynorm && gamma && beta && groups > 0 && gmd_aligned16(Ynorm), "gmd_conv3x3_groupnorm: null or unaligned GroupNorm argument");
    const bool split = gmd_is_split(dtype);
    const GnTail gn{Ynorm, groups, eps, gamma, beta, silu};
    return conv3x3_impl(X, Wt, Yraw, dtype, split ? GMD_F32 : dtype, B, Hin, Win, Cin, Cout, stride, upsample, pad_mode, bias, rowbias, ldrb,
                        residual, alpha, nullptr, 0, workspace, workspace_bytes, stream, &gn);
}

}  // extern "C"

GMD_WG_TRACE_SETTER(gemm)

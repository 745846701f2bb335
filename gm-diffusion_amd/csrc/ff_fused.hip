// GEGLU feed-forward of a BasicTransformerBlock as ONE kernel (diffusers FeedForward: net.0 = GEGLU(C -> 4C), net.2 =
// Linear(4C -> C); restated in oracle/unet.py BasicTransformerBlock.ff):
//
//     Y = (value * gelu_erf(gate)) @ W2^T + b2 + residual,   [value | gate] = X @ W1^T + b1
//
// The [tokens, 4C] GEGLU tensor is the largest tensor of a transformer block that exists only to be re-read by the next
// launch (83.9 MB written + 105.7 MB re-read per level-0 block at batch 8, profiles/r02_pmc_conv_gemm_traffic.txt).  Here a
// workgroup owns 128 token rows for the whole feed-forward and walks the hidden dimension in chunks of 128:
//
//     per chunk c:  H_c[128, 128]   = GEGLU(X[128, C] @ W1i[c]^T + b1i[c])      5 K steps of 64, accumulators -> registers
//                   (H_c -> LDS as a 16-bit tile: the A operand of the second product)
//                   Y[128, C]      += H_c @ W2[:, c]^T                           2 K steps of 64, accumulators stay in registers
//     epilogue:     Y + b2 + residual -> 16-bit rows, row-contiguous through LDS
//
// so H never reaches HBM.  Operands stream through a two-stage LDS ring filled by LDS-DMA exactly as in gemm_ring_kernel
// (128-byte rows, chunk index XOR (row >> 1) & 7 applied to the per-lane SOURCE address, one barrier per K step); the X tile
// of the workgroup's rows is re-streamed per chunk (it is L2-resident after the first pass; keeping it in LDS would leave no
// room for the ring).  W1 is the interleaved [value | gate] layout of the fused GEGLU epilogue (16-row groups), so a lane holds
// value column j and gate column j of the same row.  8 waves (2 x 4) = two waves per SIMD inside one workgroup per CU: the
// round-2 one-wave-per-SIMD loss does not recur.  Instantiated for C = 320 (level 0 of SD-1.5, where the weights -- 2.46 MB
// per pass -- are small against the rows; at C = 640 / 1280 the accumulators of a 128-row block do not fit the register file).
#include "gemm_shared.h"

namespace {

struct FFParams {
    const void* X;      // [M, C]
    const void* W1;     // [8C, C] interleaved value | gate rows
    const float* b1;    // [8C] interleaved alike
    const void* W2;     // [C, 4C]
    const float* b2;    // [C]
    const void* R;      // residual [M, C]
    void* Y;            // [M, C]
    int M;
    unsigned x_bytes, w1_bytes, w2_bytes;
};

template <typename HT, int C>
__global__ __launch_bounds__(512, 2) void ff_fused_kernel(const FFParams p) {
    GMD_WG_TRACE_SCOPE(WGK_FF_FUSED);
    constexpr int BM = 128, CH = 128;               // token rows per workgroup, hidden units per chunk
    constexpr int N1 = 2 * CH;                      // interleaved value | gate columns per chunk
    constexpr int K1 = C / BK, K2 = CH / BK;        // K steps of the first / second product per chunk
    constexpr int NCH = 4 * C / CH, T = K1 + K2;    // chunks, tiles per chunk
    constexpr int TM = 4, TN1 = 4, TN2 = C / 64;    // 16x16 tiles per wave: rows, GEMM1 columns (64), GEMM2 columns (C / 4)
    constexpr int RPP = 64;                         // rows per staging pass (8 waves x 8 rows)
    constexpr int NX = BM / RPP, NW1 = N1 / RPP, NW2 = C / RPP;
    static_assert(C % 64 == 0 && (4 * C) % CH == 0 && C / 4 == TN2 * 16, "C must be a multiple of 64 with C/4 a whole number of 16-column tiles");
    constexpr int kH = 2 * BM * 128;                // H tile: two K steps of [128 rows][128 B]
    constexpr int kStage = (BM + N1) * 128;         // ring stage: X tile + W1 tile (>= the W2 tile of C rows)
    static_assert(C * 128 <= kStage, "W2 tile must fit a ring stage");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const sH = smem;
    unsigned char* const sRing = smem + kH;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 2, wc = wid & 3;
    const int wuni = __builtin_amdgcn_readfirstlane(wid);
    const int m0 = blockIdx.x * BM;
    const int srow = tid >> 3;                                // 0 .. 63
    const int chunk = (tid & 7) ^ ((srow >> 1) & 7);          // swizzled SOURCE chunk
    const int frow = lane & 15, fq = lane >> 4;

    // ONE per-lane byte offset per operand (its first staging pass); the later passes (64 rows further) and the K / chunk
    // position go into the scalar soffset of the DMA
    const unsigned xoff = (unsigned)srow * (unsigned)(C * 2) + (unsigned)chunk * 16u;
    const unsigned w2off = (unsigned)srow * (unsigned)(4 * C * 2) + (unsigned)chunk * 16u;
    const unsigned xbase = (unsigned)m0 * (unsigned)(C * 2);

    const unsigned lds_ring = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)sRing);
    auto desc = [&](const void* base, unsigned bytes) {
        const uint64_t b = (uint64_t)base;
        return u32x4{(unsigned)__builtin_amdgcn_readfirstlane((unsigned)b), (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(b >> 32) & 0xffffu),
                     (unsigned)__builtin_amdgcn_readfirstlane(bytes), 0x00020000u};
    };
    const u32x4 dX = desc(p.X, p.x_bytes), dW1 = desc(p.W1, p.w1_bytes), dW2 = desc(p.W2, p.w2_bytes);
    auto dma16 = [&](const u32x4& d, unsigned lds_addr, unsigned voff, unsigned soff) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds"
                     :
                     : "v"(voff), "s"(lds_addr), "s"(d), "s"(soff)
                     : "memory", "m0");
#pragma clang diagnostic pop
    };
    // tile t = chunk c, phase ph: ph < K1 -> K step ph of the first product (X tile + W1 tile), else K step ph - K1 of the second (W2 tile)
    auto issue = [&](int t, int stage) {
        const int c = t / T, ph = t - c * T;
        const unsigned base = lds_ring + (unsigned)stage * kStage + (unsigned)wuni * (8 * 128);
        if (ph < K1) {
            const unsigned kb = (unsigned)ph * (BK * 2);
#pragma unroll
            for (int i = 0; i < NX; ++i) dma16(dX, base + i * (RPP * 128), xoff, xbase + kb + (unsigned)i * (RPP * C * 2));
            const unsigned wb = (unsigned)c * (unsigned)(N1 * C * 2) + kb;
#pragma unroll
            for (int i = 0; i < NW1; ++i) dma16(dW1, base + BM * 128 + i * (RPP * 128), xoff, wb + (unsigned)i * (RPP * C * 2));  // W1 rows are C wide too
        } else {
            const unsigned kb = (unsigned)(c * CH + (ph - K1) * BK) * 2u;
#pragma unroll
            for (int i = 0; i < NW2; ++i) dma16(dW2, base + i * (RPP * 128), w2off, kb + (unsigned)i * (RPP * 4 * C * 2));
        }
    };

    f32x4 acc2[TM][TN2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN2; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr int total = NCH * T;
    // wait for tile t, release the other stage, start the DMA of tile t + 1 into it
    auto sync_and_prefetch = [&](int t) {
        wait_vmcnt<0>();   // this wave's pieces of tile t have landed (two stages: nothing younger is in flight yet)
        __syncthreads();   // ... and every wave's; every wave is done with tile t-1 (its stage is free, H of a finished chunk is complete)
        if (t + 1 < total) issue(t + 1, (t + 1) & 1);
    };
    issue(0, 0);
    int t = 0;
    for (int c = 0; c < NCH; ++c) {
        // ---- first product: [128, 256] = X @ W1i[c]^T, accumulators live for this chunk only ----
        f32x4 acc1[TM][TN1];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN1; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int ph = 0; ph < K1; ++ph, ++t) {
            sync_and_prefetch(t);
            const unsigned char* sA = sRing + (t & 1) * kStage;
            const unsigned char* sW = sA + BM * 128;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                uint4 a[TM], b[TN1];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const uint4*>(sA + lds_off(wr * 64 + i * 16 + frow, 4 * s2 + fq));
#pragma unroll
                for (int j = 0; j < TN1; ++j) b[j] = *reinterpret_cast<const uint4*>(sW + lds_off(wc * 64 + j * 16 + frow, 4 * s2 + fq));
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN1; ++j) acc1[i][j] = Half<HT>::mfma16(b[j], a[i], acc1[i][j]);
            }
        }
        {
            // GEGLU in registers (value tile j, gate tile j + 1 of the same lane), H chunk -> LDS as the second product's A
            // operand: hidden column hc = wc*32 + (j/2)*16 + fq*4 + e lives in K step hc / 64, 16-byte chunk (hc % 64) / 8.
            // (The H tile was last read two barriers ago; the next barrier publishes this one.)
            const float* b1 = p.b1 + (size_t)c * N1 + wc * 64 + fq * 4;
#pragma unroll
            for (int j = 0; j < TN1; j += 2) {
                const float4 bv = *reinterpret_cast<const float4*>(b1 + j * 16), bg = *reinterpret_cast<const float4*>(b1 + (j + 1) * 16);
                const float bvv[4] = {bv.x, bv.y, bv.z, bv.w}, bgg[4] = {bg.x, bg.y, bg.z, bg.w};
                const int hc = wc * 32 + (j / 2) * 16 + fq * 4;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    float o4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float hv = acc1[i][j][e] + bvv[e];
                        const float g = acc1[i][j + 1][e] + bgg[e];
                        o4[e] = hv * (0.5f * g * (1.0f + fast_erf(g * 0.70710678118654752440f)));
                    }
                    const int row = wr * 64 + i * 16 + frow;
                    *reinterpret_cast<uint2*>(sH + (hc >> 6) * (BM * 128) + lds_off(row, (hc & 63) >> 3) + (fq & 1) * 8) =
                        make_uint2(Half<HT>::pack2(o4[0], o4[1]), Half<HT>::pack2(o4[2], o4[3]));
                }
            }
        }
        // ---- second product: Y[128, C] += H_c @ W2[:, c]^T ----
        for (int ks = 0; ks < K2; ++ks, ++t) {
            sync_and_prefetch(t);
            const unsigned char* st = sRing + (t & 1) * kStage;
            const unsigned char* sA = sH + ks * (BM * 128);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                uint4 a[TM], b[TN2];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const uint4*>(sA + lds_off(wr * 64 + i * 16 + frow, 4 * s2 + fq));
#pragma unroll
                for (int j = 0; j < TN2; ++j) b[j] = *reinterpret_cast<const uint4*>(st + lds_off(wc * (TN2 * 16) + j * 16 + frow, 4 * s2 + fq));
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN2; ++j) acc2[i][j] = Half<HT>::mfma16(b[j], a[i], acc2[i][j]);
            }
        }
    }
    __syncthreads();  // every wave is done with the last tile: the output strips overwrite the ring

    // ---- epilogue: Y = acc2 + b2 + residual, row-contiguous through a per-wave LDS strip (gemm.hip epilogue_rows) ----
    constexpr int NCOL = TN2 * 16, ROWF = NCOL + 4, CHK = NCOL / 8;
    constexpr int ITER = (32 * CHK + 63) / 64;
    static_assert((size_t)8 * 32 * ROWF * 4 <= (size_t)2 * kStage, "epilogue strips must fit in the ring");
    float* strip = reinterpret_cast<float*>(sRing) + wid * (32 * ROWF);
    const int mw = m0 + wr * 64, nw = wc * NCOL;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int j = 0; j < TN2; ++j)
                *reinterpret_cast<float4*>(strip + (i2 * 16 + frow) * ROWF + j * 16 + fq * 4) =
                    make_float4(acc2[2 * h + i2][j][0], acc2[2 * h + i2][j][1], acc2[2 * h + i2][j][2], acc2[2 * h + i2][j][3]);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < ITER; ++t) {
            const int idx = lane + 64 * t;
            const int r = idx / CHK, cc = idx - r * CHK;
            if (32 * CHK % 64 != 0 && r >= 32) continue;
            const int m = mw + h * 32 + r, n = nw + cc * 8;
            const float4 a0 = *reinterpret_cast<const float4*>(strip + r * ROWF + cc * 8);
            const float4 a1 = *reinterpret_cast<const float4*>(strip + r * ROWF + cc * 8 + 4);
            float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            float add[8];
            const uint4 w = *reinterpret_cast<const uint4*>((const HT*)p.R + (int64_t)m * C + n);
            const unsigned ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) Half<HT>::unpack2(ww[e], add[2 * e], add[2 * e + 1]);
            const float4 t0 = *reinterpret_cast<const float4*>(p.b2 + n), t1 = *reinterpret_cast<const float4*>(p.b2 + n + 4);
            const float bz[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] * 1.0f + bz[e] + add[e];  // the association of gemm.hip's epilogues (alpha = 1)
            *reinterpret_cast<uint4*>((HT*)p.Y + (int64_t)m * C + n) =
                make_uint4(Half<HT>::pack2(v[0], v[1]), Half<HT>::pack2(v[2], v[3]), Half<HT>::pack2(v[4], v[5]), Half<HT>::pack2(v[6], v[7]));
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <typename HT, int C>
int launch_ff(const FFParams& p, hipStream_t s) {
    constexpr size_t smem = 2 * 128 * 128 + 2 * (size_t)(128 + 256) * 128;
    hipError_t e = opt_in_lds(reinterpret_cast<const void*>(&ff_fused_kernel<HT, C>), (int)smem);
    if (e == hipSuccess) {
        ff_fused_kernel<HT, C><<<dim3(p.M / 128), 512, smem, s>>>(p);
        e = hipGetLastError();
    }
    if (e != hipSuccess) {
        gmd_set_error("gmd_ff_geglu_fused: launch failed: %s", hipGetErrorString(e));
        return GMD_ERR_LAUNCH;
    }
    return GMD_OK;
}

}  // namespace

extern "C" int gmd_ff_geglu_fused_supported(int dtype, int64_t M, int C) {
    return gmd_is_half(dtype) && C == 320 && M > 0 && M % 128 == 0 && M * (int64_t)C * 2 < 0xFFFF0000LL ? 1 : 0;
}

extern "C" int gmd_ff_geglu_fused(const void* X, const void* W1i, const float* b1i, const void* W2, const float* b2, const void* residual,
                                  void* Y, int dtype, int64_t M, int C, gmd_stream_t stream) {
    GMD_REQUIRE(gmd_ff_geglu_fused_supported(dtype, M, C), "gmd_ff_geglu_fused: needs a 16-bit dtype, C == 320 and M %% 128 == 0 (got dtype %d, M=%lld, C=%d)",
                dtype, (long long)M, C);
    GMD_REQUIRE(X && W1i && b1i && W2 && b2 && residual && Y, "gmd_ff_geglu_fused: null pointer");
    GMD_REQUIRE(gmd_aligned16(X) && gmd_aligned16(W1i) && gmd_aligned16(b1i) && gmd_aligned16(W2) && gmd_aligned16(b2) && gmd_aligned16(residual) &&
                    gmd_aligned16(Y), "gmd_ff_geglu_fused: pointers must be 16-byte aligned");
    FFParams p;
    p.X = X; p.W1 = W1i; p.b1 = b1i; p.W2 = W2; p.b2 = b2; p.R = residual; p.Y = Y; p.M = (int)M;
    p.x_bytes = (unsigned)(M * C * 2); p.w1_bytes = (unsigned)(8 * C * C * 2); p.w2_bytes = (unsigned)(C * 4 * C * 2);
    return dtype == GMD_F16 ? launch_ff<f16_t, 320>(p, (hipStream_t)stream) : launch_ff<bf16_t, 320>(p, (hipStream_t)stream);
}

GMD_WG_TRACE_SETTER(ff_fused)

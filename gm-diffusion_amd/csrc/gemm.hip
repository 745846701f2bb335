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
#include "gmd_common.h"

namespace {

struct GemmParams {
    const void* A;
    const void* W;
    void* C;
    int M, N, K;
    int64_t lda, ldw, ldc, sA, sW, sC;
    const float* bias;
    const float* rowbias;
    int rows_per_group;
    const void* residual;
    int64_t ldr, sR;
    float alpha;
    int act;
    int out_f32;
    // conv3x3 geometry (CONV instantiations only)
    int Hin, Win, Cin, Hout, Wout, stride, upsample, pad_lo;
};

// Row-invariant part of the A address of one staging slot.
struct RowCtx {
    bool valid;
    int b, oy, ox;         // conv: output pixel
    int64_t base;          // gemm: element offset of the row start
};

template <bool CONV>
__device__ __forceinline__ RowCtx make_row(const GemmParams& p, int m) {
    RowCtx r;
    r.valid = m < p.M;
    r.b = r.oy = r.ox = 0;
    r.base = 0;
    if (!r.valid) return r;
    if (CONV) {
        const int hw = p.Hout * p.Wout;
        r.b = m / hw;
        const int rem = m - r.b * hw;
        r.oy = rem / p.Wout;
        r.ox = rem - r.oy * p.Wout;
    } else {
        r.base = (int64_t)m * p.lda;
    }
    return r;
}

// element offset into A of (row, k-step starting at channel c0 of tap (ky,kx)), or -1 for padding
template <bool CONV>
__device__ __forceinline__ int64_t a_offset(const GemmParams& p, const RowCtx& r, int k0, int ky, int kx, int c0) {
    if (!r.valid) return -1;
    if (!CONV) return r.base + k0;
    int iy, ix;
    if (p.upsample) {
        const int uy = r.oy + ky - 1, ux = r.ox + kx - 1;
        if (uy < 0 || ux < 0 || uy >= 2 * p.Hin || ux >= 2 * p.Win) return -1;
        iy = uy >> 1;
        ix = ux >> 1;
    } else {
        iy = r.oy * p.stride + ky - p.pad_lo;
        ix = r.ox * p.stride + kx - p.pad_lo;
        if (iy < 0 || ix < 0 || iy >= p.Hin || ix >= p.Win) return -1;
    }
    return (((int64_t)r.b * p.Hin + iy) * p.Win + ix) * p.Cin + c0;
}

__device__ __forceinline__ float apply_act(float v, int act) { return act == GMD_ACT_SILU ? silu_f(v) : v; }

// ------------------------------------------------------------------------------------------------
// bf16 MFMA kernel
// ------------------------------------------------------------------------------------------------
constexpr int BK = 64;  // bf16 elements per K step = 128 bytes = 8 chunks of 16 bytes

// byte offset of 16-byte chunk `chunk` of row `row` in a [rows][128 B] tile; the XOR makes both the
// 8-lane ds_write_b128 groups and the 16-lane ds_read_b128 groups of a 16x16x32 fragment conflict-free
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <bool CONV, int BM, int BN>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const GemmParams p) {
    constexpr int NA = BM / 32, NW = BN / 32;  // staging slots per thread
    constexpr int TM = BM / 32, TN = BN / 32;  // 16x16 tiles per wave along M / N (wave tile = BM/2 x BN/2)
    constexpr int LDC = BN + 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int kStage = (BM + BN) * 128;  // bytes per pipeline stage: A tile then W tile

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
    const int z = blockIdx.z;
    const bf16_t* A = (const bf16_t*)p.A + (int64_t)z * p.sA;
    const bf16_t* W = (const bf16_t*)p.W + (int64_t)z * p.sW;

    const int chunk = tid & 7, srow = tid >> 3;  // staging: 8 chunks per row, 32 rows per pass
    RowCtx ra[NA];
    int64_t wbase[NW];
#pragma unroll
    for (int i = 0; i < NA; ++i) ra[i] = make_row<CONV>(p, m0 + srow + 32 * i);
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int n = n0 + srow + 32 * i;
        wbase[i] = n < p.N ? (int64_t)n * p.ldw : -1;
    }

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    uint4 rga[NA], rgw[NW];
    const int nk = p.K / BK;
    int tap = 0, c0 = 0;  // conv: running (tap, channel) of the K step being LOADED

    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
        const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int64_t off = a_offset<CONV>(p, ra[i], k0, ky, kx, c0);
            rga[i] = off >= 0 ? *reinterpret_cast<const uint4*>(A + off + chunk * 8) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NW; ++i)
            rgw[i] = wbase[i] >= 0 ? *reinterpret_cast<const uint4*>(W + wbase[i] + k0 + chunk * 8) : make_uint4(0, 0, 0, 0);
        if (CONV) {
            c0 += BK;
            if (c0 >= p.Cin) { c0 = 0; ++tap; }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<uint4*>(smem + buf * kStage + lds_off(srow + 32 * i, chunk)) = rga[i];
#pragma unroll
        for (int i = 0; i < NW; ++i) *reinterpret_cast<uint4*>(smem + buf * kStage + BM * 128 + lds_off(srow + 32 * i, chunk)) = rgw[i];
    };

    load_tile(0);
    store_tile(0);
    __syncthreads();

    const int frow = lane & 15, fq = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[i] = as_frag(*reinterpret_cast<const uint4*>(smem + cur * kStage + lds_off(wr * (BM / 2) + i * 16 + frow, 4 * s + fq)));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                b[j] = as_frag(*reinterpret_cast<const uint4*>(smem + cur * kStage + BM * 128 + lds_off(wc * (BN / 2) + j * 16 + frow, 4 * s + fq)));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: accumulators -> LDS (fp32, 64 rows per pass) -> coalesced fused store ----
    float* Cs = reinterpret_cast<float*>(smem);
    constexpr int PASSES = BM / 64;  // BM=128: wave row `h` per pass; BM=64: one pass, both wave rows
    constexpr int CH = BN / 8;
    const bool vec_ok = (p.ldc % 8 == 0) && (p.sC % 8 == 0) && (p.residual == nullptr || (p.ldr % 8 == 0 && p.sR % 8 == 0));
    for (int h = 0; h < PASSES; ++h) {
        if (PASSES == 1 || wr == h) {
            const int rbase = PASSES == 1 ? wr * (BM / 2) : 0;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        // C/D layout of mfma_f32_16x16x32: row = 4*(lane>>4)+reg, col = lane&15
                        const int ml = rbase + i * 16 + fq * 4 + r;
                        const int nl = wc * (BN / 2) + j * 16 + frow;
                        Cs[ml * LDC + nl] = acc[i][j][r];
                    }
        }
        __syncthreads();
        for (int idx = tid; idx < 64 * CH; idx += 256) {
            const int ml = idx / CH, ch = idx - ml * CH;
            const int m = m0 + h * 64 + ml, n = n0 + ch * 8;
            if (m >= p.M || n >= p.N) continue;
            float v[8];
            const float4 v0 = *reinterpret_cast<const float4*>(Cs + ml * LDC + ch * 8);
            const float4 v1 = *reinterpret_cast<const float4*>(Cs + ml * LDC + ch * 8 + 4);
            v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
            const int nvalid = p.N - n < 8 ? p.N - n : 8;
            const float* rb = p.rowbias ? p.rowbias + (int64_t)(m / p.rows_per_group) * p.N : nullptr;
            const bf16_t* res = p.residual ? (const bf16_t*)p.residual + (int64_t)z * p.sR + (int64_t)m * p.ldr + n : nullptr;
            float rv[8];
            if (res) {
                if (nvalid == 8 && vec_ok) load_vec(res, rv);
                else
                    for (int j = 0; j < 8; ++j) rv[j] = j < nvalid ? bf16_to_f32(res[j]) : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j < nvalid) {
                    float x = v[j] * p.alpha;
                    if (p.bias) x += p.bias[n + j];
                    if (rb) x += rb[n + j];
                    if (res) x += rv[j];
                    v[j] = apply_act(x, p.act);
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
                bf16_t* o = (bf16_t*)p.C + coff;
                if (nvalid == 8 && vec_ok) store_vec(o, v);
                else
                    for (int j = 0; j < nvalid; ++j) o[j] = f32_to_bf16(v[j]);
            }
        }
        if (h + 1 < PASSES) __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// float32 FMA kernel (parity path): 64x64x16 tiles, 4x4 outputs per thread
// ------------------------------------------------------------------------------------------------
template <bool CONV>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmParams p) {
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
        const float* rb = p.rowbias ? p.rowbias + (int64_t)(m / p.rows_per_group) * p.N : nullptr;
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

template <bool CONV>
int launch(const GemmParams& p, int dtype, int batch, hipStream_t s, const char* name) {
    if (dtype == GMD_BF16) {
        const int64_t tiles128 = (int64_t)((p.M + 127) / 128) * ((p.N + 127) / 128) * batch;
        if (tiles128 >= 192 && p.N > 64) {
            dim3 grid((p.N + 127) / 128, (p.M + 127) / 128, batch);
            const size_t smem = 2 * (128 + 128) * 128;  // 64 KiB main-loop buffers (epilogue staging reuses 33 KiB of it)
            gemm_bf16_kernel<CONV, 128, 128><<<grid, 256, smem, s>>>(p);
        } else {
            dim3 grid((p.N + 63) / 64, (p.M + 63) / 64, batch);
            const size_t smem = 2 * (64 + 64) * 128;  // main-loop buffers (32 KiB) > epilogue staging (17 KiB)
            gemm_bf16_kernel<CONV, 64, 64><<<grid, 256, smem, s>>>(p);
        }
    } else {
        dim3 grid((p.N + 63) / 64, (p.M + 63) / 64, batch);
        gemm_f32_kernel<CONV><<<grid, 256, 0, s>>>(p);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        gmd_set_error("%s: launch failed: %s", name, hipGetErrorString(e));
        return GMD_ERR_LAUNCH;
    }
    return GMD_OK;
}

}  // namespace

extern "C" {

int gmd_gemm_nt(const void* A, const void* W, void* C, int dtype, int out_dtype, int M, int N, int K, int64_t lda,
                int64_t ldw, int64_t ldc, int batch, int64_t strideA, int64_t strideW, int64_t strideC, const float* bias,
                const float* rowbias, int rows_per_group, const void* residual, int64_t ldr, int64_t strideR, float alpha,
                int act, gmd_stream_t stream) {
    GMD_REQUIRE(dtype == GMD_BF16 || dtype == GMD_F32, "gmd_gemm_nt: bad dtype %d", dtype);
    GMD_REQUIRE(out_dtype == dtype || out_dtype == GMD_F32, "gmd_gemm_nt: out_dtype must be F32 or the input dtype");
    GMD_REQUIRE(M >= 0 && N >= 0 && K > 0 && batch >= 0, "gmd_gemm_nt: bad shape M=%d N=%d K=%d batch=%d", M, N, K, batch);
    if (M == 0 || N == 0 || batch == 0) return GMD_OK;
    const int kmul = dtype == GMD_BF16 ? 64 : 4, vec = dtype == GMD_BF16 ? 8 : 4;
    GMD_REQUIRE(K % kmul == 0, "gmd_gemm_nt: K=%d must be a multiple of %d", K, kmul);
    GMD_REQUIRE(lda >= K && ldw >= K && ldc >= N, "gmd_gemm_nt: leading dimension too small");
    GMD_REQUIRE(lda % vec == 0 && ldw % vec == 0 && strideA % vec == 0 && strideW % vec == 0,
                "gmd_gemm_nt: lda/ldw/strides must be multiples of %d elements", vec);
    GMD_REQUIRE(A && W && C && gmd_aligned16(A) && gmd_aligned16(W) && gmd_aligned16(C), "gmd_gemm_nt: null or unaligned pointer");
    GMD_REQUIRE(rowbias == nullptr || rows_per_group > 0, "gmd_gemm_nt: rows_per_group must be positive");
    GMD_REQUIRE(residual == nullptr || (ldr >= N && gmd_aligned16(residual)), "gmd_gemm_nt: bad residual");
    GMD_REQUIRE(residual == nullptr || out_dtype == dtype || dtype == GMD_F32, "gmd_gemm_nt: residual needs out_dtype == dtype");
    GMD_REQUIRE(act == GMD_ACT_NONE || act == GMD_ACT_SILU, "gmd_gemm_nt: bad act %d", act);
    GMD_REQUIRE(batch <= 65535, "gmd_gemm_nt: batch too large");
    GemmParams p{};
    p.A = A; p.W = W; p.C = C; p.M = M; p.N = N; p.K = K;
    p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.sA = strideA; p.sW = strideW; p.sC = strideC;
    p.bias = bias; p.rowbias = rowbias; p.rows_per_group = rows_per_group > 0 ? rows_per_group : 1;
    p.residual = residual; p.ldr = ldr; p.sR = strideR; p.alpha = alpha; p.act = act;
    p.out_f32 = out_dtype == GMD_F32;
    return launch<false>(p, dtype, batch, (hipStream_t)stream, "gmd_gemm_nt");
}

int gmd_conv3x3(const void* X, const void* Wt, void* Y, int dtype, int out_dtype, int B, int Hin, int Win, int Cin, int Cout,
                int stride, int upsample, int pad_mode, const float* bias, const float* rowbias, const void* residual,
                gmd_stream_t stream) {
    GMD_REQUIRE(dtype == GMD_BF16 || dtype == GMD_F32, "gmd_conv3x3: bad dtype %d", dtype);
    GMD_REQUIRE(out_dtype == dtype || out_dtype == GMD_F32, "gmd_conv3x3: out_dtype must be F32 or the input dtype");
    GMD_REQUIRE(B > 0 && Hin > 0 && Win > 0 && Cin > 0 && Cout > 0, "gmd_conv3x3: bad shape");
    GMD_REQUIRE(stride == 1 || stride == 2, "gmd_conv3x3: stride must be 1 or 2");
    GMD_REQUIRE(!(upsample && stride != 1), "gmd_conv3x3: upsample requires stride 1");
    GMD_REQUIRE(pad_mode == 0 || (pad_mode == 1 && stride == 2 && !upsample), "gmd_conv3x3: pad_mode 1 requires stride 2");
    const int kmul = dtype == GMD_BF16 ? 64 : 16;
    GMD_REQUIRE(Cin % kmul == 0, "gmd_conv3x3: Cin=%d must be a multiple of %d (pad the channels)", Cin, kmul);
    GMD_REQUIRE(X && Wt && Y && gmd_aligned16(X) && gmd_aligned16(Wt) && gmd_aligned16(Y), "gmd_conv3x3: null or unaligned pointer");
    GMD_REQUIRE(residual == nullptr || gmd_aligned16(residual), "gmd_conv3x3: unaligned residual");
    GMD_REQUIRE(residual == nullptr || out_dtype == dtype, "gmd_conv3x3: residual needs out_dtype == dtype");
    int Hout, Wout, pad_lo;
    if (upsample) { Hout = 2 * Hin; Wout = 2 * Win; pad_lo = 1; }
    else if (pad_mode == 1) { Hout = (Hin + 1 - 3) / 2 + 1; Wout = (Win + 1 - 3) / 2 + 1; pad_lo = 0; }
    else { Hout = (Hin + 2 - 3) / stride + 1; Wout = (Win + 2 - 3) / stride + 1; pad_lo = 1; }
    const int64_t M = (int64_t)B * Hout * Wout;
    GMD_REQUIRE(M < (1LL << 31), "gmd_conv3x3: too many output pixels");
    GemmParams p{};
    p.A = X; p.W = Wt; p.C = Y; p.M = (int)M; p.N = Cout; p.K = 9 * Cin;
    p.lda = Cin; p.ldw = 9 * (int64_t)Cin; p.ldc = Cout;
    p.bias = bias; p.rowbias = rowbias; p.rows_per_group = Hout * Wout;
    p.residual = residual; p.ldr = Cout; p.alpha = 1.0f; p.act = GMD_ACT_NONE;
    p.out_f32 = out_dtype == GMD_F32;
    p.Hin = Hin; p.Win = Win; p.Cin = Cin; p.Hout = Hout; p.Wout = Wout; p.stride = stride; p.upsample = upsample; p.pad_lo = pad_lo;
    return launch<true>(p, dtype, 1, (hipStream_t)stream, "gmd_conv3x3");
}

}  // extern "C"

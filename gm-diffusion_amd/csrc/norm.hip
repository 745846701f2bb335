// GroupNorm (statistics + fused affine/SiLU apply), LayerNorm and row softmax over
// channels-last activations.  HBM-bound: 16-byte accesses, wave-shuffle reductions, and a
// fixed (deterministic) reduction order everywhere -- no float atomics.
#include "gmd_common.h"
#include <math.h>
#include <type_traits>
#include <stdlib.h>

namespace {

constexpr int kThreads = 256;

// ---------------------------------------------------------------------------------------------
// GroupNorm statistics.  grid (nsplit, B); each block reduces a contiguous pixel range.
// Thread t owns channel chunk (t % CVB) (+ column-block offset) of pixel row (t / CVB):
// per-channel partials go to LDS [PY][C], then thread g < G folds its group's channels in a
// fixed order and writes {sum, sumsq} to workspace[b][split][g].
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kThreads) void gn_partial_kernel(const T* __restrict__ X, int64_t HW, int C, int G,
                                                              int nsplit, float* __restrict__ ws) {
    GMD_WG_TRACE_SCOPE(WGK_GN_PARTIAL);
    constexpr int V = Elem<T>::kVec;
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [PY][C] sums, [PY][C] sumsq
    const int b = blockIdx.y, split = blockIdx.x;
    const int CV = C / V;
    const int CVB = CV < kThreads ? CV : kThreads;
    const int PY = kThreads / CVB;
    const int py = threadIdx.x / CVB, cx = threadIdx.x % CVB;
    const int64_t per = (HW + nsplit - 1) / nsplit;
    const int64_t p0 = (int64_t)split * per;
    const int64_t p1 = (p0 + per < HW) ? p0 + per : HW;
    float* s_sum = smem;
    float* s_sq = smem + (size_t)PY * C;
    const T* Xb = X + (int64_t)b * HW * C;
    for (int cb = 0; cb < CV; cb += CVB) {
        const int chunk = cb + cx;
        float a[V], q[V];
#pragma unroll
        for (int j = 0; j < V; ++j) a[j] = q[j] = 0.0f;
        if (py < PY && chunk < CV) {
            int64_t p = p0 + py;
            for (; p + PY < p1; p += 2 * PY) {  // two independent 16-byte loads in flight per thread
                float v[V], w[V];
                load_vec(Xb + p * C + (int64_t)chunk * V, v);
                load_vec(Xb + (p + PY) * C + (int64_t)chunk * V, w);
#pragma unroll
                for (int j = 0; j < V; ++j) { a[j] += v[j]; q[j] += v[j] * v[j]; }
#pragma unroll
                for (int j = 0; j < V; ++j) { a[j] += w[j]; q[j] += w[j] * w[j]; }
            }
            if (p < p1) {
                float v[V];
                load_vec(Xb + p * C + (int64_t)chunk * V, v);
#pragma unroll
                for (int j = 0; j < V; ++j) { a[j] += v[j]; q[j] += v[j] * v[j]; }
            }
#pragma unroll
            for (int j = 0; j < V; ++j) {
                s_sum[(size_t)py * C + chunk * V + j] = a[j];
                s_sq[(size_t)py * C + chunk * V + j] = q[j];
            }
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < G) {
        const int g = threadIdx.x, cpg = C / G;
        double s = 0.0, s2 = 0.0;
        for (int r = 0; r < PY; ++r)
            for (int c = g * cpg; c < (g + 1) * cpg; ++c) { s += s_sum[(size_t)r * C + c]; s2 += s_sq[(size_t)r * C + c]; }
        float* o = ws + (((int64_t)b * nsplit + split) * G + g) * 2;
        o[0] = (float)s;
        o[1] = (float)s2;
    }
}

// grid B: fold the splits (8 lanes per group, each a fixed strided subset, then a fixed-order shuffle tree:
// deterministic), then per-channel affine
__global__ __launch_bounds__(kThreads) void gn_finalize_kernel(const float* __restrict__ ws, int nsplit, int64_t HW, int C,
                                                               int G, float eps, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* __restrict__ ss) {
    GMD_WG_TRACE_SCOPE(WGK_GN_FINALIZE);
    __shared__ float s_mean[64], s_rstd[64];
    const int b = blockIdx.x, cpg = C / G;
    for (int g0 = 0; g0 < G; g0 += kThreads / 8) {
        const int g = g0 + (int)threadIdx.x / 8, sub = threadIdx.x & 7;
        double s = 0.0, s2 = 0.0;
        if (g < G) {
#pragma unroll 8
            for (int k = sub; k < nsplit; k += 8) {
                const float2 o = *reinterpret_cast<const float2*>(ws + (((int64_t)b * nsplit + k) * G + g) * 2);
                s += o.x; s2 += o.y;
            }
        }
        s = group8_sum(s);
        s2 = group8_sum(s2);
        if (g < G && sub == 0) {
            const double n = (double)HW * cpg;
            const double mean = s / n;
            double var = s2 / n - mean * mean;
            if (var < 0.0) var = 0.0;
            s_mean[g] = (float)mean;
            s_rstd[g] = (float)(1.0 / sqrt(var + (double)eps));
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const int g = c / cpg;
        const float sc = s_rstd[g] * gamma[c];
        ss[((int64_t)b * C + c) * 2] = sc;
        ss[((int64_t)b * C + c) * 2 + 1] = beta[c] - s_mean[g] * sc;
    }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void gn_apply_kernel(const T* __restrict__ X, T* __restrict__ Y, int B, int64_t HW, int C,
                                                            const float* __restrict__ ss, int silu) {
    GMD_WG_TRACE_SCOPE(WGK_GN_APPLY);
    constexpr int V = Elem<T>::kVec;
    const int CV = C / V;
    const int64_t total = (int64_t)B * HW * CV;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int chunk = (int)(i % CV);
        const int64_t b = i / ((int64_t)HW * CV);
        float v[V];
        load_vec(X + i * V, v);
        const float* p = ss + ((int64_t)b * C + (int64_t)chunk * V) * 2;
#pragma unroll
        for (int j = 0; j < V; j += 2) {
            const float4 q = *reinterpret_cast<const float4*>(p + 2 * j);
            v[j] = v[j] * q.x + q.y;
            v[j + 1] = v[j + 1] * q.z + q.w;
        }
        if (silu) {
#pragma unroll
            for (int j = 0; j < V; ++j) v[j] = silu_f(v[j]);
        }
        store_vec(Y + i * V, v);
    }
}

// Two-launch split path: apply with the statistics folded in.  grid (nb, B): every workgroup first folds the nsplit
// partial sums of ITS sample (8 lanes per group over a fixed strided subset, then a fixed-order shuffle tree -- the same
// deterministic order as gn_finalize_kernel), builds the per-channel scale / shift in LDS, then normalises its slice of the
// sample's rows.  The redundant fold costs ~1 us per workgroup and saves the finalize launch (6.5 us of pure latency).
// GMD_F32SA as the dtype of an apply kernel: float32 in, the OUTPUT stored pre-split for the next contraction (gmd_common.h:
// gmd_store_split4; `e` = element index of v[0] over the whole tensor, rows are multiples of 32 elements)
template <bool SPLIT_OUT, typename T, int V>
__device__ __forceinline__ void store_vec_maybe_split(T* Y, int64_t e, const float (&v)[V]) {
    if constexpr (SPLIT_OUT) {
        static_assert(sizeof(T) == 4 && V == 4, "pre-split output is a float32 format");
        gmd_store_split4(reinterpret_cast<float*>(Y), e, v[0], v[1], v[2], v[3]);
    } else {
        store_vec(Y + e, v);
    }
}


// The apply pass of gn_apply_ws_kernel / gn_apply_cs_kernel.  A workgroup owns `span` = 256 * U consecutive 16-byte vectors of one
// sample (U in {2, 4, 6, 10}, chosen by the host: gn_apply_span_u), thread t the vectors first + t + 256 u; channel chunk = index mod CV, scale /
// shift per channel in LDS.  Round 5: a thread's loads are requested in two halves around the statistics fold (X does not
// depend on it, see gn_rows_apply) and there is no loop -- the old form kept ONE 16-byte load per thread outstanding behind a 64-bit modulo (2048
// waves x 1 KB = 2 MB in flight on the whole chip: 1.9-3.4 TB/s alone, less beside a second stream).  Loads are unconditional with a
// clamped index and stores go through a range-checked buffer descriptor, so the pass is ONE basic block (see gn_rows_request).
typedef unsigned gn_u32x4 __attribute__((ext_vector_type(4)));
template <typename T>
__device__ __forceinline__ void gn_unpack16(const uint4& r, float (&v)[Elem<T>::kVec]) {
    const unsigned w[4] = {r.x, r.y, r.z, r.w};
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = __uint_as_float(w[k]);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) Half<T>::unpack2(w[k], v[2 * k], v[2 * k + 1]);
    }
}
// loads u in [U0, U1) of a thread's vectors.  Unconditional, the index clamped to the workgroup's last vector: a load inside an
// `if (i < vend)` is waited for right there, and even a wave-uniform skip splits the code into blocks at whose edges the
// compiler's wait-count pass falls back to vmcnt(0) (only the last workgroup of a sample can have a thread past the end)
template <typename T, int U0, int U1, int U>
__device__ __forceinline__ void gn_rows_request(const T* Xb, int vbeg, int vend, uint4 (&raw)[U]) {
    constexpr int V = Elem<T>::kVec;
    const int vlast = vend > 0 ? vend - 1 : 0;
#pragma unroll
    for (int u = U0; u < U1; ++u) {
        const int i = vbeg + u * kThreads;
        raw[u] = *reinterpret_cast<const uint4*>(Xb + (int64_t)(i < vlast ? i : vlast) * V);
    }
}
// The first half of a thread's vectors is requested before the statistics are folded, the second half here, right behind the
// prologue's last barrier and in front of the first half's arithmetic and stores: the workgroups of a launch run in lock-step (all
// resident at once), and with every load up front the launch was a read phase followed by a write phase (10.3 us for 42 MB without
// SiLU where LayerNorm's small workgroups take 6.7); now the second half's reads overlap the first half's writes.  X and Y are NOT
// declared restrict here: the compiler must keep the second half's loads above the first half's stores.
template <bool SPLIT_OUT, typename T, int U>
__device__ __forceinline__ void gn_rows_apply(const T* Xb, T* Yb, const float* ss, int CV, int vbeg, int vend, int silu, uint4 (&raw)[U]) {
    constexpr int V = Elem<T>::kVec;
    int c = vbeg % CV;
    const int step = kThreads % CV;
    gn_rows_request<T, U / 2, U, U>(Xb, vbeg, vend, raw);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(Yb, 0, vend * 16, 0x00020000);  // (vend < 2^27, so byte offsets stay positive ints: host-checked)
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = vbeg + u * kThreads;
        float xv[V];
        gn_unpack16<T>(raw[u], xv);
        const float* q = ss + (size_t)c * V * 2;
#pragma unroll
        for (int j = 0; j < V; j += 2) {
            const float4 t = *reinterpret_cast<const float4*>(q + 2 * j);
            xv[j] = xv[j] * t.x + t.y;
            xv[j + 1] = xv[j + 1] * t.z + t.w;
        }
        if (silu) {
#pragma unroll
            for (int j = 0; j < V; ++j) xv[j] = silu_f(xv[j]);
        }
        if constexpr (SPLIT_OUT) {
            if (i < vend) store_vec_maybe_split<SPLIT_OUT>(Yb, (int64_t)i * V, xv);
        } else {
            // range-checked buffer store (vectors >= vend are dropped by the hardware): no branch around the store, so the whole pass
            // is one basic block and the compiler counts the waits (behind per-lane `if (i < vend)` blocks it waited for ALL earlier
            // stores before the last vectors' arithmetic)
            gn_u32x4 o;
            if constexpr (sizeof(T) == 4) {
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = __float_as_uint(xv[k]);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = Half<T>::pack2(xv[2 * k], xv[2 * k + 1]);
            }
            __builtin_amdgcn_raw_buffer_store_b128(o, yrsrc, i * 16, 0, 0);
        }
        c += step;
        if (c >= CV) c -= CV;
    }
}
// vectors per thread of an apply launch over B samples of `nvec` 16-byte vectors: the largest instantiated count (2, 4, 6, 10) that
// still leaves >= 512 workgroups
static inline int gn_apply_span_u(int B, int64_t nvec) {
    static const int forced = [] { const char* e = getenv("GMD_GN_U"); return e ? atoi(e) : 0; }();  // (experiments)
    int64_t u = forced > 0 ? forced : (nvec * B) / (512 * (int64_t)kThreads);
    return u >= 10 ? 10 : u >= 6 ? 6 : u >= 4 ? 4 : 2;
}
// f(IntC<U>) for the instantiated U
template <typename F>
static inline void gn_for_u(int u, F&& f) {
    if (u == 10) f(std::integral_constant<int, 10>{});
    else if (u == 6) f(std::integral_constant<int, 6>{});
    else if (u == 4) f(std::integral_constant<int, 4>{});
    else f(std::integral_constant<int, 2>{});
}

template <typename T, bool SPLIT_OUT, int U>
__global__ __launch_bounds__(kThreads) void gn_apply_ws_kernel(const T* __restrict__ X, T* __restrict__ Y, int64_t HW, int C, int G,
                                                               int nsplit, float eps, const float* __restrict__ ws,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               int silu, int span) {
    GMD_WG_TRACE_SCOPE(WGK_GN_APPLY_WS);
    constexpr int V = Elem<T>::kVec;
    extern __shared__ __attribute__((aligned(16))) float ss[];  // [C][2] = {scale, shift}
    __shared__ float s_mean[64], s_rstd[64];
    const int b = blockIdx.y, cpg = C / G;
    const int CV = C / V;
    const T* Xb = X + ((int64_t)b * HW) * C;
    T* Yb = Y + ((int64_t)b * HW) * C;
    const int vtot = (int)HW * CV;  // (< 2^27: host-checked)
    const int vbeg = (int)blockIdx.x * span + (int)threadIdx.x;
    const int vend = (int)blockIdx.x * span + span < vtot ? (int)blockIdx.x * span + span : vtot;
    uint4 xv[U];
    gn_rows_request<T, 0, U / 2, U>(Xb, vbeg, vend, xv);
    for (int g0 = 0; g0 < G; g0 += kThreads / 8) {
        const int g = g0 + (int)threadIdx.x / 8, sub = threadIdx.x & 7;
        double s = 0.0, s2 = 0.0;
        if (g < G) {
            for (int k = sub; k < nsplit; k += 8) {
                const float2 o = *reinterpret_cast<const float2*>(ws + (((int64_t)b * nsplit + k) * G + g) * 2);
                s += o.x; s2 += o.y;
            }
        }
        s = group8_sum(s);
        s2 = group8_sum(s2);
        if (g < G && sub == 0) {
            const double n = (double)HW * cpg;
            const double mean = s / n;
            double var = s2 / n - mean * mean;
            if (var < 0.0) var = 0.0;
            s_mean[g] = (float)mean;
            s_rstd[g] = (float)(1.0 / sqrt(var + (double)eps));
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += kThreads) {
        const int g = c / cpg;
        const float sc = s_rstd[g] * gamma[c];
        ss[2 * c] = sc;
        ss[2 * c + 1] = beta[c] - s_mean[g] * sc;
    }
    __syncthreads();
    gn_rows_apply<SPLIT_OUT, T, U>(Xb, Yb, ss, CV, vbeg, vend, silu, xv);
}

// Apply with the statistics taken from the PRODUCER of X: the GEMM / convolution that wrote X left {sum, sum of squares} per
// 64-row block and per bucket of `bucket` adjacent channels (gmd_gemm_nt / gmd_conv3x3 `colstats`), so the statistics pass
// over X -- a third of this operator's traffic and one of its two launches -- disappears.  X may be the channel concatenation
// of two producers' outputs (the up blocks' skip connections): channels < Ca take their sums from sa [rows/64][Ca/bucket][2],
// the others from sb [rows/64][(C-Ca)/bucket][2].  Same fold as gn_apply_ws_kernel (8 lanes per group over a fixed strided
// subset of its (row block, bucket) items, double accumulation, fixed-order shuffle tree): deterministic.
template <typename T, bool SPLIT_OUT, int U>
__global__ __launch_bounds__(kThreads) void gn_apply_cs_kernel(const T* __restrict__ X, T* __restrict__ Y, int64_t HW, int C, int G,
                                                               float eps, const float* __restrict__ sa, int Ca,
                                                               const float* __restrict__ sb, int bucket,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               int silu, int span) {
    GMD_WG_TRACE_SCOPE(WGK_GN_APPLY_CS);
    constexpr int V = Elem<T>::kVec;
    extern __shared__ __attribute__((aligned(16))) float ss[];  // [C][2] = {scale, shift}
    __shared__ float s_mean[64], s_rstd[64];
    const int b = blockIdx.y, cpg = C / G;
    const int CV = C / V;
    const T* Xb = X + ((int64_t)b * HW) * C;
    T* Yb = Y + ((int64_t)b * HW) * C;
    const int vtot = (int)HW * CV;  // (< 2^27: host-checked)
    const int vbeg = (int)blockIdx.x * span + (int)threadIdx.x;
    const int vend = (int)blockIdx.x * span + span < vtot ? (int)blockIdx.x * span + span : vtot;
    uint4 xv[U];
    gn_rows_request<T, 0, U / 2, U>(Xb, vbeg, vend, xv);
    const int rb = (int)(HW / 64), bpg = cpg / bucket, nba = Ca / bucket, nbb = (C - Ca) / bucket;
    const int items = rb * bpg;
    for (int g0 = 0; g0 < G; g0 += kThreads / 8) {
        const int g = g0 + (int)threadIdx.x / 8, sub = threadIdx.x & 7;
        double s = 0.0, s2 = 0.0;
        if (g < G) {
#pragma unroll 8  // the partial sums are independent loads: issue them together (same order of additions)
            for (int k = sub; k < items; k += 8) {
                const int r = k / bpg, kb = g * bpg + (k - r * bpg);
                const int64_t row = (int64_t)b * rb + r;
                const float* src = kb < nba ? sa + (row * nba + kb) * 2 : sb + (row * nbb + (kb - nba)) * 2;
                const float2 o = *reinterpret_cast<const float2*>(src);
                s += o.x; s2 += o.y;
            }
        }
        s = group8_sum(s);
        s2 = group8_sum(s2);
        if (g < G && sub == 0) {
            const double n = (double)HW * cpg;
            const double mean = s / n;
            double var = s2 / n - mean * mean;
            if (var < 0.0) var = 0.0;
            s_mean[g] = (float)mean;
            s_rstd[g] = (float)(1.0 / sqrt(var + (double)eps));
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += kThreads) {
        const int g = c / cpg;
        const float sc = s_rstd[g] * gamma[c];
        ss[2 * c] = sc;
        ss[2 * c + 1] = beta[c] - s_mean[g] * sc;
    }
    __syncthreads();
    gn_rows_apply<SPLIT_OUT, T, U>(Xb, Yb, ss, CV, vbeg, vend, silu, xv);
}

// ---------------------------------------------------------------------------------------------
// Fused GroupNorm(+SiLU), one workgroup per (sample, group) over the group's slab [HW][C/G]: exact two-pass mean /
// variance (fixed reduction order: deterministic) and the normalised result written straight back -- ONE launch
// instead of partial + finalize + apply, for slabs small enough to be re-read from L2.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();  // protect `red` from the previous use
    if (lane == 0) red[wid] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// The slab (<= 128 KiB, host-checked) is read three times -- sum, centred sum of squares, normalise -- the second
// and third time out of L2; no register array, so the kernel stays small and many workgroups share a CU.
template <typename T, int VB>  // VB = bytes per access (4, 8 or 16): the widest that divides the group's row of C/G channels
__global__ __launch_bounds__(kThreads) void gn_fused_kernel(const T* __restrict__ X, T* __restrict__ Y, int HW, int C, int G,
                                                            float eps, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int silu) {
    GMD_WG_TRACE_SCOPE(WGK_GN_FUSED);
    constexpr int EP = VB / (int)sizeof(T);       // elements per access
    constexpr int NW = VB / 4;                    // 32-bit words per access
    typedef unsigned vec_t __attribute__((ext_vector_type(NW)));
    __shared__ float red[4];
    const int g = blockIdx.x, b = blockIdx.y;
    const int cpg = C / G, vpr = cpg / EP;        // accesses per pixel row of this group
    const int total = HW * vpr;
    const T* Xg = X + ((int64_t)b * HW) * C + (int64_t)g * cpg;
    T* Yg = Y + ((int64_t)b * HW) * C + (int64_t)g * cpg;
    // item it = tid + 256 k -> (pixel, access-in-row), advanced incrementally (no division in the loops)
    const int dq = kThreads / vpr, dr = kThreads - dq * vpr;
    const int px0 = (int)threadIdx.x / vpr, j0 = (int)threadIdx.x - px0 * vpr;
    auto unpack = [](const vec_t& w, float (&v)[EP]) {
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            if constexpr (sizeof(T) == 2) Half<T>::unpack2(w[k], v[(2 * k) % EP], v[(2 * k + 1) % EP]);
            else v[k] = __uint_as_float(w[k]);
        }
    };
    float s = 0.f;
    for (int it = threadIdx.x, px = px0, j = j0; it < total; it += kThreads) {
        float v[EP];
        unpack(*reinterpret_cast<const vec_t*>(Xg + (int64_t)px * C + j * EP), v);
#pragma unroll
        for (int k = 0; k < EP; ++k) s += v[k];
        px += dq; j += dr;
        if (j >= vpr) { j -= vpr; ++px; }
    }
    const float n = (float)HW * (float)cpg;
    const float mean = block_sum_256(s, red) / n;
    float q = 0.f;
    for (int it = threadIdx.x, px = px0, j = j0; it < total; it += kThreads) {
        float v[EP];
        unpack(*reinterpret_cast<const vec_t*>(Xg + (int64_t)px * C + j * EP), v);
#pragma unroll
        for (int k = 0; k < EP; ++k) { const float a = v[k] - mean; q += a * a; }
        px += dq; j += dr;
        if (j >= vpr) { j -= vpr; ++px; }
    }
    const float rstd = rsqrtf(block_sum_256(q, red) / n + eps);
    for (int it = threadIdx.x, px = px0, j = j0; it < total; it += kThreads) {
        float v[EP];
        unpack(*reinterpret_cast<const vec_t*>(Xg + (int64_t)px * C + j * EP), v);
        const int c0 = g * cpg + j * EP;
#pragma unroll
        for (int k = 0; k < EP; ++k) {
            float a = (v[k] - mean) * rstd * gamma[c0 + k] + beta[c0 + k];
            if (silu) a = silu_f(a);
            v[k] = a;
        }
        vec_t o;
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            if constexpr (sizeof(T) == 2) o[k] = Half<T>::pack2(v[(2 * k) % EP], v[(2 * k + 1) % EP]);
            else o[k] = __float_as_uint(v[k % EP]);
        }
        *reinterpret_cast<vec_t*>(Yg + (int64_t)px * C + j * EP) = o;
        px += dq; j += dr;
        if (j >= vpr) { j -= vpr; ++px; }
    }
}

// Register-resident form of gn_fused_kernel for slabs of at most 256 * MAXIT accesses (the SD-1.5 GroupNorms of the 16x16
// level): ONE round trip to memory -- all of a thread's accesses are issued together and stay in registers through the two
// reductions and the normalisation -- instead of three dependent passes of one load at a time (round 2: 9-10 us per launch for
// 20-40 KB, pure latency).  Same items per thread in the same order, same reductions: bit-identical to gn_fused_kernel.
// gn_reg_finish is the part after the loads: statistics, normalisation, store (shared with gn_slab_kernel below).
template <typename T, int VB, int MAXIT>
__device__ __forceinline__ void gn_reg_finish(const unsigned (&raw)[MAXIT][VB / 4], T* __restrict__ Yg, int HW, int C, int g, int cpg,
                                              float eps, const float* __restrict__ gamma, const float* __restrict__ beta, int silu,
                                              float* red) {
    constexpr int EP = VB / (int)sizeof(T);
    constexpr int NW = VB / 4;
    typedef unsigned vec_t __attribute__((ext_vector_type(NW)));
    const int vpr = cpg / EP;
    const int total = HW * vpr;
    const int dq = kThreads / vpr, dr = kThreads - dq * vpr;
    const int px0 = (int)threadIdx.x / vpr, j0 = (int)threadIdx.x - px0 * vpr;
    auto unpack = [](const unsigned (&w)[NW], float (&v)[EP]) {
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            if constexpr (sizeof(T) == 2) Half<T>::unpack2(w[k], v[(2 * k) % EP], v[(2 * k + 1) % EP]);
            else v[k] = __uint_as_float(w[k]);
        }
    };
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < MAXIT; ++k) {
        if ((int)threadIdx.x + k * kThreads < total) {
            float v[EP];
            unpack(raw[k], v);
#pragma unroll
            for (int e = 0; e < EP; ++e) s += v[e];
        }
    }
    const float n = (float)HW * (float)cpg;
    const float mean = block_sum_256(s, red) / n;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < MAXIT; ++k) {
        if ((int)threadIdx.x + k * kThreads < total) {
            float v[EP];
            unpack(raw[k], v);
#pragma unroll
            for (int e = 0; e < EP; ++e) { const float a = v[e] - mean; q += a * a; }
        }
    }
    const float rstd = rsqrtf(block_sum_256(q, red) / n + eps);
    int px = px0, j = j0;
#pragma unroll
    for (int k = 0; k < MAXIT; ++k) {
        if ((int)threadIdx.x + k * kThreads < total) {
            float v[EP];
            unpack(raw[k], v);
            const int c0 = g * cpg + j * EP;
#pragma unroll
            for (int e = 0; e < EP; ++e) {
                float a = (v[e] - mean) * rstd * gamma[c0 + e] + beta[c0 + e];
                if (silu) a = silu_f(a);
                v[e] = a;
            }
            vec_t o;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                if constexpr (sizeof(T) == 2) o[w] = Half<T>::pack2(v[(2 * w) % EP], v[(2 * w + 1) % EP]);
                else o[w] = __float_as_uint(v[w % EP]);
            }
            *reinterpret_cast<vec_t*>(Yg + (int64_t)px * C + j * EP) = o;
        }
        px += dq; j += dr;
        if (j >= vpr) { j -= vpr; ++px; }
    }
}

template <typename T, int VB, int MAXIT>
__global__ __launch_bounds__(kThreads) void gn_fused_reg_kernel(const T* __restrict__ X, T* __restrict__ Y, int HW, int C, int G,
                                                                float eps, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, int silu) {
    GMD_WG_TRACE_SCOPE(WGK_GN_FUSED_REG);
    constexpr int EP = VB / (int)sizeof(T);
    constexpr int NW = VB / 4;
    typedef unsigned vec_t __attribute__((ext_vector_type(NW)));
    __shared__ float red[4];
    const int g = blockIdx.x, b = blockIdx.y;
    const int cpg = C / G, vpr = cpg / EP;
    const int total = HW * vpr;
    const T* Xg = X + ((int64_t)b * HW) * C + (int64_t)g * cpg;
    T* Yg = Y + ((int64_t)b * HW) * C + (int64_t)g * cpg;
    const int dq = kThreads / vpr, dr = kThreads - dq * vpr;
    const int px0 = (int)threadIdx.x / vpr, j0 = (int)threadIdx.x - px0 * vpr;
    unsigned raw[MAXIT][NW];
    {
        int px = px0, j = j0;
#pragma unroll
        for (int k = 0; k < MAXIT; ++k) {
            if ((int)threadIdx.x + k * kThreads < total) {
                const vec_t w = *reinterpret_cast<const vec_t*>(Xg + (int64_t)px * C + j * EP);
#pragma unroll
                for (int e = 0; e < NW; ++e) raw[k][e] = w[e];
            }
            px += dq; j += dr;
            if (j >= vpr) { j -= vpr; ++px; }
        }
    }
    gn_reg_finish<T, VB, MAXIT>(raw, Yg, HW, C, g, cpg, eps, gamma, beta, silu, red);
}

// GroupNorm (+SiLU) straight from the float32 partial slabs of a split-K convolution / GEMM (gmd_conv3x3_groupnorm): the
// workgroup of one (sample, group) sums its columns of the `ksplit` slabs in slab order, applies the producer's epilogue (alpha,
// bias, row bias, residual -- the association of splitk_reduce_kernel for the 16-bit types and of splitk_reduce_f32_kernel for
// float32), rounds to the activation type exactly as the stored tensor would be rounded, optionally stores that tensor, and
// normalises from registers.  Replaces splitk_reduce + GroupNorm: the reduced tensor is neither written nor re-read (unless the
// caller wants it), one launch less.  Items, order and reductions are those of gn_fused_kernel on the reduced tensor, so the
// result is bit-identical to the two-launch path.  16-byte accesses of the OUTPUT type only (cpg * sizeof(T) % 16 == 0).
struct SlabSource {
    const float* ws;       // [ksplit][M][C] partial sums
    int ksplit;
    int64_t slab;          // M * C
    float alpha;
    const float* bias;     // [C] or null
    const float* rowbias;  // [B][ldrb] or null
    int64_t ldrb;
    const void* residual;  // [M][C] of T or null
};

template <typename T, int MAXIT>
__global__ __launch_bounds__(kThreads) void gn_slab_kernel(const SlabSource sp, T* __restrict__ Yraw, T* __restrict__ Y, int HW, int C,
                                                           int G, float eps, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, int silu) {
    GMD_WG_TRACE_SCOPE(WGK_GN_SLAB);
    constexpr int EP = 16 / (int)sizeof(T);  // elements per item: 8 (16-bit) or 4 (float32)
    __shared__ float red[4];
    const int g = blockIdx.x, b = blockIdx.y;
    const int cpg = C / G, vpr = cpg / EP;
    const int total = HW * vpr;
    const int64_t base = ((int64_t)b * HW) * C + (int64_t)g * cpg;
    const int dq = kThreads / vpr, dr = kThreads - dq * vpr;
    const int px0 = (int)threadIdx.x / vpr, j0 = (int)threadIdx.x - px0 * vpr;
    int off[MAXIT], ch[MAXIT];  // element offset inside the (sample, group) slice / first channel of the item
    {
        int px = px0, j = j0;
#pragma unroll
        for (int k = 0; k < MAXIT; ++k) {
            off[k] = px * C + j * EP;
            ch[k] = g * cpg + j * EP;
            px += dq; j += dr;
            if (j >= vpr) { j -= vpr; ++px; }
        }
    }
    float acc[MAXIT][EP];
#pragma unroll
    for (int k = 0; k < MAXIT; ++k)
#pragma unroll
        for (int e = 0; e < EP; ++e) acc[k][e] = 0.f;
    // U slabs per round: all of a thread's loads of U slabs are in flight together (one workgroup per CU at most -- the grid is
    // G x B -- so registers are free and the chain of dependent round trips is what costs), added in slab order
    constexpr int U = MAXIT <= 2 ? 8 : MAXIT <= 6 ? 4 : 2;
    const float* src = sp.ws + base;
    for (int s = 0; s < sp.ksplit; s += U, src += U * sp.slab) {
        float4 t[U][MAXIT][EP / 4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (s + u < sp.ksplit) {
#pragma unroll
                for (int k = 0; k < MAXIT; ++k)
                    if ((int)threadIdx.x + k * kThreads < total) {
#pragma unroll
                        for (int h = 0; h < EP / 4; ++h) t[u][k][h] = *reinterpret_cast<const float4*>(src + u * sp.slab + off[k] + 4 * h);
                    }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (s + u < sp.ksplit) {
#pragma unroll
                for (int k = 0; k < MAXIT; ++k)
                    if ((int)threadIdx.x + k * kThreads < total) {
#pragma unroll
                        for (int h = 0; h < EP / 4; ++h) {
                            acc[k][4 * h] += t[u][k][h].x; acc[k][4 * h + 1] += t[u][k][h].y;
                            acc[k][4 * h + 2] += t[u][k][h].z; acc[k][4 * h + 3] += t[u][k][h].w;
                        }
                    }
            }
        }
    }
    const float* rb = sp.rowbias ? sp.rowbias + (int64_t)b * sp.ldrb : nullptr;
    unsigned raw[MAXIT][4];
#pragma unroll
    for (int k = 0; k < MAXIT; ++k) {
        if ((int)threadIdx.x + k * kThreads < total) {
            float rv[EP];
            if (sp.residual) {
                typedef unsigned vec_t __attribute__((ext_vector_type(4)));
                const vec_t w = *reinterpret_cast<const vec_t*>((const T*)sp.residual + base + off[k]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if constexpr (sizeof(T) == 2) Half<T>::unpack2(w[e], rv[2 * e], rv[2 * e + 1]);
                    else rv[e] = __uint_as_float(w[e]);
                }
            }
#pragma unroll
            for (int e = 0; e < EP; ++e) {
                float x = acc[k][e] * sp.alpha;
                if (sp.bias) x += sp.bias[ch[k] + e];
                {  // (acc*alpha + bias) + (residual + rowbias): the association of EVERY epilogue of gemm.hip / gemm_split.hip (round 5:
                   // the 16-bit reduction kernel used to add row bias and residual one after the other -- the two never meet in the UNet)
                    float add = 0.f;
                    if (sp.residual) add += rv[e];
                    if (rb) add += rb[ch[k] + e];
                    x += add;
                }
                acc[k][e] = x;
            }
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                if constexpr (sizeof(T) == 2) raw[k][w] = Half<T>::pack2(acc[k][2 * w], acc[k][2 * w + 1]);
                else raw[k][w] = __float_as_uint(acc[k][w]);
            }
            if (Yraw) {
                typedef unsigned vec_t __attribute__((ext_vector_type(4)));
                vec_t o;
#pragma unroll
                for (int w = 0; w < 4; ++w) o[w] = raw[k][w];
                *reinterpret_cast<vec_t*>(Yraw + base + off[k]) = o;
            }
        }
    }
    gn_reg_finish<T, 16, MAXIT>(raw, Y + base, HW, C, g, cpg, eps, gamma, beta, silu, red);
}

// ---------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, row held in registers, exact two-pass variance.
// ---------------------------------------------------------------------------------------------
template <typename T, int MAXCH, int ROWS, bool SPLIT_OUT = false>
__global__ __launch_bounds__(kThreads) void layernorm_kernel(const T* __restrict__ X, T* __restrict__ Y, int64_t rows, int C,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float eps) {
    GMD_WG_TRACE_SCOPE(WGK_LN);
    // one wave per ROWS consecutive rows, each row held in registers (exact two-pass variance); the loads of all ROWS rows
    // are issued before the first reduction, so a wave keeps ROWS x MAXCH 16-byte loads in flight instead of MAXCH
    constexpr int V = Elem<T>::kVec;
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6)) * ROWS;
    if (row0 >= rows) return;
    const int CV = C / V;
    float v[ROWS][MAXCH][V];
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr)
#pragma unroll
        for (int k = 0; k < MAXCH; ++k) {
            const int chunk = lane + 64 * k;
            if (chunk < CV && row0 + rr < rows) {
                load_vec(X + (row0 + rr) * C + (int64_t)chunk * V, v[rr][k]);
            } else {
#pragma unroll
                for (int j = 0; j < V; ++j) v[rr][k][j] = 0.0f;
            }
        }
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr) {
        if (row0 + rr >= rows) break;  // wave-uniform
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < MAXCH; ++k)
#pragma unroll
            for (int j = 0; j < V; ++j) s += v[rr][k][j];
        const float mean = wave_sum(s) / (float)C;
        float s2 = 0.0f;
#pragma unroll
        for (int k = 0; k < MAXCH; ++k) {
            if (lane + 64 * k < CV) {
#pragma unroll
                for (int j = 0; j < V; ++j) { const float d = v[rr][k][j] - mean; s2 += d * d; }
            }
        }
        const float rstd = rsqrtf(wave_sum(s2) / (float)C + eps);
#pragma unroll
        for (int k = 0; k < MAXCH; ++k) {
            const int chunk = lane + 64 * k;
            if (chunk < CV) {
                float o[V];
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const int c = chunk * V + j;
                    o[j] = (v[rr][k][j] - mean) * rstd * gamma[c] + beta[c];
                }
                store_vec_maybe_split<SPLIT_OUT>(Y, (row0 + rr) * C + (int64_t)chunk * V, o);
            }
        }
    }
}

// LayerNorm, packed form for the SD-1.5 widths: a row of C = 8 * LPR * CPL elements is shared by LPR lanes (8 / 16 / 32 for
// C = 320 / 640 / 1280, five 16-byte chunks each), so one wave normalises 64 / LPR rows with EVERY lane busy -- the
// one-wave-per-row form above leaves 24 of 64 lanes idle at C = 320.  Lane `sub` of a row owns chunks sub + LPR * k: one
// wave-instruction reads LPR * 16 contiguous bytes of each of its rows.  Exact two-pass variance; the LPR-lane sums run on
// the DPP path (quad permutes, row mirrors; one cross-row step for LPR = 32) in a fixed order.
template <int LPR>
__device__ __forceinline__ float lane_group_sum(float v) {
    static_assert(LPR == 8 || LPR == 16 || LPR == 32, "lanes per row");
    v += dpp_f32<0xB1>(v);
    v += dpp_f32<0x4E>(v);
    v += dpp_f32<0x141>(v);
    if (LPR >= 16) v += dpp_f32<0x140>(v);
    if (LPR >= 32) v += __shfl_xor(v, 16, 64);
    return v;
}

template <typename T, int LPR, int CPL>
__global__ __launch_bounds__(kThreads) void layernorm_packed_kernel(const T* __restrict__ X, T* __restrict__ Y, int64_t rows, int C,
                                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                    float eps) {
    GMD_WG_TRACE_SCOPE(WGK_LN_PACKED);
    constexpr int V = Elem<T>::kVec, RPW = 64 / LPR;
    constexpr int CC = V * LPR * CPL;  // the row length this instance serves (host-checked == C)
    // Round 5: gamma / beta of the workgroup's rows come from LDS (staged once, under the flight of the row loads).  Read from
    // global memory per 16-byte chunk they were 20 of the 30 vector-memory instructions of a thread -- twice the L1 traffic of the
    // rows themselves -- and were requested only after the second reduction.
    __shared__ __attribute__((aligned(16))) float s_gb[2 * CC];
    const int lane = threadIdx.x & 63, sub = lane % LPR;
    const int64_t row = ((int64_t)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6)) * RPW + lane / LPR;
    const bool live = row < rows;  // whole lane groups are live or dead: the DPP sums never mix rows
    uint4 raw[CPL];
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        const int64_t rr = live ? row : rows - 1;  // (clamped: a dead lane group re-reads the last row, never stored)
        raw[k] = *reinterpret_cast<const uint4*>(X + rr * C + (int64_t)(sub + LPR * k) * V);
    }
    for (int c4 = threadIdx.x; c4 < CC / 4; c4 += kThreads) {
        *reinterpret_cast<float4*>(s_gb + 4 * c4) = *reinterpret_cast<const float4*>(gamma + 4 * c4);
        *reinterpret_cast<float4*>(s_gb + CC + 4 * c4) = *reinterpret_cast<const float4*>(beta + 4 * c4);
    }
    float v[CPL][V];
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        const unsigned w[4] = {raw[k].x, raw[k].y, raw[k].z, raw[k].w};
        if constexpr (sizeof(T) == 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[k][e] = __uint_as_float(w[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) Half<T>::unpack2(w[e], v[k][2 * e], v[k][2 * e + 1]);
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < CPL; ++k)
#pragma unroll
        for (int j = 0; j < V; ++j) s += v[k][j];
    const float mean = lane_group_sum<LPR>(s) / (float)C;
    float s2 = 0.0f;
#pragma unroll
    for (int k = 0; k < CPL; ++k)
#pragma unroll
        for (int j = 0; j < V; ++j) { const float d = v[k][j] - mean; s2 += d * d; }
    const float rstd = rsqrtf(lane_group_sum<LPR>(s2) / (float)C + eps);
    __syncthreads();  // gamma / beta are staged
    if (!live) return;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        const int c0 = (sub + LPR * k) * V;
        float ga[V], be[V], o[V];
#pragma unroll
        for (int e = 0; e < V; e += 4) {
            const float4 g4 = *reinterpret_cast<const float4*>(s_gb + c0 + e), b4 = *reinterpret_cast<const float4*>(s_gb + CC + c0 + e);
            ga[e] = g4.x; ga[e + 1] = g4.y; ga[e + 2] = g4.z; ga[e + 3] = g4.w;
            be[e] = b4.x; be[e + 1] = b4.y; be[e + 2] = b4.z; be[e + 3] = b4.w;
        }
#pragma unroll
        for (int j = 0; j < V; ++j) o[j] = (v[k][j] - mean) * rstd * ga[j] + be[j];
        store_vec(Y + row * C + c0, o);
    }
}

// ---------------------------------------------------------------------------------------------
// Row softmax (float32 logits -> T probabilities), one block per row, three passes.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kThreads) void softmax_rows_kernel(const float* __restrict__ S, int64_t lds_, T* __restrict__ P,
                                                                int64_t ldp, int cols_all, float scale, int causal_nq) {
    GMD_WG_TRACE_SCOPE(WGK_OTHER);
    __shared__ float red[kThreads / 64];
    const int64_t row = blockIdx.x;
    // causal rows attend columns 0 .. (row % Nq) only; the rest of the row is written as zeros
    const int cols = causal_nq > 0 ? min(cols_all, (int)(row % causal_nq) + 1) : cols_all;
    const float* s = S + row * lds_;
    T* p = P + row * ldp;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float m = -INFINITY;
    for (int c = threadIdx.x; c < cols; c += kThreads) m = fmaxf(m, s[c] * scale);
    m = wave_max(m);
    if (lane == 0) red[wid] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.0f;
    for (int c = threadIdx.x; c < cols; c += kThreads) sum += expf(s[c] * scale - m);
    sum = wave_sum(sum);
    if (lane == 0) red[wid] = sum;
    __syncthreads();
    sum = (red[0] + red[1]) + (red[2] + red[3]);
    const float inv = 1.0f / sum;
    for (int c = threadIdx.x; c < (int)ldp; c += kThreads)
        Elem<T>::st(p + c, c < cols ? expf(s[c] * scale - m) * inv : 0.0f);
}

inline int grid_for(int64_t n) {
    int64_t g = (n + kThreads - 1) / kThreads;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

bool gmd_gn_from_slabs_ok(int dtype, int B, int64_t HW, int C, int G) {
    if (!gmd_known_dtype(dtype) || B <= 0 || B > 65535 || HW <= 0 || C <= 0 || G <= 0 || C % G) return false;
    const int esz = gmd_is_half(dtype) ? 2 : 4, cpg = C / G;
    if ((cpg * esz) % 16 || (C * esz) % 16 || HW * (int64_t)C >= (1LL << 31)) return false;
    return HW * (int64_t)(cpg * esz / 16) <= (int64_t)kThreads * 12;
}

int gmd_launch_gn_from_slabs(const float* ws, int ksplit, float alpha, const float* bias, const float* rowbias, int64_t ldrb,
                             const void* residual, void* Yraw, void* Ynorm, int dtype, int B, int64_t HW, int C, int G, float eps,
                             const float* gamma, const float* beta, int silu, hipStream_t stream) {
    GMD_REQUIRE(gmd_gn_from_slabs_ok(dtype, B, HW, C, G), "GroupNorm from split-K slabs: shape not supported (B=%d HW=%lld C=%d G=%d)", B,
                (long long)HW, C, G);
    GMD_REQUIRE(ws && ksplit >= 1 && Ynorm && gamma && beta && gmd_aligned16(ws) && gmd_aligned16(Ynorm) && (Yraw == nullptr || gmd_aligned16(Yraw)) &&
                    (residual == nullptr || gmd_aligned16(residual)),
                "GroupNorm from split-K slabs: null or unaligned pointer");
    SlabSource sp{ws, ksplit, (int64_t)B * HW * C, alpha, bias, rowbias, ldrb > 0 ? ldrb : C, residual};
    const int esz = gmd_is_half(dtype) ? 2 : 4;
    const int64_t accesses = HW * (int64_t)((C / G) * esz / 16);
    dim3 grid(G, B);
    gmd_for_dtype(dtype, [&](auto tag) {
        using T = decltype(tag);
        if (accesses <= (int64_t)kThreads * 2) gn_slab_kernel<T, 2><<<grid, kThreads, 0, stream>>>(sp, (T*)Yraw, (T*)Ynorm, (int)HW, C, G, eps, gamma, beta, silu);
        else if (accesses <= (int64_t)kThreads * 6) gn_slab_kernel<T, 6><<<grid, kThreads, 0, stream>>>(sp, (T*)Yraw, (T*)Ynorm, (int)HW, C, G, eps, gamma, beta, silu);
        else gn_slab_kernel<T, 12><<<grid, kThreads, 0, stream>>>(sp, (T*)Yraw, (T*)Ynorm, (int)HW, C, G, eps, gamma, beta, silu);
    });
    GMD_CHECK_LAUNCH("gmd_conv3x3_groupnorm (GroupNorm from slabs)");
    return GMD_OK;
}

extern "C" {

int gmd_groupnorm_nsplit(int64_t HW) {
    // >= 32 pixels per block (64 from 64x64 latents up: the apply workgroups fold these partials again, and the coarser split
    // wins there -- tools/bench_gn.py); B * nsplit blocks should cover the 256 CUs several times
    int64_t n = HW >= 4096 ? HW / 64 : HW / 32;
    if (n < 1) n = 1;
    if (n > 256) n = 256;
    return (int)n;
}

int gmd_groupnorm_stats(const void* X, int dtype, int B, int64_t HW, int C, int G, float eps, const float* gamma,
                        const float* beta, float* workspace, float* scale_shift, gmd_stream_t stream) {
    GMD_REQUIRE(B > 0 && HW > 0 && C > 0 && G > 0 && G <= 64, "gmd_groupnorm_stats: bad shape B=%d HW=%lld C=%d G=%d", B, (long long)HW, C, G);
    GMD_REQUIRE(C % G == 0, "gmd_groupnorm_stats: C=%d not divisible by G=%d", C, G);
    GMD_REQUIRE(X && gamma && beta && workspace && scale_shift, "gmd_groupnorm_stats: null pointer");
    GMD_REQUIRE(gmd_aligned16(X), "gmd_groupnorm_stats: X not 16-byte aligned");
    const int V = gmd_is_half(dtype) ? 8 : 4;
    GMD_REQUIRE(gmd_known_dtype(dtype), "gmd_groupnorm_stats: bad dtype %d", dtype);
    GMD_REQUIRE(C % V == 0, "gmd_groupnorm_stats: C=%d must be a multiple of %d", C, V);
    const int nsplit = gmd_groupnorm_nsplit(HW);
    const int CV = C / V, CVB = CV < kThreads ? CV : kThreads, PY = kThreads / CVB;
    const size_t smem = (size_t)PY * C * 2 * sizeof(float);
    GMD_REQUIRE(smem <= 64 * 1024, "gmd_groupnorm_stats: C=%d too large", C);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(nsplit, B);
    gmd_for_dtype(dtype, [&](auto tag) {
        using T = decltype(tag);
        gn_partial_kernel<T><<<grid, kThreads, smem, s>>>((const T*)X, HW, C, G, nsplit, workspace);
    });
    GMD_CHECK_LAUNCH("gmd_groupnorm_stats(partial)");
    gn_finalize_kernel<<<B, kThreads, 0, s>>>(workspace, nsplit, HW, C, G, eps, gamma, beta, scale_shift);
    GMD_CHECK_LAUNCH("gmd_groupnorm_stats(finalize)");
    return GMD_OK;
}

int gmd_groupnorm_split(const void* X, void* Y, int dtype, int B, int64_t HW, int C, int G, float eps, const float* gamma,
                        const float* beta, float* workspace, int silu, gmd_stream_t stream) {
    GMD_REQUIRE(B > 0 && HW > 0 && C > 0 && G > 0 && G <= 64 && C % G == 0, "gmd_groupnorm_split: bad shape B=%d HW=%lld C=%d G=%d", B, (long long)HW, C, G);
    GMD_REQUIRE(X && Y && gamma && beta && workspace, "gmd_groupnorm_split: null pointer");
    GMD_REQUIRE(gmd_aligned16(X) && gmd_aligned16(Y), "gmd_groupnorm_split: pointers must be 16-byte aligned");
    const bool split_out = dtype == GMD_F32SA;  // float32 in, output stored pre-split
    GMD_REQUIRE(gmd_known_dtype(dtype) || split_out, "gmd_groupnorm_split: bad dtype %d", dtype);
    GMD_REQUIRE(B <= 65535, "gmd_groupnorm_split: batch too large");
    const int V = gmd_is_half(dtype) ? 8 : 4;
    GMD_REQUIRE(C % V == 0 && (!split_out || C % 32 == 0), "gmd_groupnorm_split: C=%d must be a multiple of %d", C, split_out ? 32 : V);
    const int nsplit = gmd_groupnorm_nsplit(HW);
    const int CV = C / V, CVB = CV < kThreads ? CV : kThreads, PY = kThreads / CVB;
    const size_t smem = (size_t)PY * C * 2 * sizeof(float);
    GMD_REQUIRE(smem <= 64 * 1024 && (size_t)C * 8 <= 64 * 1024, "gmd_groupnorm_split: C=%d too large", C);
    GMD_REQUIRE(HW * (int64_t)(C / V) < (1ll << 27), "gmd_groupnorm_split: one sample of %lld x %d exceeds the 32-bit byte offset of the apply pass", (long long)HW, C);
    hipStream_t s = (hipStream_t)stream;
    // apply slices: 256 * U vectors per workgroup (U <= 10 per thread, in flight in two halves), at least ~512 workgroups where the tensor allows
    const int64_t nvec = HW * (C / V);
    const int span_u = gn_apply_span_u(B, nvec), span = kThreads * span_u;
    const int64_t nb = (nvec + span - 1) / span;
    bool partial_ok = true;
    gmd_for_dtype(dtype, [&](auto tag) {
        using T = decltype(tag);
        gn_partial_kernel<T><<<dim3(nsplit, B), kThreads, smem, s>>>((const T*)X, HW, C, G, nsplit, workspace);
        if (hipGetLastError() != hipSuccess) { partial_ok = false; return; }
        gn_for_u(span_u, [&](auto uc) {
            constexpr int U = decltype(uc)::value;
            if constexpr (sizeof(T) == 4) {
                if (split_out) {
                    gn_apply_ws_kernel<T, true, U><<<dim3((unsigned)nb, B), kThreads, (size_t)C * 8, s>>>((const T*)X, (T*)Y, HW, C, G, nsplit, eps, workspace, gamma, beta, silu, span);
                    return;
                }
            }
            gn_apply_ws_kernel<T, false, U><<<dim3((unsigned)nb, B), kThreads, (size_t)C * 8, s>>>((const T*)X, (T*)Y, HW, C, G, nsplit, eps, workspace, gamma, beta, silu, span);
        });
    });
    if (!partial_ok) {
        gmd_set_error("gmd_groupnorm_split(partial): launch failed");
        return GMD_ERR_LAUNCH;
    }
    GMD_CHECK_LAUNCH("gmd_groupnorm_split(apply)");
    return GMD_OK;
}

int gmd_groupnorm_colstats(const void* X, void* Y, int dtype, int B, int64_t HW, int C, int G, float eps, const float* gamma,
                           const float* beta, const float* stats_a, int Ca, const float* stats_b, int bucket, int silu,
                           gmd_stream_t stream) {
    GMD_REQUIRE(B > 0 && HW > 0 && C > 0 && G > 0 && G <= 64 && C % G == 0, "gmd_groupnorm_colstats: bad shape B=%d HW=%lld C=%d G=%d", B, (long long)HW, C, G);
    GMD_REQUIRE(X && Y && gamma && beta && stats_a, "gmd_groupnorm_colstats: null pointer");
    GMD_REQUIRE(gmd_aligned16(X) && gmd_aligned16(Y), "gmd_groupnorm_colstats: pointers must be 16-byte aligned");
    const bool split_out = dtype == GMD_F32SA;  // float32 in, output stored pre-split
    GMD_REQUIRE(gmd_known_dtype(dtype) || split_out, "gmd_groupnorm_colstats: bad dtype %d", dtype);
    GMD_REQUIRE(B <= 65535, "gmd_groupnorm_colstats: batch too large");
    const int V = gmd_is_half(dtype) ? 8 : 4;
    GMD_REQUIRE(C % V == 0 && (!split_out || C % 32 == 0) && (size_t)C * 8 <= 64 * 1024, "gmd_groupnorm_colstats: C=%d must be a multiple of %d and at most 8192", C, split_out ? 32 : V);
    GMD_REQUIRE(HW % 64 == 0, "gmd_groupnorm_colstats: the statistics are per 64-row block, HW=%lld is not a multiple of 64", (long long)HW);
    GMD_REQUIRE(HW * (int64_t)(C / V) < (1ll << 27), "gmd_groupnorm_colstats: one sample of %lld x %d exceeds the 32-bit byte offset of the apply pass", (long long)HW, C);
    GMD_REQUIRE(bucket > 0 && (C / G) % bucket == 0 && Ca > 0 && Ca <= C && Ca % bucket == 0 && (C - Ca) % bucket == 0,
                "gmd_groupnorm_colstats: bucket=%d must divide the group size %d and both channel ranges (%d, %d)", bucket, C / G, Ca, C - Ca);
    GMD_REQUIRE(Ca == C || stats_b, "gmd_groupnorm_colstats: statistics of the second channel range are missing");
    GMD_REQUIRE((reinterpret_cast<uintptr_t>(stats_a) & 7) == 0 && (reinterpret_cast<uintptr_t>(stats_b) & 7) == 0, "gmd_groupnorm_colstats: statistics must be 8-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    // apply slices: 256 * U vectors per workgroup (U <= 10 per thread, in flight in two halves), at least ~512 workgroups where the tensor allows
    const int64_t nvec = HW * (C / V);
    const int span_u = gn_apply_span_u(B, nvec), span = kThreads * span_u;
    const int64_t nb = (nvec + span - 1) / span;
    gmd_for_dtype(dtype, [&](auto tag) {
        using T = decltype(tag);
        gn_for_u(span_u, [&](auto uc) {
            constexpr int U = decltype(uc)::value;
            if constexpr (sizeof(T) == 4) {
                if (split_out) {
                    gn_apply_cs_kernel<T, true, U><<<dim3((unsigned)nb, B), kThreads, (size_t)C * 8, s>>>((const T*)X, (T*)Y, HW, C, G, eps, stats_a, Ca,
                                                                                                         stats_b, bucket, gamma, beta, silu, span);
                    return;
                }
            }
            gn_apply_cs_kernel<T, false, U><<<dim3((unsigned)nb, B), kThreads, (size_t)C * 8, s>>>((const T*)X, (T*)Y, HW, C, G, eps, stats_a, Ca,
                                                                                               stats_b, bucket, gamma, beta, silu, span);
        });
    });
    GMD_CHECK_LAUNCH("gmd_groupnorm_colstats");
    return GMD_OK;
}

int gmd_groupnorm_apply(const void* X, void* Y, int dtype, int B, int64_t HW, int C, const float* scale_shift, int silu,
                        gmd_stream_t stream) {
    GMD_REQUIRE(B > 0 && HW > 0 && C > 0, "gmd_groupnorm_apply: bad shape");
    GMD_REQUIRE(X && Y && scale_shift, "gmd_groupnorm_apply: null pointer");
    GMD_REQUIRE(gmd_aligned16(X) && gmd_aligned16(Y) && gmd_aligned16(scale_shift), "gmd_groupnorm_apply: pointers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    GMD_REQUIRE(gmd_known_dtype(dtype), "gmd_groupnorm_apply: bad dtype %d", dtype);
    GMD_REQUIRE(C % (gmd_is_half(dtype) ? 8 : 4) == 0, "gmd_groupnorm_apply: C=%d must be a multiple of %d", C, gmd_is_half(dtype) ? 8 : 4);
    gmd_for_dtype(dtype, [&](auto tag) {
        using T = decltype(tag);
        gn_apply_kernel<T><<<grid_for((int64_t)B * HW * (C / Elem<T>::kVec)), kThreads, 0, s>>>((const T*)X, (T*)Y, B, HW, C, scale_shift, silu);
    });
    GMD_CHECK_LAUNCH("gmd_groupnorm_apply");
    return GMD_OK;
}

int gmd_groupnorm_fused(const void* X, void* Y, int dtype, int B, int64_t HW, int C, int G, float eps, const float* gamma,
                        const float* beta, int silu, gmd_stream_t stream) {
    GMD_REQUIRE(B > 0 && HW > 0 && C > 0 && G > 0 && C % G == 0, "gmd_groupnorm_fused: bad shape");
    GMD_REQUIRE(X && Y && gamma && beta, "gmd_groupnorm_fused: null pointer");
    GMD_REQUIRE(gmd_known_dtype(dtype), "gmd_groupnorm_fused: bad dtype %d", dtype);
    GMD_REQUIRE(B <= 65535, "gmd_groupnorm_fused: batch too large");
    const int cpg = C / G, epw = gmd_is_half(dtype) ? 2 : 1;
    GMD_REQUIRE(cpg % epw == 0 && C % epw == 0 && (reinterpret_cast<uintptr_t>(X) & 3) == 0 && (reinterpret_cast<uintptr_t>(Y) & 3) == 0,
                "gmd_groupnorm_fused: channels per group must be even for the 16-bit types");
    const int64_t slab_bytes = HW * cpg * (gmd_is_half(dtype) ? 2 : 4);
    if (slab_bytes > 128 * 1024 || HW * (int64_t)C >= (1LL << 31)) {
        gmd_set_error("gmd_groupnorm_fused: group slab of %lld bytes is too large for the single-launch kernel (use gmd_groupnorm_stats + _apply)", (long long)slab_bytes);
        return GMD_ERR_UNSUPPORTED;
    }
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(G, B);
    // widest access that divides a group's channel row and keeps every access aligned (row stride C, group offset g*cpg)
    const int esz = gmd_is_half(dtype) ? 2 : 4;
    const auto ok = [&](int vb) {
        return (cpg * esz) % vb == 0 && (C * esz) % vb == 0 && (reinterpret_cast<uintptr_t>(X) % vb) == 0 && (reinterpret_cast<uintptr_t>(Y) % vb) == 0;
    };
    const int vb = ok(16) ? 16 : ok(8) ? 8 : 4;
    // slabs of 5..12 accesses per thread (the 16x16-level GroupNorms: 20-40 KB) stay in registers: one memory round trip instead
    // of three dependent passes (tools/bench_gn.py: 11.0 -> 9.5 us at C = 1280, 21.2 -> 17.4 us at C = 2560).  Below that (8x8
    // level, <= 4 accesses) the three short passes are as fast; the apply loops of the two-launch kernels did NOT gain from the
    // same treatment (four loads in flight per thread, twice the workgroups: 24.8 us either way at 64x64) and were left alone.
    const int64_t accesses = HW * (int64_t)(cpg * esz / vb);
    const bool in_regs = vb >= 8 && accesses > (int64_t)kThreads * 4 && accesses <= (int64_t)kThreads * 12;
#define GMD_GN_FUSED(T, VB)                                                                                                                 \
    do {                                                                                                                                    \
        if (in_regs && VB >= 8)                                                                                                             \
            gn_fused_reg_kernel<T, (VB >= 8 ? VB : 8), 12><<<grid, kThreads, 0, s>>>((const T*)X, (T*)Y, (int)HW, C, G, eps, gamma, beta, silu); \
        else                                                                                                                                \
            gn_fused_kernel<T, VB><<<grid, kThreads, 0, s>>>((const T*)X, (T*)Y, (int)HW, C, G, eps, gamma, beta, silu);                   \
    } while (0)
    gmd_for_dtype(dtype, [&](auto tag) {
        using T = decltype(tag);
        if (vb == 16) GMD_GN_FUSED(T, 16); else if (vb == 8) GMD_GN_FUSED(T, 8); else GMD_GN_FUSED(T, 4);
    });
#undef GMD_GN_FUSED
    GMD_CHECK_LAUNCH("gmd_groupnorm_fused");
    return GMD_OK;
}

int gmd_layernorm(const void* X, void* Y, int dtype, int64_t rows, int C, const float* gamma, const float* beta, float eps,
                  gmd_stream_t stream) {
    GMD_REQUIRE(rows >= 0 && C > 0 && C <= 2048, "gmd_layernorm: bad shape rows=%lld C=%d", (long long)rows, C);
    if (rows == 0) return GMD_OK;
    GMD_REQUIRE(X && Y && gamma && beta, "gmd_layernorm: null pointer");
    GMD_REQUIRE(gmd_aligned16(X) && gmd_aligned16(Y), "gmd_layernorm: pointers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int grid = (int)((rows + 3) / 4);
    if (gmd_is_half(dtype)) {
        GMD_REQUIRE(C % 8 == 0, "gmd_layernorm: C=%d must be a multiple of 8", C);
        const bool vecp = gmd_aligned16(gamma) && gmd_aligned16(beta);
        gmd_for_dtype(dtype, [&](auto tag) {
            using T = decltype(tag);
            if constexpr (sizeof(T) == 2) {
                // SD-1.5 widths: every lane busy (LPR lanes per row, five 16-byte chunks each; measured, tools/bench_graph_ops.py:
                // 32768 x 320 12.8 -> 10.3 us, 8192 x 640 7.9 -> 7.5 us; at C = 1280 the 32-lane form needs a cross-row shuffle
                // and LOSES to one wave per row, 4.9 -> 5.8 us, so it is not used there)
                if (vecp && C == 320)
                    layernorm_packed_kernel<T, 8, 5><<<(int)((rows + 31) / 32), kThreads, 0, s>>>((const T*)X, (T*)Y, rows, C, gamma, beta, eps);
                else if (vecp && C == 640)
                    layernorm_packed_kernel<T, 16, 5><<<(int)((rows + 15) / 16), kThreads, 0, s>>>((const T*)X, (T*)Y, rows, C, gamma, beta, eps);
                // narrow rows (one 16-byte chunk per lane) and many of them: four rows per wave keep four loads in flight
                else if (C <= 512 && rows >= 8192)
                    layernorm_kernel<T, 1, 4><<<(int)((rows + 15) / 16), kThreads, 0, s>>>((const T*)X, (T*)Y, rows, C, gamma, beta, eps);
                else if (C <= 1024 && rows >= 4096)
                    layernorm_kernel<T, 2, 2><<<(int)((rows + 7) / 8), kThreads, 0, s>>>((const T*)X, (T*)Y, rows, C, gamma, beta, eps);
                else
                    layernorm_kernel<T, 4, 1><<<grid, kThreads, 0, s>>>((const T*)X, (T*)Y, rows, C, gamma, beta, eps);
            }
        });
    } else if (dtype == GMD_F32) {
        GMD_REQUIRE(C % 4 == 0, "gmd_layernorm: C=%d must be a multiple of 4", C);
        layernorm_kernel<float, 8, 1><<<grid, kThreads, 0, s>>>((const float*)X, (float*)Y, rows, C, gamma, beta, eps);
    } else if (dtype == GMD_F32SA) {  // float32 in, output stored pre-split (the A operand of the projections that follow)
        GMD_REQUIRE(C % 32 == 0, "gmd_layernorm: a pre-split output needs C=%d to be a multiple of 32", C);
        layernorm_kernel<float, 8, 1, true><<<grid, kThreads, 0, s>>>((const float*)X, (float*)Y, rows, C, gamma, beta, eps);
    } else {
        GMD_REQUIRE(false, "gmd_layernorm: bad dtype %d", dtype);
    }
    GMD_CHECK_LAUNCH("gmd_layernorm");
    return GMD_OK;
}

int gmd_softmax_rows(const float* S, int64_t lds_, void* P, int out_dtype, int64_t ldp, int64_t rows, int cols, float scale,
                     int causal_nq, gmd_stream_t stream) {
    GMD_REQUIRE(causal_nq >= 0, "gmd_softmax_rows: bad causal_nq");
    GMD_REQUIRE(rows >= 0 && cols > 0 && lds_ >= cols && ldp >= cols, "gmd_softmax_rows: bad shape");
    if (rows == 0) return GMD_OK;
    GMD_REQUIRE(S && P, "gmd_softmax_rows: null pointer");
    GMD_REQUIRE(rows < (1LL << 31), "gmd_softmax_rows: too many rows");
    hipStream_t s = (hipStream_t)stream;
    if (out_dtype == GMD_BF16)
        softmax_rows_kernel<bf16_t><<<(int)rows, kThreads, 0, s>>>(S, lds_, (bf16_t*)P, ldp, cols, scale, causal_nq);
    else if (out_dtype == GMD_F16)
        softmax_rows_kernel<f16_t><<<(int)rows, kThreads, 0, s>>>(S, lds_, (f16_t*)P, ldp, cols, scale, causal_nq);
    else if (out_dtype == GMD_F32)
        softmax_rows_kernel<float><<<(int)rows, kThreads, 0, s>>>(S, lds_, (float*)P, ldp, cols, scale, causal_nq);
    else
        GMD_REQUIRE(false, "gmd_softmax_rows: bad dtype %d", out_dtype);
    GMD_CHECK_LAUNCH("gmd_softmax_rows");
    return GMD_OK;
}

}  // extern "C"

GMD_WG_TRACE_SETTER(norm)

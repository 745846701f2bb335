// Small HBM-bound elementwise kernels of the UNet: GEGLU gate, sinusoidal timestep embedding,
// channel concat (skip connections) and dtype cast.  16-byte accesses throughout.
#include "gmd_common.h"
#include <math.h>

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <typename T>
__global__ __launch_bounds__(kThreads) void geglu_kernel(const T* __restrict__ X, T* __restrict__ Y, int64_t rows, int F) {
    GMD_WG_TRACE_SCOPE(WGK_OTHER);
    constexpr int V = Elem<T>::kVec;
    const int FV = F / V;
    const int64_t total = rows * FV;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / FV;
        const int f = (int)(i - r * FV) * V;
        float h[V], g[V];
        load_vec(X + r * 2 * F + f, h);
        load_vec(X + r * 2 * F + F + f, g);
#pragma unroll
        for (int j = 0; j < V; ++j) h[j] = h[j] * gelu_erf(g[j]);
        store_vec(Y + r * F + f, h);
    }
}

// diffusers get_timestep_embedding: emb_i = t * exp(-ln(10000) * i / (half - shift)); [sin | cos], flipped to [cos | sin]
template <typename T>
__global__ void temb_kernel(const float* __restrict__ t_dev, T* __restrict__ out, int B, int dim, int flip, float shift) {
    GMD_WG_TRACE_SCOPE(WGK_TEMB);
    const int half = dim / 2;
    const float t = *t_dev;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * half; i += gridDim.x * blockDim.x) {
        const int b = i / half, k = i - b * half;
        const float e = expf(-9.210340371976184f * (float)k / ((float)half - shift));
        const float a = t * e;
        const float sn = sinf(a), cs = cosf(a);
        T* o = out + (int64_t)b * dim;
        Elem<T>::st(o + (flip ? half + k : k), sn);
        Elem<T>::st(o + (flip ? k : half + k), cs);
    }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void concat_kernel(const T* __restrict__ A, int Ca, const T* __restrict__ Bm, int Cb,
                                                          T* __restrict__ out, int64_t rows) {
    GMD_WG_TRACE_SCOPE(WGK_CONCAT);
    constexpr int V = Elem<T>::kVec;
    const int CV = (Ca + Cb) / V, CaV = Ca / V;
    const int64_t total = rows * CV;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / CV;
        const int c = (int)(i - r * CV);
        const uint4 v = c < CaV ? *reinterpret_cast<const uint4*>(A + r * Ca + (int64_t)c * V)
                                : *reinterpret_cast<const uint4*>(Bm + r * Cb + (int64_t)(c - CaV) * V);
        *reinterpret_cast<uint4*>(out + i * V) = v;
    }
}

template <typename TI, typename TO>
__global__ __launch_bounds__(kThreads) void cast_kernel(const TI* __restrict__ in, TO* __restrict__ out, int64_t n) {
    GMD_WG_TRACE_SCOPE(WGK_CAST);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        Elem<TO>::st(out + i, Elem<TI>::ld(in + i));
}

// out[r, c] = table[ids[r], c] + pos[r % T, c]: one row per (batch, token); an id outside [0, vocab) reads row 0 (the host
// front end rejects such ids before the launch -- this only keeps a wild pointer impossible)
template <typename T>
__global__ __launch_bounds__(kThreads) void embedding_kernel(const int32_t* __restrict__ ids, const T* __restrict__ table,
                                                             const T* __restrict__ pos, T* __restrict__ out, int64_t rows, int Tn, int C,
                                                             int vocab) {
    GMD_WG_TRACE_SCOPE(WGK_OTHER);
    const int64_t total = rows * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / C;
        const int c = (int)(i - r * C);
        int id = ids[r];
        if (id < 0 || id >= vocab) id = 0;
        Elem<T>::st(out + i, Elem<T>::ld(table + (int64_t)id * C + c) + Elem<T>::ld(pos + (int64_t)(r % Tn) * C + c));
    }
}

inline int grid_for(int64_t n) {
    int64_t g = (n + kThreads - 1) / kThreads;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (int)g;
}

// out[0:n] = out[n:2n] = in[0:n] in 16-byte chunks: the CFG shared prefix duplicates a tensor for the two text conditionings
// (stable_diffusion_dual_unet.py:1045 torch.cat([latents] * 2), applied where the two halves first differ).  One read, two
// writes per chunk; the runtime's generic device-to-device copy took 2 x 83 us for the 2 x 10.5 MB of a level-0 tensor inside
// the two-stream pipeline (rocprofv3, profiles/r03_*), this kernel streams it at the HBM rate.
__global__ __launch_bounds__(kThreads) void dup_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, int64_t n) {
    GMD_WG_TRACE_SCOPE(WGK_DUP);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint4 v = in[i];
        out[i] = v;
        out[n + i] = v;
    }
}

}  // namespace

extern "C" {

int gmd_geglu(const void* X, void* Y, int dtype, int64_t rows, int F, gmd_stream_t stream) {
    GMD_REQUIRE(rows >= 0 && F > 0, "gmd_geglu: bad shape");
    if (rows == 0) return GMD_OK;
    GMD_REQUIRE(X && Y && gmd_aligned16(X) && gmd_aligned16(Y), "gmd_geglu: null or unaligned pointer");
    hipStream_t s = (hipStream_t)stream;
    GMD_REQUIRE(gmd_known_dtype(dtype), "gmd_geglu: bad dtype %d", dtype);
    GMD_REQUIRE(F % (gmd_is_half(dtype) ? 8 : 4) == 0, "gmd_geglu: F=%d must be a multiple of %d", F, gmd_is_half(dtype) ? 8 : 4);
    gmd_for_dtype(dtype, [&](auto tag) {
        using T = decltype(tag);
        geglu_kernel<T><<<grid_for(rows * (F / Elem<T>::kVec)), kThreads, 0, s>>>((const T*)X, (T*)Y, rows, F);
    });
    GMD_CHECK_LAUNCH("gmd_geglu");
    return GMD_OK;
}

int gmd_timestep_embedding(const float* t_dev, void* out, int dtype, int B, int dim, int flip_sin_to_cos, float freq_shift,
                           gmd_stream_t stream) {
    GMD_REQUIRE(B > 0 && dim > 0 && dim % 2 == 0, "gmd_timestep_embedding: bad shape B=%d dim=%d", B, dim);
    GMD_REQUIRE(t_dev && out, "gmd_timestep_embedding: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int grid = (B * dim / 2 + 255) / 256;
    GMD_REQUIRE(gmd_known_dtype(dtype), "gmd_timestep_embedding: bad dtype %d", dtype);
    gmd_for_dtype(dtype, [&](auto tag) {
        using T = decltype(tag);
        temb_kernel<T><<<grid, 256, 0, s>>>(t_dev, (T*)out, B, dim, flip_sin_to_cos, freq_shift);
    });
    GMD_CHECK_LAUNCH("gmd_timestep_embedding");
    return GMD_OK;
}

int gmd_concat_channels(const void* A, int Ca, const void* Bm, int Cb, void* out, int dtype, int64_t rows, gmd_stream_t stream) {
    GMD_REQUIRE(rows >= 0 && Ca > 0 && Cb > 0, "gmd_concat_channels: bad shape");
    if (rows == 0) return GMD_OK;
    GMD_REQUIRE(A && Bm && out, "gmd_concat_channels: null pointer");
    GMD_REQUIRE(gmd_aligned16(A) && gmd_aligned16(Bm) && gmd_aligned16(out), "gmd_concat_channels: pointers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    GMD_REQUIRE(gmd_known_dtype(dtype), "gmd_concat_channels: bad dtype %d", dtype);
    const int cv = gmd_is_half(dtype) ? 8 : 4;
    GMD_REQUIRE(Ca % cv == 0 && Cb % cv == 0, "gmd_concat_channels: channel counts must be multiples of %d", cv);
    gmd_for_dtype(dtype, [&](auto tag) {
        using T = decltype(tag);
        concat_kernel<T><<<grid_for(rows * ((Ca + Cb) / cv)), kThreads, 0, s>>>((const T*)A, Ca, (const T*)Bm, Cb, (T*)out, rows);
    });
    GMD_CHECK_LAUNCH("gmd_concat_channels");
    return GMD_OK;
}

int gmd_embedding_lookup(const int32_t* ids, const void* table, const void* pos, void* out, int dtype, int64_t rows, int T_, int C,
                         int vocab, gmd_stream_t stream) {
    GMD_REQUIRE(rows >= 0 && T_ > 0 && C > 0 && vocab > 0, "gmd_embedding_lookup: bad shape rows=%lld T=%d C=%d vocab=%d", (long long)rows, T_, C, vocab);
    if (rows == 0) return GMD_OK;
    GMD_REQUIRE(ids && table && pos && out, "gmd_embedding_lookup: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int grid = grid_for(rows * C);
    GMD_REQUIRE(gmd_known_dtype(dtype), "gmd_embedding_lookup: bad dtype %d", dtype);
    gmd_for_dtype(dtype, [&](auto tag) {
        using T = decltype(tag);
        embedding_kernel<T><<<grid, kThreads, 0, s>>>(ids, (const T*)table, (const T*)pos, (T*)out, rows, T_, C, vocab);
    });
    GMD_CHECK_LAUNCH("gmd_embedding_lookup");
    return GMD_OK;
}

int gmd_dup_batch(const void* in, void* out, int64_t bytes, gmd_stream_t stream) {
    GMD_REQUIRE(bytes >= 0 && bytes % 16 == 0, "gmd_dup_batch: byte count must be a non-negative multiple of 16");
    if (bytes == 0) return GMD_OK;
    GMD_REQUIRE(in && out && gmd_aligned16(in) && gmd_aligned16(out), "gmd_dup_batch: null or unaligned pointer");
    const int64_t n = bytes / 16;
    dup_kernel<<<grid_for(n), kThreads, 0, (hipStream_t)stream>>>((const uint4*)in, (uint4*)out, n);
    GMD_CHECK_LAUNCH("gmd_dup_batch");
    return GMD_OK;
}

int gmd_cast(const void* in, int in_dtype, void* out, int out_dtype, int64_t n, gmd_stream_t stream) {
    GMD_REQUIRE(n >= 0, "gmd_cast: negative n");
    if (n == 0) return GMD_OK;
    GMD_REQUIRE(in && out, "gmd_cast: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int grid = grid_for(n);
    GMD_REQUIRE(gmd_known_dtype(in_dtype) && gmd_known_dtype(out_dtype), "gmd_cast: bad dtypes %d -> %d", in_dtype, out_dtype);
    gmd_for_dtype(in_dtype, [&](auto ti) {
        using TI = decltype(ti);
        gmd_for_dtype(out_dtype, [&](auto to) {
            using TO = decltype(to);
            cast_kernel<TI, TO><<<grid, kThreads, 0, s>>>((const TI*)in, (TO*)out, n);
        });
    });
    GMD_CHECK_LAUNCH("gmd_cast");
    return GMD_OK;
}

}  // extern "C"


// ------------------------------------------------------------------------------------------------
// device-time stamp (measurement only): one thread writes the constant-rate 100 MHz counter (s_memrealtime, common to the whole
// device) into *slot.  Launched between the kernels of BOTH streams of the pipeline -- also from inside captured HIP graphs -- it gives
// the concurrent timeline of the shipped two-stream path, which rocprofv3 cannot (its tracing serialises the streams).
// ------------------------------------------------------------------------------------------------
namespace {
__global__ void stamp_kernel(unsigned long long* base, const int* row, int stride, int k) {
    GMD_WG_TRACE_SCOPE(WGK_STAMP);
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    base[(row ? (long long)*row * stride : 0) + k] = t;
}
}  // namespace

extern "C" int gmd_stamp(uint64_t* base, const int* row, int stride, int k, gmd_stream_t stream) {
    GMD_REQUIRE(base != nullptr && (reinterpret_cast<uintptr_t>(base) & 7) == 0 && stride >= 0 && k >= 0, "gmd_stamp: bad argument");
    stamp_kernel<<<1, 1, 0, (hipStream_t)stream>>>(reinterpret_cast<unsigned long long*>(base), row, stride, k);
    GMD_CHECK_LAUNCH("gmd_stamp");
    return GMD_OK;
}

GMD_WG_TRACE_SETTER(elementwise)
#ifdef GMD_WG_TRACE
// Diagnostic build only (not part of the product ABI): point every translation unit's trace hook at `ring` (a GmdWgTraceHeader followed
// by capacity x 32-byte records, zeroed by the caller; nullptr switches the trace off).  Not stream-ordered: call it with the device idle.
extern "C" int gmd_wg_trace_set_gemm(void*);
extern "C" int gmd_wg_trace_set_attention(void*);
extern "C" int gmd_wg_trace_set_norm(void*);
extern "C" int gmd_wg_trace_set_ff_fused(void*);
extern "C" int gmd_wg_trace_set_latent_step(void*);
extern "C" int gmd_wg_trace_enable(void* ring) {
    return gmd_wg_trace_set_gemm(ring) | gmd_wg_trace_set_attention(ring) | gmd_wg_trace_set_norm(ring) | gmd_wg_trace_set_ff_fused(ring) |
           gmd_wg_trace_set_latent_step(ring) | gmd_wg_trace_set_elementwise(ring);
}
#endif

// Flash-style attention on float32 tensors with the two contractions on the matrix cores as three float16 products
// each (the float32 operand x taken as hi + lo, hi = f16(x), lo = f16(x - hi); a*b ~ a_hi*b_hi + a_hi*b_lo + a_lo*b_hi with
// float32 accumulation -- see gemm_split.hip).  The attention of the float32 ("split") path: the reference's own float32
// numerics (scripts/inference/experiments/formal_improved.py:199) at matrix-core speed, and without the [heads, Nq, Nk]
// score tensor the GEMM -> softmax -> GEMM composition writes and re-reads (4.3 GB per level-0 self-attention at batch 8).
//
// Same "transposed" formulation as attention.hip (S^T = K Q^T, O^T += V^T P^T with mfma_f32_32x32x16_f16: the query sits on
// the lane, so the softmax state is per-lane and the S^T accumulator tile is the B operand of the second product), with
//   * K [64 keys][d] and V^T [d][64 keys] tiles staged through registers: float32 global loads, split ONCE per workgroup
//     into float16 hi / lo planes in LDS (the four waves share them);
//   * Q split once per wave into hi / lo fragments kept in registers;
//   * the classic online softmax in float32 (running maximum first, no lagged stabiliser), exp2 on scores scaled by
//     scale*log2(e); P is carried as 2^11 * p so that its lo half stays a normal float16 down to p = 2^-14 of the row
//     maximum (the factor cancels against the row sum, which is accumulated from the same float32 values);
//   * 2 LDS stages for d <= 64 (two workgroups per CU), one stage with two barriers per tile for d = 80 / 160.
#include "gemm_shared.h"

namespace {

struct AttnSplitParams {
    const float* Q;
    const float* K;
    const float* Vt;
    float* O;
    int Nq, Nk, H;
    int64_t ldq, ldk, ldvt, ldo, sQ, sK, sVt, sO;
    float scale_log2;  // softmax scale * log2(e)
    int o_split;       // store O pre-split for the out-projection (GMD_F32SA: O contiguous [B][Nq][ldo], ldo % 32 == 0)
};

constexpr int KV = 64;      // keys per tile
constexpr int VROW = 136;   // bytes per V^T LDS row: 64 keys * 2 B + 8 B pad (conflict-free b64 reads)
constexpr float kNegBig = -1.0e30f;
constexpr float kPShift = 11.0f;  // P is carried as 2^11 * p

typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;

__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
    const f16x2_t h = {(_Float16)a, (_Float16)b};
    const unsigned hw = __builtin_bit_cast(unsigned, h);
    float ra, rb;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(hw), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(hw), "v"(b));
    const f16x2_t l = {(_Float16)ra, (_Float16)rb};
    hi = hw;
    lo = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ f32x16 mfma_h(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <int D>
__global__ __launch_bounds__(256, D <= 64 ? 2 : 1) void attn_split_kernel(const AttnSplitParams p) {
    constexpr int DK = (D + 15) / 16;         // 16-wide k-steps of Q K^T over d
    constexpr int DT = (D + 31) / 32;         // 32-row tiles of O^T over d
    constexpr int KROW = (2 * DK + 1) * 16;   // bytes per K LDS row (float16): odd number of 16-byte slots
    constexpr int DC = D / 4;                 // float4 chunks per K row in global memory
    constexpr int NST = D <= 64 ? 2 : 1;
    static_assert(D % 8 == 0, "head dim must be a multiple of 8");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int kKPlane = KV * KROW, kVPlane = DT * 32 * VROW;
    constexpr int kStageBytes = 2 * kKPlane + 2 * kVPlane;  // K hi, K lo, V^T hi, V^T lo
    constexpr int NKC = (KV * DC + 255) / 256;               // float4 K chunks staged per thread
    constexpr int NVC = (D * 16 + 255) / 256;                // float4 V^T chunks staged per thread (16 per row of 64 keys)

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    int qb, head, b;
    {   // XCD-aware order (attention.hip): the query blocks of one (batch, head) share one L2
        const int nwg = gridDim.x, id = blockIdx.x;
        const int qq = nwg >> 3, rr = nwg & 7, xcd = id & 7, j = id >> 3;
        const int L = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + j;
        const int nqb = (p.Nq + 127) / 128;
        qb = L % nqb;
        const int t = L / nqb;
        head = t % p.H;
        b = t / p.H;
    }
    const int q = qb * 128 + wid * 32 + r;
    const bool qvalid = q < p.Nq;
    const float* Qb = p.Q + (int64_t)b * p.sQ + (int64_t)head * D;
    const float* Kb = p.K + (int64_t)b * p.sK + (int64_t)head * D;
    const float* Vb = p.Vt + (int64_t)b * p.sVt + (int64_t)head * D * p.ldvt;

    // zero the LDS padding that is read but never staged: K columns [D, DK*16), V^T rows [D, DT*32), both planes, every stage
#pragma unroll
    for (int st = 0; st < NST; ++st) {
        unsigned char* base = smem + st * kStageBytes;
        if (DK * 16 > D) {
            for (int i = tid; i < 2 * KV; i += 256)
                *reinterpret_cast<uint4*>(base + (i >> 6) * kKPlane + (i & 63) * KROW + (D / 8) * 16) = make_uint4(0, 0, 0, 0);
        }
        if constexpr (DT * 32 > D) {
            constexpr int kPad = (DT * 32 - D) * (VROW / 8);
            for (int i = tid; i < 2 * kPad; i += 256) {
                const int pl = i / kPad, k = i - pl * kPad;
                const int row = D + k / (VROW / 8), c = k % (VROW / 8);
                *reinterpret_cast<uint2*>(base + 2 * kKPlane + pl * kVPlane + row * VROW + c * 8) = make_uint2(0, 0);
            }
        }
    }

    // Q fragments (B operand), split once: lane (r, hh) element j = Q[q][16 s + 8 hh + j]
    uint4 qh[DK], ql[DK];
#pragma unroll
    for (int s = 0; s < DK; ++s) {
        const int d0 = 16 * s + 8 * hh;
        float4 x0 = make_float4(0.f, 0.f, 0.f, 0.f), x1 = x0;
        if (qvalid && d0 + 8 <= D) {
            x0 = *reinterpret_cast<const float4*>(Qb + (int64_t)q * p.ldq + d0);
            x1 = *reinterpret_cast<const float4*>(Qb + (int64_t)q * p.ldq + d0 + 4);
        }
        split2(x0.x, x0.y, qh[s].x, ql[s].x);
        split2(x0.z, x0.w, qh[s].y, ql[s].y);
        split2(x1.x, x1.y, qh[s].z, ql[s].z);
        split2(x1.z, x1.w, qh[s].w, ql[s].w);
    }

    f32x16 ot[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) ot[t][i] = 0.f;
    float m_run = kNegBig, l_run = 0.f;

    // staging: BYTE offsets of this thread's float4 chunks inside a tile (32-bit voffset of raw buffer loads; the tile start
    // is the scalar soffset).  K rows at or beyond Nk and the slots a thread does not own fall outside the descriptor's range
    // and read as zeros.
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)Kb, 0, (int)((((int64_t)p.Nk - 1) * p.ldk + D) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV =
        __builtin_amdgcn_make_buffer_rsrc((void*)Vb, 0, (int)((((int64_t)D - 1) * p.ldvt + (p.Nk + 3) / 4 * 4) * 4), 0x00020000);
    float4 kreg[NKC], vreg[NVC];
    int kofs[NKC], vofs[NVC];
#pragma unroll
    for (int u = 0; u < NKC; ++u) {
        const int id = tid + 256 * u, key = id / DC;
        kofs[u] = id < KV * DC ? (key * (int)p.ldk + (id - key * DC) * 4) * 4 : 0x7fffffff;
    }
#pragma unroll
    for (int u = 0; u < NVC; ++u) {
        const int id = tid + 256 * u;
        vofs[u] = id < D * 16 ? ((id >> 4) * (int)p.ldvt + (id & 15) * 4) * 4 : 0x7fffffff;
    }
    auto as_f4 = [](auto v) { return *reinterpret_cast<float4*>(&v); };
    auto load_kv = [&](int k0) {
        const int ksoff = k0 * (int)p.ldk * 4, vsoff = k0 * 4;
#pragma unroll
        for (int u = 0; u < NKC; ++u) kreg[u] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rsK, kofs[u], ksoff, 0));
#pragma unroll
        for (int u = 0; u < NVC; ++u) vreg[u] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rsV, vofs[u], vsoff, 0));
        if (k0 + KV <= p.Nk) return;  // full tile (wave-uniform)
        // last, partial tile: V^T columns of keys >= Nk (row padding, or the next row's keys) are staged as zeros; their scores
        // are masked as well
#pragma unroll
        for (int u = 0; u < NVC; ++u) {
            const int nv = p.Nk - (k0 + ((tid + 256 * u) & 15) * 4);  // valid elements in this chunk
            if (nv < 4) {
                if (nv < 1) vreg[u].x = 0.f;
                if (nv < 2) vreg[u].y = 0.f;
                if (nv < 3) vreg[u].z = 0.f;
                vreg[u].w = 0.f;
            }
        }
    };
    auto store_kv = [&](int st) {
        unsigned char* Kh = smem + st * kStageBytes;
        unsigned char* Vh = Kh + 2 * kKPlane;
#pragma unroll
        for (int u = 0; u < NKC; ++u) {
            const int id = tid + 256 * u;
            const int key = id / DC, ch = id - key * DC;
            if (id < KV * DC) {
                uint2 h, l;
                split2(kreg[u].x, kreg[u].y, h.x, l.x);
                split2(kreg[u].z, kreg[u].w, h.y, l.y);
                *reinterpret_cast<uint2*>(Kh + key * KROW + ch * 8) = h;
                *reinterpret_cast<uint2*>(Kh + kKPlane + key * KROW + ch * 8) = l;
            }
        }
#pragma unroll
        for (int u = 0; u < NVC; ++u) {
            const int id = tid + 256 * u;
            const int d = id >> 4, ch = id & 15;
            if (id < D * 16) {
                uint2 h, l;
                split2(vreg[u].x, vreg[u].y, h.x, l.x);
                split2(vreg[u].z, vreg[u].w, h.y, l.y);
                *reinterpret_cast<uint2*>(Vh + d * VROW + ch * 8) = h;
                *reinterpret_cast<uint2*>(Vh + kVPlane + d * VROW + ch * 8) = l;
            }
        }
    };

    const int ntiles = (p.Nk + KV - 1) / KV;
    const float c = p.scale_log2;

    auto tile = [&](const int kt, const int STAGE) {
        const int k0 = kt * KV;
        const unsigned char* Kh = smem + STAGE * kStageBytes;
        const unsigned char* Kl = Kh + kKPlane;
        const unsigned char* Vh = Kh + 2 * kKPlane;
        const unsigned char* Vl = Vh + kVPlane;
        if (NST == 2 && kt + 1 < ntiles) load_kv(k0 + KV);  // next tile into registers, written to the other stage after the MFMAs

        // ---- S^T = K Q^T for two 32-key tiles, three passes (small terms first) ----
        f32x16 st[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int i = 0; i < 16; ++i) st[t][i] = 0.f;
#pragma unroll
            for (int s = 0; s < DK; ++s) {
                const uint4 kh = *reinterpret_cast<const uint4*>(Kh + (32 * t + r) * KROW + (2 * s + hh) * 16);
                const uint4 kl = *reinterpret_cast<const uint4*>(Kl + (32 * t + r) * KROW + (2 * s + hh) * 16);
                st[t] = mfma_h(kl, qh[s], st[t]);
                st[t] = mfma_h(kh, ql[s], st[t]);
                st[t] = mfma_h(kh, qh[s], st[t]);
            }
        }
        if (k0 + KV > p.Nk) {  // mask keys beyond Nk (last tile only; wave-uniform branch)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = k0 + 32 * t + (i & 3) + 8 * (i >> 2) + 4 * hh;
                    if (key >= p.Nk) st[t][i] = kNegBig;
                }
        }
        // ---- online softmax (query = lane&31; this half-wave holds 32 of the 64 keys) ----
        float mx = fmaxf(st[0][0], st[1][0]);
#pragma unroll
        for (int i = 1; i < 16; ++i) mx = fmaxf(fmaxf(mx, st[0][i]), st[1][i]);
        mx = half_swap_max(mx);
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
        const float mc = m_new * c - kPShift;
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float pv = __builtin_amdgcn_exp2f(st[t][i] * c - mc);
                st[t][i] = pv;
                psum += pv;
            }
        l_run = l_run * alpha + psum;
        if (!__all(alpha == 1.0f)) {  // wave-uniform: no running maximum of this wave moved -> no rescale
#pragma unroll
            for (int t = 0; t < DT; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) ot[t][i] *= alpha;
        }
        // P^T -> float16 hi / lo B fragments: k-step ks = 2 t + s uses registers 8 s .. 8 s + 7 of tile t
        uint4 ph[4], pl[4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                uint4 h, l;
                split2(st[t][8 * s + 0], st[t][8 * s + 1], h.x, l.x);
                split2(st[t][8 * s + 2], st[t][8 * s + 3], h.y, l.y);
                split2(st[t][8 * s + 4], st[t][8 * s + 5], h.z, l.z);
                split2(st[t][8 * s + 6], st[t][8 * s + 7], h.w, l.w);
                ph[2 * t + s] = h;
                pl[2 * t + s] = l;
            }
        // ---- O^T += V^T P^T, three passes ----
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                // element j of lane half hh is key 16 ks + 8 (j>>2) + 4 hh + (j&3) of this 64-key tile
                const int off = (32 * dt + r) * VROW + (16 * ks + 4 * hh) * 2;
                const uint2 h0 = *reinterpret_cast<const uint2*>(Vh + off), h1 = *reinterpret_cast<const uint2*>(Vh + off + 16);
                const uint2 l0 = *reinterpret_cast<const uint2*>(Vl + off), l1 = *reinterpret_cast<const uint2*>(Vl + off + 16);
                const uint4 vh = make_uint4(h0.x, h0.y, h1.x, h1.y), vl = make_uint4(l0.x, l0.y, l1.x, l1.y);
                ot[dt] = mfma_h(vl, ph[ks], ot[dt]);
                ot[dt] = mfma_h(vh, pl[ks], ot[dt]);
                ot[dt] = mfma_h(vh, ph[ks], ot[dt]);
            }
        }
        if (NST == 2) {
            if (kt + 1 < ntiles) store_kv(STAGE ^ 1);  // that stage was last read in iteration kt-1
            __syncthreads();
        }
    };

    if constexpr (NST == 2) {
        load_kv(0);
        __syncthreads();  // padding fill visible before the first stage is written around it
        store_kv(0);
        __syncthreads();
        for (int kt = 0; kt < ntiles; ++kt) tile(kt, kt & 1);
    } else {
        for (int kt = 0; kt < ntiles; ++kt) {
            load_kv(kt * KV);
            __syncthreads();  // every wave is done reading the previous tile (first pass: padding fill complete)
            store_kv(0);
            __syncthreads();
            tile(kt, 0);
        }
        __syncthreads();  // the output strips below overwrite the stage
    }

    // ---- normalise and store: lane owns query q, registers hold d = 32 dt + 8 g + 4 hh + i ----
    const float inv = 1.0f / half_swap_sum(l_run);
    constexpr int SROW = D * 4 + 16;  // strip row stride in bytes
    static_assert(4 * 32 * SROW <= NST * kStageBytes, "output strips must fit in the stages");
    unsigned char* strip = smem + wid * (32 * SROW);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d0 = 32 * dt + 8 * g + 4 * hh;
            if (d0 < D)
                *reinterpret_cast<float4*>(strip + r * SROW + d0 * 4) =
                    make_float4(ot[dt][4 * g + 0] * inv, ot[dt][4 * g + 1] * inv, ot[dt][4 * g + 2] * inv, ot[dt][4 * g + 3] * inv);
        }
    __builtin_amdgcn_wave_barrier();  // same wave, in-order LDS: a compiler fence only
    constexpr int CH = D / 4;         // float4 chunks per query row
    const int q0w = qb * 128 + wid * 32;
    float* Ow = p.O + (int64_t)b * p.sO + (int64_t)q0w * p.ldo + (int64_t)head * D;
#pragma unroll
    for (int t = 0; t < (32 * CH + 63) / 64; ++t) {
        const int idx = lane + 64 * t;
        const int rr = idx / CH, cc = idx - rr * CH;
        if (rr < 32 && q0w + rr < p.Nq) {
            const float4 o4 = *reinterpret_cast<const float4*>(strip + rr * SROW + cc * 16);
            if (p.o_split) gmd_store_split4(p.O, (int64_t)(Ow - p.O) + (int64_t)rr * p.ldo + cc * 4, o4.x, o4.y, o4.z, o4.w);
            else *reinterpret_cast<float4*>(Ow + (int64_t)rr * p.ldo + cc * 4) = o4;
        }
    }
}

template <int D>
hipError_t launch_attn_split(const AttnSplitParams& p, int B, hipStream_t s) {
    constexpr int DK = (D + 15) / 16, DT = (D + 31) / 32;
    constexpr int NST = D <= 64 ? 2 : 1;
    constexpr size_t smem = (size_t)NST * (2 * KV * ((2 * DK + 1) * 16) + 2 * DT * 32 * VROW);
    if (smem > 64 * 1024) {
        const hipError_t e = opt_in_lds(reinterpret_cast<const void*>(&attn_split_kernel<D>), (int)smem);
        if (e != hipSuccess) return e;
    }
    const int nqb = (p.Nq + 127) / 128;
    attn_split_kernel<D><<<dim3(nqb * p.H * B), 256, smem, s>>>(p);
    return hipGetLastError();
}

}  // namespace

// called by gmd_attention (attention.hip) for dtype GMD_F32S; arguments validated there
int gmd_launch_attention_split(const void* Q, const void* K, const void* Vt, void* O, int B, int H, int D, int Nq, int Nk, int64_t ldq,
                               int64_t ldk, int64_t ldvt, int64_t ldo, int64_t sQ, int64_t sK, int64_t sVt, int64_t sO, float scale,
                               int o_split, hipStream_t stream) {
    AttnSplitParams p;
    p.o_split = o_split;
    p.Q = (const float*)Q; p.K = (const float*)K; p.Vt = (const float*)Vt; p.O = (float*)O;
    p.Nq = Nq; p.Nk = Nk; p.H = H;
    p.ldq = ldq; p.ldk = ldk; p.ldvt = ldvt; p.ldo = ldo; p.sQ = sQ; p.sK = sK; p.sVt = sVt; p.sO = sO;
    p.scale_log2 = scale * 1.4426950408889634f;
    hipError_t e;
    switch (D) {
        case 40: e = launch_attn_split<40>(p, B, stream); break;
        case 64: e = launch_attn_split<64>(p, B, stream); break;  // SDXL: every level has heads of 64
        case 80: e = launch_attn_split<80>(p, B, stream); break;
        case 160: e = launch_attn_split<160>(p, B, stream); break;
        default:
            gmd_set_error("gmd_attention: float32 (split) attention is instantiated for head dims 40, 64, 80, 160 (got %d)", D);
            return GMD_ERR_UNSUPPORTED;
    }
    if (e != hipSuccess) {
        gmd_set_error("gmd_attention: launch failed: %s", hipGetErrorString(e));
        return GMD_ERR_LAUNCH;
    }
    return GMD_OK;
}

// Flash-style attention for the UNet transformer blocks (self-attention over h*w latent tokens,
// cross-attention over 77 text tokens), bf16 in / fp32 accumulate on the matrix cores.
//
// Formulation (everything "transposed" so that the softmax state of a query lives on ONE lane):
//   S^T[key, q] = sum_d K[key, d] Q[q, d]      mfma_f32_32x32x16_bf16(A = K fragment, B = Q fragment)
//   O^T[d,  q] += sum_key V^T[d, key] P^T[key, q]   mfma(A = V^T fragment, B = P^T)
// The accumulator tile S^T (key in registers, query on the lane) is converted to bf16 in place and
// used directly as the B operand of the second product (no LDS round trip, no cross-lane traffic);
// the row max / row sum of a query are 32 in-register values per half-wave plus ONE __shfl_xor(.,32).
// O^T keeps d in registers and the query on the lane, so the online-softmax rescale is a per-lane
// multiply.  The k index of the second product is permuted by the C/D register layout
// (row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)); the V^T fragment reads apply the same permutation.
//
// One workgroup = 4 waves = 128 queries of one (batch, head); K tile [64 keys][d] and V^T tile
// [d][64 keys] are staged through registers into LDS with row strides chosen so that the
// ds_read_b128 (K) and ds_read_b64 (V^T) fragment reads are bank-conflict free.
#include "gmd_common.h"

namespace {

struct AttnParams {
    const bf16_t* Q;
    const bf16_t* K;
    const bf16_t* Vt;
    bf16_t* O;
    int Nq, Nk;
    int64_t ldq, ldk, ldvt, ldo, sQ, sK, sVt, sO;
    float scale_log2;  // softmax scale * log2(e)
};

constexpr int KV = 64;      // keys per tile
constexpr int VROW = 136;   // bytes per V^T LDS row: 64 keys * 2 B + 8 B pad (conflict-free b64 reads)
constexpr float kNegBig = -1.0e30f;

template <int D>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnParams p) {
    constexpr int DK = (D + 15) / 16;         // 16-wide k-steps of Q K^T over d
    constexpr int DT = (D + 31) / 32;         // 32-row tiles of O^T over d
    constexpr int KROW = (2 * DK + 1) * 16;   // bytes per K LDS row: odd number of 16-byte slots
    constexpr int DC = D / 8;                 // 16-byte chunks per K row in global memory
    static_assert(D % 8 == 0, "head dim must be a multiple of 8");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int kStageBytes = KV * KROW + DT * 32 * VROW;  // one pipeline stage: K tile [KV][KROW] + V^T tile [DT*32][VROW]
    constexpr int NKC = (KV * DC + 255) / 256;                // 16-byte K chunks staged per thread
    constexpr int NVC = (D * 8 + 255) / 256;                  // 16-byte V^T chunks staged per thread

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int q = blockIdx.x * 128 + wid * 32 + r;
    const bool qvalid = q < p.Nq;
    const bf16_t* Qb = p.Q + (int64_t)b * p.sQ + (int64_t)head * D;
    const bf16_t* Kb = p.K + (int64_t)b * p.sK + (int64_t)head * D;
    const bf16_t* Vb = p.Vt + (int64_t)b * p.sVt + (int64_t)head * D * p.ldvt;

    // zero the LDS padding that is read but never staged (both stages): K columns [D, DK*16), V^T rows [D, DT*32)
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        unsigned char* Ks0 = smem + st * kStageBytes;
        unsigned char* Vs0 = Ks0 + KV * KROW;
        if (DK * 16 > D) {
            for (int i = tid; i < KV; i += 256) *reinterpret_cast<uint4*>(Ks0 + i * KROW + DC * 16) = make_uint4(0, 0, 0, 0);
        }
        for (int i = tid; i < (DT * 32 - D) * (VROW / 8); i += 256) {
            const int row = D + i / (VROW / 8), c = i % (VROW / 8);
            *reinterpret_cast<uint2*>(Vs0 + row * VROW + c * 8) = make_uint2(0, 0);
        }
    }

    // Q fragments (B operand): lane (r, hh) element j = Q[q][16 s + 8 hh + j]
    bf16x8 qf[DK];
#pragma unroll
    for (int s = 0; s < DK; ++s) {
        const int d0 = 16 * s + 8 * hh;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (qvalid && d0 + 8 <= D) v = *reinterpret_cast<const uint4*>(Qb + (int64_t)q * p.ldq + d0);
        qf[s] = as_frag(v);
    }

    f32x16 ot[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) ot[t][i] = 0.f;
    float m_run = kNegBig, l_run = 0.f;
    const float c = p.scale_log2;

    // register staging of the NEXT tile: issued before the MFMAs of the current tile, written to the other LDS
    // stage after them (one barrier per tile)
    uint4 kreg[NKC], vreg[NVC];
    auto load_kv = [&](int k0) {
#pragma unroll
        for (int u = 0; u < NKC; ++u) {
            const int id = tid + 256 * u;
            const int key = id / DC, ch = id - key * DC;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (id < KV * DC && k0 + key < p.Nk) v = *reinterpret_cast<const uint4*>(Kb + (int64_t)(k0 + key) * p.ldk + ch * 8);
            kreg[u] = v;
        }
#pragma unroll
        for (int u = 0; u < NVC; ++u) {
            const int id = tid + 256 * u;
            const int d = id >> 3, ch = id & 7;
            const int key = k0 + ch * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (id < D * 8 && key < p.Nk) {
                v = *reinterpret_cast<const uint4*>(Vb + (int64_t)d * p.ldvt + key);
                const int nv = p.Nk - key;  // valid elements in this chunk (>= 1); keys >= Nk are zeroed
                if (nv < 8) {
                    unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (2 * e >= nv) w[e] = 0;
                        else if (2 * e + 1 >= nv) w[e] &= 0xffffu;
                    }
                    v = make_uint4(w[0], w[1], w[2], w[3]);
                }
            }
            vreg[u] = v;
        }
    };
    auto store_kv = [&](int st) {
        unsigned char* Ks0 = smem + st * kStageBytes;
        unsigned char* Vs0 = Ks0 + KV * KROW;
#pragma unroll
        for (int u = 0; u < NKC; ++u) {
            const int id = tid + 256 * u;
            const int key = id / DC, ch = id - key * DC;
            if (id < KV * DC) *reinterpret_cast<uint4*>(Ks0 + key * KROW + ch * 16) = kreg[u];
        }
#pragma unroll
        for (int u = 0; u < NVC; ++u) {
            const int id = tid + 256 * u;
            const int d = id >> 3, ch = id & 7;
            if (id < D * 8) {
                unsigned char* dst = Vs0 + d * VROW + ch * 16;
                *reinterpret_cast<uint2*>(dst) = make_uint2(vreg[u].x, vreg[u].y);
                *reinterpret_cast<uint2*>(dst + 8) = make_uint2(vreg[u].z, vreg[u].w);
            }
        }
    };

    const int ntiles = (p.Nk + KV - 1) / KV;
    load_kv(0);
    __syncthreads();  // padding zero-fill visible before the first stage is written around it
    store_kv(0);
    __syncthreads();
    for (int kt = 0; kt < ntiles; ++kt) {
        const int k0 = kt * KV;
        const unsigned char* Ks = smem + (kt & 1) * kStageBytes;
        const unsigned char* Vs = Ks + KV * KROW;
        if (kt + 1 < ntiles) load_kv(k0 + KV);

        // ---- S^T = K Q^T for two 32-key tiles ----
        f32x16 st[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int i = 0; i < 16; ++i) st[t][i] = 0.f;
#pragma unroll
            for (int s = 0; s < DK; ++s) {
                const bf16x8 kf = as_frag(*reinterpret_cast<const uint4*>(Ks + (32 * t + r) * KROW + (2 * s + hh) * 16));
                st[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], st[t], 0, 0, 0);
            }
        }
        // mask keys beyond Nk (last tile only; wave-uniform branch)
        if (k0 + KV > p.Nk) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = k0 + 32 * t + (i & 3) + 8 * (i >> 2) + 4 * hh;
                    if (key >= p.Nk) st[t][i] = kNegBig;
                }
        }
        // ---- online softmax (query = lane&31; this half-wave holds 32 of the 64 keys) ----
        float mx = st[0][0];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) mx = fmaxf(mx, st[t][i]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
        const float mc = m_new * c;
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
        if (!__all(alpha == 1.0f)) {  // wave-uniform: the running max of every query of this wave is unchanged -> no rescale
#pragma unroll
            for (int t = 0; t < DT; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) ot[t][i] *= alpha;
        }
        // P^T -> bf16 B fragments: k-step ks = 2 t + s uses registers 8 s .. 8 s + 7 of tile t
        bf16x8 pf[4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                uint4 w;
                w.x = pack_bf16x2(st[t][8 * s + 0], st[t][8 * s + 1]);
                w.y = pack_bf16x2(st[t][8 * s + 2], st[t][8 * s + 3]);
                w.z = pack_bf16x2(st[t][8 * s + 4], st[t][8 * s + 5]);
                w.w = pack_bf16x2(st[t][8 * s + 6], st[t][8 * s + 7]);
                pf[2 * t + s] = as_frag(w);
            }
        // ---- O^T += V^T P^T ----
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                // element j of lane half hh is key 16 ks + 8 (j>>2) + 4 hh + (j&3) of this 64-key tile
                const unsigned char* src = Vs + (32 * dt + r) * VROW + (16 * ks + 4 * hh) * 2;
                const uint2 lo = *reinterpret_cast<const uint2*>(src);
                const uint2 hi = *reinterpret_cast<const uint2*>(src + 16);
                const bf16x8 vf = as_frag(make_uint4(lo.x, lo.y, hi.x, hi.y));
                ot[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[ks], ot[dt], 0, 0, 0);
            }
        }
        if (kt + 1 < ntiles) store_kv((kt + 1) & 1);  // that stage was last read in iteration kt-1
        __syncthreads();
    }

    // ---- normalise and store: lane owns query q, registers hold d = 32 dt + 8 g + 4 hh + i ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (qvalid) {
        bf16_t* Ob = p.O + (int64_t)b * p.sO + (int64_t)q * p.ldo + (int64_t)head * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = 32 * dt + 8 * g + 4 * hh;
                if (d0 < D) {
                    uint2 w;
                    w.x = pack_bf16x2(ot[dt][4 * g + 0] * inv, ot[dt][4 * g + 1] * inv);
                    w.y = pack_bf16x2(ot[dt][4 * g + 2] * inv, ot[dt][4 * g + 3] * inv);
                    *reinterpret_cast<uint2*>(Ob + d0) = w;
                }
            }
    }
}

template <int D>
int launch_attn(const AttnParams& p, int B, int H, hipStream_t s) {
    constexpr int DK = (D + 15) / 16, DT = (D + 31) / 32;
    const size_t smem = 2 * ((size_t)KV * (2 * DK + 1) * 16 + (size_t)DT * 32 * VROW);  // two pipeline stages
    dim3 grid((p.Nq + 127) / 128, H, B);
    attn_fwd_kernel<D><<<grid, 256, smem, s>>>(p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        gmd_set_error("gmd_attention: launch failed: %s", hipGetErrorString(e));
        return GMD_ERR_LAUNCH;
    }
    return GMD_OK;
}

}  // namespace

extern "C" int gmd_attention(const void* Q, const void* K, const void* Vt, void* O, int dtype, int B, int H, int D, int Nq,
                             int Nk, int64_t ldq, int64_t ldk, int64_t ldvt, int64_t ldo, int64_t strideQ, int64_t strideK,
                             int64_t strideVt, int64_t strideO, float scale, gmd_stream_t stream) {
    if (dtype != GMD_BF16) {
        gmd_set_error("gmd_attention: only GMD_BF16 is implemented (the F32 parity path composes gmd_gemm_nt + gmd_softmax_rows)");
        return GMD_ERR_UNSUPPORTED;
    }
    GMD_REQUIRE(B > 0 && H > 0 && Nq > 0 && Nk > 0, "gmd_attention: bad shape B=%d H=%d Nq=%d Nk=%d", B, H, Nq, Nk);
    GMD_REQUIRE(B <= 65535 && H <= 65535, "gmd_attention: grid too large");
    GMD_REQUIRE(Q && K && Vt && O, "gmd_attention: null pointer");
    GMD_REQUIRE(gmd_aligned16(Q) && gmd_aligned16(K) && gmd_aligned16(Vt) && gmd_aligned16(O), "gmd_attention: pointers must be 16-byte aligned");
    GMD_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldvt % 8 == 0 && ldo % 4 == 0, "gmd_attention: leading dimensions must be multiples of 8 (ldo: 4)");
    GMD_REQUIRE(strideQ % 8 == 0 && strideK % 8 == 0 && strideVt % 8 == 0 && strideO % 4 == 0, "gmd_attention: batch strides must be multiples of 8");
    GMD_REQUIRE(ldvt >= ((Nk + 7) / 8) * 8, "gmd_attention: ldvt=%lld must cover Nk=%d rounded up to 8", (long long)ldvt, Nk);
    GMD_REQUIRE(ldq >= (int64_t)H * D && ldk >= (int64_t)H * D && ldo >= (int64_t)H * D, "gmd_attention: row stride smaller than H*D");
    AttnParams p;
    p.Q = (const bf16_t*)Q; p.K = (const bf16_t*)K; p.Vt = (const bf16_t*)Vt; p.O = (bf16_t*)O;
    p.Nq = Nq; p.Nk = Nk; p.ldq = ldq; p.ldk = ldk; p.ldvt = ldvt; p.ldo = ldo;
    p.sQ = strideQ; p.sK = strideK; p.sVt = strideVt; p.sO = strideO;
    p.scale_log2 = scale * 1.4426950408889634f;
    hipStream_t s = (hipStream_t)stream;
    switch (D) {
        case 32: return launch_attn<32>(p, B, H, s);
        case 40: return launch_attn<40>(p, B, H, s);
        case 64: return launch_attn<64>(p, B, H, s);
        case 80: return launch_attn<80>(p, B, H, s);
        case 160: return launch_attn<160>(p, B, H, s);
        default:
            gmd_set_error("gmd_attention: head dim %d not instantiated (supported: 32, 40, 64, 80, 160)", D);
            return GMD_ERR_UNSUPPORTED;
    }
}

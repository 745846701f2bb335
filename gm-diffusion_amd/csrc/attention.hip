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

#include <type_traits>

// attention_split.hip: float32 tensors, three float16 MFMA passes per contraction (GMD_F32S)
int gmd_launch_attention_split(const void* Q, const void* K, const void* Vt, void* O, int B, int H, int D, int Nq, int Nk, int64_t ldq,
                               int64_t ldk, int64_t ldvt, int64_t ldo, int64_t sQ, int64_t sK, int64_t sVt, int64_t sO, float scale, int o_split,
                               hipStream_t stream);

namespace {

struct AttnParams {
    const bf16_t* Q;
    const bf16_t* K;
    const bf16_t* Vt;
    bf16_t* O;
    int Nq, Nk, H;
    int64_t ldq, ldk, ldvt, ldo, sQ, sK, sVt, sO;
    float scale_log2;  // softmax scale * log2(e)
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int KV = 64;      // keys per tile
constexpr int VROW = 136;   // bytes per V^T LDS row: 64 keys * 2 B + 8 B pad (conflict-free b64 reads)
constexpr float kNegBig = -1.0e30f;

template <typename HT, int D, bool CAUSAL>  // HT: bf16_t or f16_t (storage pointers stay raw 16-bit)
// d <= 40: four waves per SIMD (128 VGPRs), d <= 64: three (<= 168) -- the softmax VALU of one wave hides under the MFMAs of the others
__global__ __launch_bounds__(256, D <= 40 ? 4 : (D <= 64 ? 3 : 1)) void attn_fwd_kernel(const AttnParams p) {
    GMD_WG_TRACE_SCOPE(WGK_ATTN);
    constexpr int DK = (D + 15) / 16;         // 16-wide k-steps of Q K^T over d
    constexpr int DT = (D + 31) / 32;         // 32-row tiles of O^T over d
    constexpr int KROW = (2 * DK + 1) * 16;   // bytes per K LDS row: odd number of 16-byte slots
    constexpr int DC = D / 8;                 // 16-byte chunks per K row in global memory
    static_assert(D % 8 == 0, "head dim must be a multiple of 8");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int kStageBytes = KV * KROW + DT * 32 * VROW;  // one pipeline stage: K tile [KV][KROW] + V^T tile [DT*32][VROW]
    constexpr int NKC = (KV * DC + 255) / 256;                // 16-byte K chunks staged per thread
    constexpr int NVC = (D * 8 + 255) / 256;                  // 16-byte V^T chunks staged per thread
    // A spare V^T row (d padded to a multiple of 32) is filled with ones: O^T[D][q] then accumulates the softmax row sum
    // on the matrix core (from the same bf16 P the numerator uses) and the 32 VALU adds per tile disappear.
    constexpr bool kRowSumMfma = DT * 32 > D;
    // A spare k column of the first product (d padded to a multiple of 16) carries the softmax stabiliser through the
    // matrix core: K[key][D] = 1 and Q[q][D] = -m[q], with Q pre-multiplied by scale*log2(e), so the accumulator IS
    // log2(p) relative to the running stabiliser m of the PREVIOUS tile and exp2 applies to it directly (no per-score
    // v_fma).  m moves only when a tile's maximum exceeds it; the output is rescaled after the second product.  A tile
    // whose scores exceed the lagged stabiliser by more than 2^kLagMax (and always the first tile) takes the classic
    // path: subtract the tile's own maximum before exp2.  m is kept bf16-representable so the MFMA subtracts exactly
    // what the rescale assumes.
    constexpr bool kLagged = kRowSumMfma && DK * 16 > D;
    // (P = 2^(score - stabiliser) is converted to the 16-bit type for the second product: 2^20 fits bfloat16, float16 tops
    // out at 65504, so its window is 2^14)
    constexpr float kLagMax = std::is_same<HT, f16_t>::value ? 14.0f : 20.0f;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs (private L2 each), so workgroup ids with equal
    // id % 8 get a CONTIGUOUS run of (batch, head, query block) triples (bijective remap, as in the GEMM): the query blocks
    // that stream the same K / V^T of one (batch, head) then share one L2 instead of fetching it through all eight.
    int qb, head, b;
    {
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
    const bf16_t* Qb = p.Q + (int64_t)b * p.sQ + (int64_t)head * D;
    const bf16_t* Kb = p.K + (int64_t)b * p.sK + (int64_t)head * D;
    const bf16_t* Vb = p.Vt + (int64_t)b * p.sVt + (int64_t)head * D * p.ldvt;

    // zero the LDS padding that is read but never staged (both stages): K columns [D, DK*16), V^T rows [D, DT*32)
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        unsigned char* Ks0 = smem + st * kStageBytes;
        unsigned char* Vs0 = Ks0 + KV * KROW;
        if (DK * 16 > D) {
            for (int i = tid; i < KV; i += 256)
                *reinterpret_cast<uint4*>(Ks0 + i * KROW + DC * 16) = make_uint4(kLagged ? Half<HT>::kOne : 0u, 0, 0, 0);  // K[key][D] = 1.0
        }
        for (int i = tid; i < (DT * 32 - D) * (VROW / 8); i += 256) {
            const int row = D + i / (VROW / 8), c = i % (VROW / 8);
            const unsigned fill = (kRowSumMfma && row == D) ? (Half<HT>::kOne | (Half<HT>::kOne << 16)) : 0u;  // 1.0 pairs in the row-sum row
            *reinterpret_cast<uint2*>(Vs0 + row * VROW + c * 8) = make_uint2(fill, fill);
        }
    }

    // Q fragments (B operand): lane (r, hh) element j = Q[q][16 s + 8 hh + j]
    uint4 qf[DK];  // raw 16-bit fragments
#pragma unroll
    for (int s = 0; s < DK; ++s) {
        const int d0 = 16 * s + 8 * hh;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (qvalid && d0 + 8 <= D) v = *reinterpret_cast<const uint4*>(Qb + (int64_t)q * p.ldq + d0);
        if constexpr (kLagged) {
            unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float lo, hi;
                Half<HT>::unpack2(w[j], lo, hi);
                w[j] = Half<HT>::pack2(lo * p.scale_log2, hi * p.scale_log2);
            }
            v = make_uint4(w[0], w[1], w[2], w[3]);
        }
        qf[s] = v;
    }

    f32x16 ot[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) ot[t][i] = 0.f;
    float m_run = kLagged ? 0.f : kNegBig, l_run = 0.f;
    [[maybe_unused]] int lag_overflow = 0;
    [[maybe_unused]] auto set_stabiliser = [&](float m_new) {  // m_new must be representable in the 16-bit type
        constexpr int SP = D / 16, HP = (D % 16) / 8;          // fragment / lane half holding column D of Q
        m_run = m_new;
        const unsigned nm = Half<HT>::pack2(-m_new, 0.0f) & 0xffffu;  // element 0 of the fragment = low half of word 0
        uint4& f = qf[SP < DK ? SP : 0];
        f.x = hh == HP ? ((f.x & 0xffff0000u) | nm) : f.x;
    };

    // register staging of the NEXT tile: issued before the MFMAs of the current tile, written to the other LDS
    // stage after them (one barrier per tile)
    uint4 kreg[NKC], vreg[NVC];
    // per-thread staging slots are tile-invariant: BYTE offsets of this thread's K / V^T chunks inside a tile, used as the
    // 32-bit voffset of raw buffer loads (the tile start is the scalar soffset: no 64-bit per-lane addresses).  Rows of K
    // at or beyond Nk and the slots a thread does not own fall outside the descriptor's range and read as zeros.
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(
        (void*)Kb, 0, (int)((((int64_t)p.Nk - 1) * p.ldk + D) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(
        (void*)Vb, 0, (int)((((int64_t)D - 1) * p.ldvt + (p.Nk + 7) / 8 * 8) * 2), 0x00020000);
    int kofs[NKC], vofs[NVC];
#pragma unroll
    for (int u = 0; u < NKC; ++u) {
        const int id = tid + 256 * u, key = id / DC;
        kofs[u] = id < KV * DC ? (key * (int)p.ldk + (id - key * DC) * 8) * 2 : 0x7fffffff;
    }
#pragma unroll
    for (int u = 0; u < NVC; ++u) {
        const int id = tid + 256 * u;
        vofs[u] = id < D * 8 ? ((id >> 3) * (int)p.ldvt + (id & 7) * 8) * 2 : 0x7fffffff;
    }
    auto as_u4 = [](auto v) { return *reinterpret_cast<uint4*>(&v); };
    auto load_kv = [&](int k0) {
        const int ksoff = k0 * (int)p.ldk * 2, vsoff = k0 * 2;
#pragma unroll
        for (int u = 0; u < NKC; ++u) kreg[u] = as_u4(__builtin_amdgcn_raw_buffer_load_b128(rsK, kofs[u], ksoff, 0));
#pragma unroll
        for (int u = 0; u < NVC; ++u) vreg[u] = as_u4(__builtin_amdgcn_raw_buffer_load_b128(rsV, vofs[u], vsoff, 0));
        if (k0 + KV <= p.Nk) return;  // full tile (wave-uniform)
        // last, partial tile: V^T columns of keys >= Nk (row padding, or the next row's keys) are staged as zeros; their
        // scores are masked to -inf as well
#pragma unroll
        for (int u = 0; u < NVC; ++u) {
            const int nv = p.Nk - (k0 + ((tid + 256 * u) & 7) * 8);  // valid elements in this chunk
            if (nv < 8) {
                unsigned w[4] = {vreg[u].x, vreg[u].y, vreg[u].z, vreg[u].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (2 * e >= nv) w[e] = 0;
                    else if (2 * e + 1 >= nv) w[e] &= 0xffffu;
                }
                vreg[u] = make_uint4(w[0], w[1], w[2], w[3]);
            }
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

    // CAUSAL (query q attends keys 0..q, Nq == Nk): key tiles entirely above the diagonal of this block's last query are
    // never visited (block-uniform bound)
    const int ntiles = CAUSAL ? min((p.Nk + KV - 1) / KV, (min(qb * 128 + 127, p.Nq - 1)) / KV + 1) : (p.Nk + KV - 1) / KV;
    // softmax scale inside the loop: the lagged variant folded it into Q
    const float c = kLagged ? 1.0f : p.scale_log2;

    // one K/V tile.  STAGE is a compile-time constant when the tile loop is unrolled by two (every LDS address of the
    // fragment reads is then a per-lane base plus an immediate); LAG selects the lagged-stabiliser softmax.
    auto tile = [&](const int kt, auto stage_c, auto lag_c) {
        constexpr bool LAG = decltype(lag_c)::value;
        const int STAGE = stage_c;
        const int k0 = kt * KV;
        const unsigned char* Ks = smem + STAGE * kStageBytes;
        const unsigned char* Vs = Ks + KV * KROW;
        // prefetch of the next tile into registers: at the top for the classic loop; the lagged loop (128-VGPR budget) issues
        // it after the softmax, where the score registers are dead, and lets the second product hide the latency
        if constexpr (!LAG) {
            if (kt + 1 < ntiles) load_kv(k0 + KV);
        }

        // ---- S^T = K Q^T for two 32-key tiles ----
        f32x16 st[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int i = 0; i < 16; ++i) st[t][i] = 0.f;
#pragma unroll
            for (int s = 0; s < DK; ++s) {
                const uint4 kf = *reinterpret_cast<const uint4*>(Ks + (32 * t + r) * KROW + (2 * s + hh) * 16);
                st[t] = Half<HT>::mfma32(kf, qf[s], st[t]);
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
        if (CAUSAL && k0 + KV - 1 > qb * 128 + wid * 32) {  // the tile reaches above this wave's first query
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = k0 + 32 * t + (i & 3) + 8 * (i >> 2) + 4 * hh;
                    if (key > q) st[t][i] = kNegBig;
                }
        }
        // ---- online softmax (query = lane&31; this half-wave holds 32 of the 64 keys) ----
        float mx = fmaxf(st[0][0], st[1][0]);
#pragma unroll
        for (int i = 1; i < 16; ++i) mx = fmaxf(fmaxf(mx, st[0][i]), st[1][i]);  // v_max3_f32
        mx = half_swap_max(mx);
        float alpha;
        if constexpr (LAG) {
            // st is already log2(p) against the stabiliser of the previous tile: exp2 applies directly.  The stabiliser
            // then follows the running maximum; this tile was accumulated against the old one, so the output is rescaled
            // AFTER the second product.
            lag_overflow |= mx > kLagMax;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) st[t][i] = __builtin_amdgcn_exp2f(st[t][i]);
            const float m_new = Half<HT>::round(m_run + fmaxf(mx, 0.f));
            alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            set_stabiliser(m_new);
        } else {
            const float m_new = fmaxf(m_run, mx);
            alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
            const float mc = m_new * c;
            m_run = m_new;
            float psum = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float pv = __builtin_amdgcn_exp2f(st[t][i] * c - mc);
                    st[t][i] = pv;
                    if constexpr (!kRowSumMfma) psum += pv;
                }
            if constexpr (!kRowSumMfma) l_run = l_run * alpha + psum;
            if (!__all(alpha == 1.0f)) {  // wave-uniform: no running max of this wave moved -> no rescale
#pragma unroll
                for (int t = 0; t < DT; ++t)
#pragma unroll
                    for (int i = 0; i < 16; ++i) ot[t][i] *= alpha;
            }
        }
        // P^T -> bf16 B fragments: k-step ks = 2 t + s uses registers 8 s .. 8 s + 7 of tile t
        uint4 pf[4];
        if constexpr (LAG) {
            if (kt + 1 < ntiles) load_kv(k0 + KV);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                uint4 w;
                w.x = Half<HT>::pack2(st[t][8 * s + 0], st[t][8 * s + 1]);
                w.y = Half<HT>::pack2(st[t][8 * s + 2], st[t][8 * s + 3]);
                w.z = Half<HT>::pack2(st[t][8 * s + 4], st[t][8 * s + 5]);
                w.w = Half<HT>::pack2(st[t][8 * s + 6], st[t][8 * s + 7]);
                pf[2 * t + s] = w;
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
                const uint4 vf = make_uint4(lo.x, lo.y, hi.x, hi.y);
                ot[dt] = Half<HT>::mfma32(vf, pf[ks], ot[dt]);
            }
        }
        if constexpr (LAG) {
            if (!__all(alpha == 1.0f)) {
#pragma unroll
                for (int t = 0; t < DT; ++t)
#pragma unroll
                    for (int i = 0; i < 16; ++i) ot[t][i] *= alpha;
            }
        }
        if (kt + 1 < ntiles) store_kv(STAGE ^ 1);  // that stage was last read in iteration kt-1
        __syncthreads();
    };

    load_kv(0);
    __syncthreads();  // padding fill visible before the first stage is written around it
    store_kv(0);
    __syncthreads();
    if constexpr (kLagged) {
        // initial stabiliser: the maximum of the first tile's scores (one extra first product, 1/ntiles of the work)
        {
            float mx = kNegBig;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x16 s0;
#pragma unroll
                for (int i = 0; i < 16; ++i) s0[i] = 0.f;
#pragma unroll
                for (int s = 0; s < DK; ++s) {
                    const uint4 kf = *reinterpret_cast<const uint4*>(smem + (32 * t + r) * KROW + (2 * s + hh) * 16);
                    s0 = Half<HT>::mfma32(kf, qf[s], s0);
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = 32 * t + (i & 3) + 8 * (i >> 2) + 4 * hh;
                    if (key < p.Nk && !(CAUSAL && key > q)) mx = fmaxf(mx, s0[i]);
                }
            }
            mx = half_swap_max(mx);
            set_stabiliser(Half<HT>::round(mx));
        }
        for (int kt = 0; kt < ntiles; ++kt) tile(kt, kt & 1, std::true_type{});
        // A score more than 2^kLagMax above the lagged stabiliser could overflow exp2: redo the whole block with the
        // classic (maximum-first) softmax.  Block-wide decision -- every wave takes part in the staging barriers.
        if (__syncthreads_or(lag_overflow)) {
#pragma unroll
            for (int t = 0; t < DT; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) ot[t][i] = 0.f;
            set_stabiliser(0.f);
            m_run = kNegBig;
            load_kv(0);
            store_kv(0);
            __syncthreads();
            for (int kt = 0; kt < ntiles; ++kt) tile(kt, kt & 1, std::false_type{});
        }
    } else {
        for (int kt = 0; kt < ntiles; kt += 2) {
            tile(kt, std::integral_constant<int, 0>{}, std::false_type{});
            if (kt + 1 < ntiles) tile(kt + 1, std::integral_constant<int, 1>{}, std::false_type{});
        }
    }

    // ---- normalise and store: lane owns query q, registers hold d = 32 dt + 8 g + 4 hh + i ----
    float l_tot;
    if constexpr (kRowSumMfma) {
        // row D of O^T: tile D/32, register 4*((D%32)/8) of the lanes with hh == 0 (D is a multiple of 8)
        l_tot = __shfl(ot[D / 32][4 * ((D % 32) / 8)], r, 64);
    } else {
        l_tot = half_swap_sum(l_run);
    }
    const float inv = 1.0f / l_tot;
    // Row-contiguous output through LDS (the stages are dead after the last tile's barrier).  Out of the registers a store
    // instruction would cover 32 rows x 16 bytes, and the store path is transaction-bound (see gemm.hip, epilogue_rows);
    // instead every wave packs its 32 x D tile into a private strip and writes whole D*2-byte head segments, 16 bytes a lane.
    constexpr int SROW = D * 2 + 16;                 // strip row stride in bytes (pad: spreads the rows over the banks)
    unsigned char* strip = smem + wid * (32 * SROW);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d0 = 32 * dt + 8 * g + 4 * hh;
            if (d0 < D) {
                uint2 w;
                w.x = Half<HT>::pack2(ot[dt][4 * g + 0] * inv, ot[dt][4 * g + 1] * inv);
                w.y = Half<HT>::pack2(ot[dt][4 * g + 2] * inv, ot[dt][4 * g + 3] * inv);
                *reinterpret_cast<uint2*>(strip + r * SROW + d0 * 2) = w;
            }
        }
    __builtin_amdgcn_wave_barrier();  // same wave, in-order LDS: a compiler fence only
    constexpr int CH = D / 8;         // 16-byte chunks per query row
    const int q0w = qb * 128 + wid * 32;
    bf16_t* Ow = p.O + (int64_t)b * p.sO + (int64_t)q0w * p.ldo + (int64_t)head * D;
#pragma unroll
    for (int t = 0; t < (32 * CH + 63) / 64; ++t) {
        const int idx = lane + 64 * t;
        const int rr = idx / CH, cc = idx - rr * CH;
        if (rr < 32 && q0w + rr < p.Nq)
            *reinterpret_cast<uint4*>(Ow + (int64_t)rr * p.ldo + cc * 8) = *reinterpret_cast<const uint4*>(strip + rr * SROW + cc * 16);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Head dim 40 (the 64x64-level self-attention of SD-1.5 -- three quarters of all attention time -- and its 77-key
// cross-attention): the same transposed formulation, with the K / V^T tiles brought in by LDS-DMA instead of through
// registers.  Measured on the kernel above (B=8, N=4096): with the staging of tiles 1.. removed the launch takes 236 us
// instead of 303 us -- the register prefetch is issued after the softmax (the 128-VGPR budget has no room earlier) and
// consumed at the end of the same tile, which exposes most of an L2 round trip per tile.  Here a tile is requested TWO
// tiles ahead into a 3-stage LDS ring (`buffer_load_dwordx4 ... lds`, counted vmcnt, one barrier per tile) and costs no
// registers at all.  What the lane-linear DMA image dictates:
//   * K rows are dense (80 B = 5 slots of 16 B: an odd stride, so the ds_read_b128 fragment reads stay conflict-free);
//     the stabiliser column K[key][40] = 1 of the lagged softmax cannot live in the rows any more -- the lanes that would
//     read d = 40..47 read one constant 16-byte vector {1, 0 x 7} of the stage instead (same address: a broadcast).
//   * V^T rows are 128 B (64 keys) with the 16-byte chunk index XOR (row >> 1) & 7, applied to the per-lane SOURCE address
//     of the DMA and to the reads.  Each lane fetches the 8 keys of a k-step as ONE ds_read_b128 (two ds_read_b64 before):
//     lane r of the first product reads K row (r with bits 2 and 3 swapped), which makes the 8 values a lane holds for a
//     k-step of the second product 8 CONSECUTIVE keys (16 ks + 8 (lane>>5) + j) instead of two groups of four.
//   * the padding rows of V^T (row 40 = ones: the softmax row sum on the matrix core; rows 41.. = zeros) are two constant
//     rows per stage that the lanes of d >= 40 address directly.
// Keys at and beyond Nk in the last tile: K rows come back as zeros (the tile offset moves into the range-checked per-lane
// offset for that tile) and are masked after the first product; V^T columns are zeroed in LDS before use.
// ------------------------------------------------------------------------------------------------------------------
template <int N> __device__ __forceinline__ void attn_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// (experiment hooks, compiled to nothing in the product: static priority of a wave's MFMA / softmax segments, tools/dbg)
#if defined(GMD_ATTN40_PRIO) && GMD_ATTN40_PRIO == 1
#define ATTN40_PRIO_MFMA(x) __builtin_amdgcn_s_setprio(x);
#define ATTN40_PRIO_VALU(x)
#elif defined(GMD_ATTN40_PRIO) && GMD_ATTN40_PRIO == 2
#define ATTN40_PRIO_MFMA(x)
#define ATTN40_PRIO_VALU(x) __builtin_amdgcn_s_setprio(x);
#else
#define ATTN40_PRIO_MFMA(x)
#define ATTN40_PRIO_VALU(x)
#endif
template <typename HT>
__global__ __launch_bounds__(256, 4) void attn40_kernel(const AttnParams p) {
    GMD_WG_TRACE_SCOPE(WGK_ATTN40);
    constexpr int D = 40, DK = 3, DT = 2, NST = 3;
    constexpr int KROWB = 2 * D;                 // dense K row, bytes
    constexpr int kLagOff = KV * KROWB;          // {1.0, 0 x 7}
    constexpr int kVOff = 5376;                  // V^T rows (128 B each), 256-byte aligned
    constexpr int kStage = kVOff + (D + 2) * 128;  // + the ones row and the zero row
    static_assert(kLagOff + 16 <= kVOff && kStage % 256 == 0, "stage layout");
    constexpr float kLagMax = std::is_same<HT, f16_t>::value ? 14.0f : 20.0f;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wuni = __builtin_amdgcn_readfirstlane(wid);
    const int r = lane & 31, hh = lane >> 5;
    int qb, head, b;
    {
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
    const bf16_t* Qb = p.Q + (int64_t)b * p.sQ + (int64_t)head * D;
    const bf16_t* Kb = p.K + (int64_t)b * p.sK + (int64_t)head * D;
    const bf16_t* Vb = p.Vt + (int64_t)b * p.sVt + (int64_t)head * D * p.ldvt;

    // constant parts of the three stages
    for (int i = tid; i < NST * 24; i += 256) {
        const int st = i / 24, e = i - st * 24;
        unsigned char* base = smem + st * kStage;
        if (e < 8) {
            const unsigned one2 = Half<HT>::kOne | (Half<HT>::kOne << 16);
            *reinterpret_cast<uint4*>(base + kVOff + D * 128 + e * 16) = make_uint4(one2, one2, one2, one2);
        } else if (e < 16) {
            *reinterpret_cast<uint4*>(base + kVOff + (D + 1) * 128 + (e - 8) * 16) = make_uint4(0, 0, 0, 0);
        } else if (e == 16) {
            *reinterpret_cast<uint4*>(base + kLagOff) = make_uint4(Half<HT>::kOne, 0, 0, 0);
        }
    }

    // Q fragments (B operand of the first product), pre-multiplied by scale * log2(e)
    uint4 qf[DK];
#pragma unroll
    for (int s = 0; s < DK; ++s) {
        const int d0 = 16 * s + 8 * hh;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (qvalid && d0 + 8 <= D) v = *reinterpret_cast<const uint4*>(Qb + (int64_t)q * p.ldq + d0);
        unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float lo, hi;
            Half<HT>::unpack2(w[j], lo, hi);
            w[j] = Half<HT>::pack2(lo * p.scale_log2, hi * p.scale_log2);
        }
        qf[s] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    f32x16 ot[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) ot[t][i] = 0.f;
    float m_run = 0.f;
    int lag_overflow = 0;
    auto set_stabiliser = [&](float m_new) {  // Q[q][40] = -m: fragment 2, lanes of the upper half, element 0
        m_run = m_new;
        const unsigned nm = Half<HT>::pack2(-m_new, 0.0f) & 0xffffu;
        qf[2].x = hh == 1 ? ((qf[2].x & 0xffff0000u) | nm) : qf[2].x;
    };

    // ---- per-lane LDS read offsets inside a stage (tile-invariant; the stage is an immediate) ----
    const int kr = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1);  // K row read by first-product row r (bits 2 <-> 3)
    const int kb = kr * KROWB + hh * 16;                           // k-steps 0, 1 at +32 s; second key block at +32 * KROWB
    const int kl0 = hh ? kLagOff : kr * KROWB + 64;                // k-step 2: d 32..39, or the stabiliser vector
    const int kl1 = hh ? kLagOff : (32 + kr) * KROWB + 64;
    int v0[4], v1[4];
    {
        const int fsw = (r >> 1) & 7;
        const int vr = r < 8 ? 32 + r : (r == 8 ? D : D + 1);     // d = 32 + r: real row, ones row, zero row
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int c = 2 * ks + hh;
            v0[ks] = kVOff + r * 128 + ((c ^ fsw) << 4);
            v1[ks] = kVOff + vr * 128 + (((r < 8) ? (c ^ fsw) : (c ^ 4)) << 4);  // constant rows: any chunk; this one avoids the real rows' banks
        }
    }

    // ---- DMA plan: 640 16-byte chunks per tile (K 320 + V^T 320) = three pieces per thread, the third for waves 0, 1 ----
    // issue() and wait_tile() below are tied through these two counts: a wave's counted vmcnt leaves exactly ONE younger
    // tile's pieces in flight, so wait_tile must name the number of pieces issue() gives THAT wave per tile.
    constexpr int kPiecesW01 = 3, kPiecesW23 = 2;
    static_assert((2 * kPiecesW01 + 2 * kPiecesW23) * 64 == KV * (D / 8) + D * (KV / 8), "every 16-byte chunk of a K / V^T tile is one lane of one piece");
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);
    u32x4 dK, dV;
    {
        const uint64_t bk = (uint64_t)Kb, bv = (uint64_t)Vb;
        const unsigned kbytes = (unsigned)((((int64_t)p.Nk - 1) * p.ldk + D) * 2);
        const unsigned vbytes = (unsigned)((((int64_t)D - 1) * p.ldvt + (p.Nk + 7) / 8 * 8) * 2);
        dK = u32x4{(unsigned)__builtin_amdgcn_readfirstlane((unsigned)bk), (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(bk >> 32) & 0xffffu),
                   (unsigned)__builtin_amdgcn_readfirstlane(kbytes), 0x00020000u};
        dV = u32x4{(unsigned)__builtin_amdgcn_readfirstlane((unsigned)bv), (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(bv >> 32) & 0xffffu),
                   (unsigned)__builtin_amdgcn_readfirstlane(vbytes), 0x00020000u};
    }
    auto k_src = [&](int g) { const int key = g / 5, c = g - key * 5; return (unsigned)(key * (int)p.ldk + c * 8) * 2u; };
    auto v_src = [&](int v) { const int d = v >> 3, c = (v & 7) ^ ((d >> 1) & 7); return (unsigned)(d * (int)p.ldvt + c * 8) * 2u; };
    const unsigned off0 = k_src(tid);                                             // K chunks 0..255
    const unsigned off1 = wuni == 0 ? k_src(tid + 256) : v_src(tid - 64);         // wave 0: K 256..319; waves 1-3: V^T 0..191
    const unsigned off2 = v_src(tid + 192);                                       // waves 0, 1: V^T 192..319
    const unsigned dst0 = (unsigned)wuni * 1024u;
    const unsigned dst1 = wuni == 0 ? 4096u : (unsigned)kVOff + (unsigned)(wuni - 1) * 1024u;
    const unsigned dst2 = (unsigned)kVOff + 3072u + (unsigned)wuni * 1024u;
    auto dma16 = [&](const u32x4& desc, unsigned lds_addr, unsigned voff, unsigned soff) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds"
                     :
                     : "v"(voff), "s"(lds_addr), "s"(desc), "s"(soff)
                     : "memory", "m0");
#pragma clang diagnostic pop
    };
    const int ntiles = (p.Nk + KV - 1) / KV;
    auto issue = [&](int kt, int stage) {
        const int k0 = kt * KV;
        unsigned ks_off = (unsigned)k0 * (unsigned)p.ldk * 2u, vs_off = (unsigned)k0 * 2u;
        unsigned a0 = off0, a1 = off1, a2 = off2;
        if (k0 + KV > p.Nk) {  // last, partial tile (wave-uniform): the tile offset joins the range-checked per-lane offset
            a0 += ks_off;
            a1 += wuni == 0 ? ks_off : vs_off;
            a2 += vs_off;
            ks_off = vs_off = 0;
        }
        const unsigned sb = lds_base + (unsigned)stage * kStage;
        dma16(dK, sb + dst0, a0, ks_off);                    // piece 1: every wave
        if (wuni == 0) dma16(dK, sb + dst1, a1, ks_off);     // piece 2: every wave
        else dma16(dV, sb + dst1, a1, vs_off);
        if (wuni < 2) dma16(dV, sb + dst2, a2, vs_off);      // piece 3: waves 0, 1 only (kPiecesW01 = 3, kPiecesW23 = 2)
    };
    // wait until this wave's pieces of the oldest tile in flight have landed (`more`: a younger tile stays in flight)
    auto wait_tile = [&](bool more) {
        if (!more) attn_wait_vmcnt<0>();
        else if (wuni < 2) attn_wait_vmcnt<kPiecesW01>();
        else attn_wait_vmcnt<kPiecesW23>();
    };
    auto first_product = [&](const unsigned char* Sb, int t) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int s = 0; s < 2; ++s)
            acc = Half<HT>::mfma32(*reinterpret_cast<const uint4*>(Sb + kb + t * (32 * KROWB) + 32 * s), qf[s], acc);
        return Half<HT>::mfma32(*reinterpret_cast<const uint4*>(Sb + (t ? kl1 : kl0)), qf[2], acc);
    };

    // one K / V^T tile out of ring stage STAGE
    auto tile = [&](const int kt, auto stage_c, auto lag_c) {
        constexpr bool LAG = decltype(lag_c)::value;
        constexpr int STAGE = decltype(stage_c)::value;
        const int k0 = kt * KV;
        wait_tile(kt + 1 < ntiles);
        __syncthreads();  // every wave's pieces of tile kt are in LDS; every wave is done reading tile kt-1
        const unsigned char* Sb = smem + STAGE * kStage;
        if (k0 + KV > p.Nk) {  // block-uniform: zero the V^T columns of keys >= Nk (whatever follows them in memory)
            for (int id = tid; id < D * 8; id += 256) {
                const int d = id >> 3, c = (id & 7) ^ ((d >> 1) & 7);
                const int nv = p.Nk - (k0 + c * 8);  // valid keys in this chunk
                if (nv < 8) {
                    uint4* cell = reinterpret_cast<uint4*>(smem + STAGE * kStage + kVOff + id * 16);
                    const uint4 x = *cell;
                    unsigned w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (2 * e >= nv) w[e] = 0;
                        else if (2 * e + 1 >= nv) w[e] &= 0xffffu;
                    }
                    *cell = make_uint4(w[0], w[1], w[2], w[3]);
                }
            }
            __syncthreads();
        }
        if (kt + 2 < ntiles) issue(kt + 2, (STAGE + 2) % NST);  // that stage held tile kt-1

        f32x16 st[2];
        ATTN40_PRIO_MFMA(1)
#pragma unroll
        for (int t = 0; t < 2; ++t) st[t] = first_product(Sb, t);
        ATTN40_PRIO_MFMA(0)
        ATTN40_PRIO_VALU(1)
        // register i of key block t, lane half hh  <->  key k0 + 32 t + 16 (i >> 3) + 8 hh + (i & 7)
        if (k0 + KV > p.Nk) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = k0 + 32 * t + 16 * (i >> 3) + 8 * hh + (i & 7);
                    if (key >= p.Nk) st[t][i] = kNegBig;
                }
        }
        float mx = fmaxf(st[0][0], st[1][0]);
#pragma unroll
        for (int i = 1; i < 16; ++i) mx = fmaxf(fmaxf(mx, st[0][i]), st[1][i]);
        mx = half_swap_max(mx);
        float alpha;
        if constexpr (LAG) {
            lag_overflow |= mx > kLagMax;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) st[t][i] = __builtin_amdgcn_exp2f(st[t][i]);
            const float m_new = Half<HT>::round(m_run + fmaxf(mx, 0.f));
            alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            set_stabiliser(m_new);
        } else {
            // classic maximum-first softmax (only after a stabiliser overflow): Q[q][40] = 0, scores are scale*log2e*q.k
            const float m_new = fmaxf(m_run, mx);
            alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) st[t][i] = __builtin_amdgcn_exp2f(st[t][i] - m_new);
            if (!__all(alpha == 1.0f)) {
#pragma unroll
                for (int t = 0; t < DT; ++t)
#pragma unroll
                    for (int i = 0; i < 16; ++i) ot[t][i] *= alpha;
            }
        }
        uint4 pf[4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                uint4 w;
                w.x = Half<HT>::pack2(st[t][8 * s + 0], st[t][8 * s + 1]);
                w.y = Half<HT>::pack2(st[t][8 * s + 2], st[t][8 * s + 3]);
                w.z = Half<HT>::pack2(st[t][8 * s + 4], st[t][8 * s + 5]);
                w.w = Half<HT>::pack2(st[t][8 * s + 6], st[t][8 * s + 7]);
                pf[2 * t + s] = w;
            }
        ATTN40_PRIO_VALU(0)
        ATTN40_PRIO_MFMA(1)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            ot[0] = Half<HT>::mfma32(*reinterpret_cast<const uint4*>(Sb + v0[ks]), pf[ks], ot[0]);
            ot[1] = Half<HT>::mfma32(*reinterpret_cast<const uint4*>(Sb + v1[ks]), pf[ks], ot[1]);
        }
        ATTN40_PRIO_MFMA(0)
        if constexpr (LAG) {
            if (!__all(alpha == 1.0f)) {
#pragma unroll
                for (int t = 0; t < DT; ++t)
#pragma unroll
                    for (int i = 0; i < 16; ++i) ot[t][i] *= alpha;
            }
        }
    };
    auto run = [&](auto lag_c) {
        constexpr bool LAG = decltype(lag_c)::value;
        issue(0, 0);
        if (ntiles > 1) issue(1, 1);
        if constexpr (LAG) {
            // initial stabiliser: the maximum of the first tile's scores (one extra first product)
            wait_tile(ntiles > 1);
            __syncthreads();
            float mx = kNegBig;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const f32x16 s0 = first_product(smem, t);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = 32 * t + 16 * (i >> 3) + 8 * hh + (i & 7);
                    if (key < p.Nk) mx = fmaxf(mx, s0[i]);
                }
            }
            mx = half_swap_max(mx);
            set_stabiliser(Half<HT>::round(mx));
        }
        for (int kt = 0; kt < ntiles; kt += 3) {
            tile(kt, std::integral_constant<int, 0>{}, lag_c);
            if (kt + 1 < ntiles) tile(kt + 1, std::integral_constant<int, 1>{}, lag_c);
            if (kt + 2 < ntiles) tile(kt + 2, std::integral_constant<int, 2>{}, lag_c);
        }
    };
    run(std::true_type{});
    // A score more than 2^kLagMax above the lagged stabiliser could overflow exp2: redo the whole block with the classic
    // (maximum-first) softmax.  Block-wide decision (it is also the barrier that ends the last tile's LDS reads).
    if (__syncthreads_or(lag_overflow)) {
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) ot[t][i] = 0.f;
        set_stabiliser(0.f);
        m_run = kNegBig;
        run(std::false_type{});
        __syncthreads();
    }

    // ---- normalise and store (as above): row 40 of O^T is the softmax row sum ----
    const float l_tot = __shfl(ot[1][4], r, 64);  // d = 40: tile 1, register 4 * ((40 % 32) / 8), lanes of the lower half
    const float inv = 1.0f / l_tot;
    constexpr int SROW = D * 2 + 16;
    unsigned char* strip = smem + wid * (32 * SROW);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d0 = 32 * dt + 8 * g + 4 * hh;
            if (d0 < D) {
                uint2 w;
                w.x = Half<HT>::pack2(ot[dt][4 * g + 0] * inv, ot[dt][4 * g + 1] * inv);
                w.y = Half<HT>::pack2(ot[dt][4 * g + 2] * inv, ot[dt][4 * g + 3] * inv);
                *reinterpret_cast<uint2*>(strip + r * SROW + d0 * 2) = w;
            }
        }
    __builtin_amdgcn_wave_barrier();
    constexpr int CH = D / 8;
    const int q0w = qb * 128 + wid * 32;
    bf16_t* Ow = p.O + (int64_t)b * p.sO + (int64_t)q0w * p.ldo + (int64_t)head * D;
#pragma unroll
    for (int t = 0; t < (32 * CH + 63) / 64; ++t) {
        const int idx = lane + 64 * t;
        const int rr = idx / CH, cc = idx - rr * CH;
        if (rr < 32 && q0w + rr < p.Nq)
            *reinterpret_cast<uint4*>(Ow + (int64_t)rr * p.ldo + cc * 8) = *reinterpret_cast<const uint4*>(strip + rr * SROW + cc * 16);
    }
}

template <typename HT>
int launch_attn40(const AttnParams& p, int B, int H, hipStream_t s) {
    constexpr size_t smem = 3 * (5376 + 42 * 128);
    dim3 grid(((p.Nq + 127) / 128) * H * B, 1, 1);
    attn40_kernel<HT><<<grid, 256, smem, s>>>(p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        gmd_set_error("gmd_attention: launch failed: %s", hipGetErrorString(e));
        return GMD_ERR_LAUNCH;
    }
    return GMD_OK;
}

template <typename HT, int D, bool CAUSAL = false>
int launch_attn(const AttnParams& p, int B, int H, hipStream_t s) {
    constexpr int DK = (D + 15) / 16, DT = (D + 31) / 32;
    const size_t smem = 2 * ((size_t)KV * (2 * DK + 1) * 16 + (size_t)DT * 32 * VROW);  // two pipeline stages
    dim3 grid(((p.Nq + 127) / 128) * H * B, 1, 1);  // 1-D: remapped per XCD in the kernel
    attn_fwd_kernel<HT, D, CAUSAL><<<grid, 256, smem, s>>>(p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        gmd_set_error("gmd_attention: launch failed: %s", hipGetErrorString(e));
        return GMD_ERR_LAUNCH;
    }
    return GMD_OK;
}

template <typename HT>
int dispatch_attn(const AttnParams& p, int B, int H, int D, int Nq, int Nk, int causal, hipStream_t s) {
    if (causal) {  // the text encoder's mask; instantiated for its head dims (CLIP ViT-L/14: 64)
        GMD_REQUIRE(Nq == Nk, "gmd_attention: the causal mask needs Nq == Nk");
        if (D == 64) return launch_attn<HT, 64, true>(p, B, H, s);
        if (D == 32) return launch_attn<HT, 32, true>(p, B, H, s);
        gmd_set_error("gmd_attention: causal attention is instantiated for head dims 32 and 64 only (got %d)", D);
        return GMD_ERR_UNSUPPORTED;
    }
    switch (D) {
        case 32: return launch_attn<HT, 32>(p, B, H, s);
        case 40: return launch_attn40<HT>(p, B, H, s);
        case 64: return launch_attn<HT, 64>(p, B, H, s);
        case 80: return launch_attn<HT, 80>(p, B, H, s);
        case 160: return launch_attn<HT, 160>(p, B, H, s);
        default:
            gmd_set_error("gmd_attention: head dim %d not instantiated (supported: 32, 40, 64, 80, 160)", D);
            return GMD_ERR_UNSUPPORTED;
    }
}

}  // namespace

extern "C" int gmd_attention(const void* Q, const void* K, const void* Vt, void* O, int dtype, int B, int H, int D, int Nq,
                             int Nk, int64_t ldq, int64_t ldk, int64_t ldvt, int64_t ldo, int64_t strideQ, int64_t strideK,
                             int64_t strideVt, int64_t strideO, float scale, int causal, gmd_stream_t stream) {
    if (dtype == GMD_F32S || dtype == GMD_F32SA) {  // float32 tensors, both contractions as three float16 products (attention_split.hip);
                                                     // GMD_F32SA: O stored pre-split for the out-projection that reads it
        GMD_REQUIRE(!causal, "gmd_attention: the float32 (split) kernel has no causal mask");
        GMD_REQUIRE(B >= 0 && H > 0 && Nq >= 0 && Nk > 0, "gmd_attention: bad shape B=%d H=%d Nq=%d Nk=%d", B, H, Nq, Nk);
        if (B == 0 || Nq == 0) return GMD_OK;
        GMD_REQUIRE((int64_t)((Nq + 127) / 128) * H * B < (1ll << 31), "gmd_attention: grid too large");
        GMD_REQUIRE(Q && K && Vt && O && gmd_aligned16(Q) && gmd_aligned16(K) && gmd_aligned16(Vt) && gmd_aligned16(O),
                    "gmd_attention: null or unaligned pointer");
        GMD_REQUIRE(ldq % 4 == 0 && ldk % 4 == 0 && ldvt % 4 == 0 && ldo % 4 == 0 && strideQ % 4 == 0 && strideK % 4 == 0 && strideVt % 4 == 0 &&
                        strideO % 4 == 0, "gmd_attention: float32 leading dimensions / batch strides must be multiples of 4");
        GMD_REQUIRE((int64_t)Nk * ldk < (1ll << 29) && (int64_t)D * ldvt < (1ll << 29), "gmd_attention: K / V^T slab of one head exceeds 2 GiB");
        GMD_REQUIRE(ldvt >= ((Nk + 3) / 4) * 4, "gmd_attention: ldvt=%lld must cover Nk=%d rounded up to 4", (long long)ldvt, Nk);
        GMD_REQUIRE(ldq >= (int64_t)H * D && ldk >= (int64_t)H * D && ldo >= (int64_t)H * D, "gmd_attention: row stride smaller than H*D");
        GMD_REQUIRE(dtype != GMD_F32SA || (ldo % 32 == 0 && strideO == (int64_t)Nq * ldo && (H * D) % 4 == 0),
                    "gmd_attention: a pre-split output needs contiguous rows of whole 32-element chunks (ldo=%lld)", (long long)ldo);
        return gmd_launch_attention_split(Q, K, Vt, O, B, H, D, Nq, Nk, ldq, ldk, ldvt, ldo, strideQ, strideK, strideVt, strideO, scale,
                                          dtype == GMD_F32SA ? 1 : 0, (hipStream_t)stream);
    }
    if (dtype != GMD_BF16 && dtype != GMD_F16) {
        gmd_set_error("gmd_attention: GMD_BF16 / GMD_F16 / GMD_F32S are implemented (the exact F32 path composes gmd_gemm_nt + gmd_softmax_rows)");
        return GMD_ERR_UNSUPPORTED;
    }
    GMD_REQUIRE(B >= 0 && H > 0 && Nq >= 0 && Nk > 0, "gmd_attention: bad shape B=%d H=%d Nq=%d Nk=%d", B, H, Nq, Nk);
    if (B == 0 || Nq == 0) return GMD_OK;  // empty batch / no queries: nothing to write
    GMD_REQUIRE((int64_t)((Nq + 127) / 128) * H * B < (1ll << 31), "gmd_attention: grid too large");
    GMD_REQUIRE(Q && K && Vt && O, "gmd_attention: null pointer");
    GMD_REQUIRE(gmd_aligned16(Q) && gmd_aligned16(K) && gmd_aligned16(Vt) && gmd_aligned16(O), "gmd_attention: pointers must be 16-byte aligned");
    GMD_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldvt % 8 == 0 && ldo % 8 == 0, "gmd_attention: leading dimensions must be multiples of 8");
    GMD_REQUIRE(strideQ % 8 == 0 && strideK % 8 == 0 && strideVt % 8 == 0 && strideO % 8 == 0, "gmd_attention: batch strides must be multiples of 8");
    GMD_REQUIRE((int64_t)Nk * ldk < (1ll << 30) && (int64_t)D * ldvt < (1ll << 30), "gmd_attention: K / V^T slab of one head exceeds 2 GiB");
    GMD_REQUIRE(ldvt >= ((Nk + 7) / 8) * 8, "gmd_attention: ldvt=%lld must cover Nk=%d rounded up to 8", (long long)ldvt, Nk);
    GMD_REQUIRE(ldq >= (int64_t)H * D && ldk >= (int64_t)H * D && ldo >= (int64_t)H * D, "gmd_attention: row stride smaller than H*D");
    AttnParams p;
    p.Q = (const bf16_t*)Q; p.K = (const bf16_t*)K; p.Vt = (const bf16_t*)Vt; p.O = (bf16_t*)O;
    p.Nq = Nq; p.Nk = Nk; p.H = H; p.ldq = ldq; p.ldk = ldk; p.ldvt = ldvt; p.ldo = ldo;
    p.sQ = strideQ; p.sK = strideK; p.sVt = strideVt; p.sO = strideO;
    p.scale_log2 = scale * 1.4426950408889634f;
    hipStream_t s = (hipStream_t)stream;
    return dtype == GMD_F16 ? dispatch_attn<f16_t>(p, B, H, D, Nq, Nk, causal, s) : dispatch_attn<bf16_t>(p, B, H, D, Nq, Nk, causal, s);
}

GMD_WG_TRACE_SETTER(attention)

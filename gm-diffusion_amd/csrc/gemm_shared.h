// Pieces shared by the dense-contraction translation units (gemm.hip: 16-bit and exact float32 kernels; gemm_split.hip:
// float32 on the matrix cores as three float16 products): parameter block, operand addressing, LDS image, activation.
#pragma once
#include "gmd_common.h"
#include <mutex>

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
    int64_t ldrb;        // row stride of rowbias (>= N)
    const void* residual;
    int64_t ldr, sR;
    float alpha;
    int act;
    int out_f32;
    int c_split;       // float32 split path: store the output pre-split ([hi | lo] per 32 elements, GMD_F32SA as out_dtype; full-tile row epilogues)
    unsigned a_bytes, w_bytes;  // extents of the A / W operands (one batch slab) for the buffer descriptors
    int ksplit;          // > 1: grid z splits K; raw fp32 partial sums go to `ws` [ksplit][M][N], epilogue in splitk_reduce
    float* ws;
    // conv3x3 geometry (CONV instantiations only)
    int Hin, Win, Cin, Hout, Wout, stride, upsample, pad_lo;
    // conv3x3 K order of the ring kernel: channels are walked in blocks of `cblk` (a multiple of 64 dividing Cin), all nine
    // taps of a block before the next block.  cblk == Cin is the plain tap-major order.  A smaller block keeps the rows an
    // XCD re-reads for the next tap inside its 4 MiB L2 (see gmd_conv3x3).
    int cblk;
    // optional column statistics of the stored (rounded) output, for a following GroupNorm: {sum, sum of squares} over each
    // 64-row block and each bucket of `cs_bucket` adjacent columns -> colstats[M/64][N/cs_bucket][2] (ring kernel, row epilogue)
    float* colstats;
    int cs_bucket;
    // split-K only: leave the partial slabs in `ws` and do NOT launch the reduction (the consumer sums them:
    // gmd_conv3x3_groupnorm -> gn_slab_kernel of norm.hip)
    int defer_reduce;
    // tile order of the round-4 kernels inside an XCD's contiguous run of tiles: M-panels are walked in groups of `tile_group`
    // (m fastest inside a group, then the next N tile, then the next group); 1 = n fastest (the ring kernels' order).  Chosen on
    // the host so that what an XCD re-reads between reuses stays inside its 4 MiB L2 (gemm.hip: pick_tile_group).
    int tile_group;
    // fused Q|K|V projection (gmd_gemm_qkv_vt): column tiles from vt_col0 on are the V columns and leave TRANSPOSED, as the
    // attention kernels read them: vt_out[sample][column - vt_col0][token], row stride vt_ld, `vt_tokens` rows of C per sample
    void* vt_out;
    int vt_col0, vt_tokens;
    int64_t vt_ld;
    // in-kernel split-K reduction (round 5, splitk_fixup in gemm.hip): the K slices 0 .. ksplit-2 of a tile leave their accumulator
    // fragments in `ws` and count themselves in fix_cnt[tile]; the LAST slice (dispatched last) waits for them, adds them in slice
    // order and runs the fused epilogue -- no slab round trip, no reduction launch.  fix_bytes: extent of the fragment area.
    int fixup;
    unsigned fix_bytes;
    unsigned* fix_cnt;
};

// The last GMD_WS_TAIL bytes of a caller's workspace hold the arrival counters of the in-kernel split-K reduction (one per tile):
// zero when the workspace is first handed to the library, left zero by every launch.  Slabs / fragments never reach into them.
constexpr int kFixupCounters = 16384;
constexpr int64_t kWsTail = (int64_t)kFixupCounters * 4;
static inline int64_t gmd_ws_usable_bytes(int64_t bytes) { return bytes > kWsTail ? bytes - kWsTail : 0; }

// (row tile, column tile) of linear tile index L under GemmParams::tile_group
__device__ __forceinline__ void grouped_tile(int L, int tiles_m, int tiles_n, int group, int& mt, int& nt) {
    if (group <= 1) {
        mt = L / tiles_n;
        nt = L - mt * tiles_n;
        return;
    }
    const int gsz = group * tiles_n;
    const int pg = L / gsz, r = L - pg * gsz;
    const int left = tiles_m - pg * group;
    const int rows = left < group ? left : group;  // the last group may be short
    nt = r / rows;
    mt = pg * group + (r - nt * rows);
}

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

// erf for the bf16 GEGLU epilogue: Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7 (two orders below bf16 rounding) in ~14
// instructions (one v_rcp, one v_exp) instead of libm's branchy erff -- the epilogue of the K=320 ff1 GEMM evaluates 32 of
// them per thread for only 5 K steps of MFMA work.  The float32 parity path (gmd_geglu) keeps erff.
__device__ __forceinline__ float fast_erf(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float y = fmaf(1.061405429f, t, -1.453152027f);
    y = fmaf(y, t, 1.421413741f);
    y = fmaf(y, t, -0.284496736f);
    y = fmaf(y, t, 0.254829592f);
    y = y * t * __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
    return copysignf(1.0f - y, x);
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == GMD_ACT_SILU) return silu_f(v);
    if (act == GMD_ACT_QUICK_GELU) return v / (1.0f + __expf(-1.702f * v));
    return v;
}

// ---- producer column statistics: the column pass of the row epilogues (gemm.hip epilogue_rows, gemm_split.hip epilogue_rows_f32) ----
// One float32 add that the compiler can neither pack with a neighbour nor fold away.  The column pass below keeps {sum, sum of
// squares} pairs; hipcc (ROCm 7.2) packs the pair updates into v_pk_add_f32 and, where the pair order of the two operands
// differs, adds `op_sel:[0,1] op_sel_hi:[1,0]`.  With the float32 kernels' register allocation that swizzled form sat two
// instructions in front of an EXEC change (the `column < NCOL` branch of the next 64 columns), and under a second stream's load
// lanes 48..63 of its low result came out without one of the two addends on three different MI355X (tools/stress_colstats.py:
// 1 launch in 200-1500).  Both halves of the pattern are gone: these adds are single v_add_f32, and the column
// loops run with every lane active (clamped column), only the stores predicated.
__device__ __forceinline__ float add_f32_single(float a, float b) {
    float r;
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// cs[k] / cq[k] += column sums of the strip's 32 rows for column lane + 64 k (lanes past NCOL re-read the last column: never stored)
template <int NCOL, int ROWF>
__device__ __forceinline__ void colstats_pass(const float* strip, int lane, float (&cs)[(NCOL + 63) / 64], float (&cq)[(NCOL + 63) / 64]) {
#pragma unroll
    for (int k = 0; k < (NCOL + 63) / 64; ++k) {
        const int cc = lane + 64 * k < NCOL ? lane + 64 * k : NCOL - 1;
        float s_ = 0.f, q_ = 0.f;
#pragma unroll 8
        for (int r = 0; r < 32; ++r) {
            const float x = strip[r * ROWF + cc];
            s_ += x;
            q_ += x * x;
        }
        cs[k] = add_f32_single(cs[k], s_);
        cq[k] = add_f32_single(cq[k], q_);
    }
}

// {sum, sum of squares} of each bucket of `bucket` adjacent columns of this wave tile -> out[(row block) * (N / bucket) + ...]
template <int NCOL, int ROWF>
__device__ __forceinline__ void colstats_store(float* strip, int lane, const float (&cs)[(NCOL + 63) / 64], const float (&cq)[(NCOL + 63) / 64],
                                               int bucket, float* out) {
#pragma unroll
    for (int k = 0; k < (NCOL + 63) / 64; ++k) {
        const int cc = lane + 64 * k;
        if (cc < NCOL) {
            strip[cc] = cs[k];
            strip[ROWF + cc] = cq[k];
        }
    }
    __builtin_amdgcn_wave_barrier();
    const int nb = NCOL / bucket;
    const int b0 = (lane < nb ? lane : nb - 1) * bucket;  // every lane folds a bucket (the idle ones the last): no EXEC change around the adds
    float s_ = 0.f, q_ = 0.f;
    for (int e = 0; e < bucket; ++e) {
        s_ = add_f32_single(s_, strip[b0 + e]);
        q_ = add_f32_single(q_, strip[ROWF + b0 + e]);
    }
    if (lane < nb) *reinterpret_cast<float2*>(out + lane * 2) = make_float2(s_, q_);
}

constexpr int BK = 64;  // bf16 elements per K step = 128 bytes = 8 chunks of 16 bytes

// byte offset of 16-byte chunk `chunk` of row `row` in a [rows][128 B] tile; the XOR makes both the
// 8-lane ds_write_b128 groups and the 16-lane ds_read_b128 groups of a 16x16x32 fragment conflict-free
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <int V> struct IntC { static constexpr int value = V; };

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOOB = 0xFFFF0000u;  // byte offset beyond every buffer (extents are checked < kOOB on the host)

// Operand addressing of the bf16 kernel: every staging slot keeps ONE 32-bit byte offset into a raw buffer
// resource; rows beyond M/N and conv padding taps hold kOOB, for which the buffer load returns zeros -- no
// branches and no 64-bit arithmetic in the K loop.  For conv3x3 the offsets are recomputed only when the
// K loop crosses into the next filter tap (every Cin/64 steps).
template <bool CONV>
__device__ __forceinline__ unsigned conv_tap_offset(const GemmParams& p, bool valid, int b, int oy, int ox, int ky, int kx, int chunk) {
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
    return (unsigned)(((b * p.Hin + iy) * p.Win + ix) * p.Cin) * 2u + (unsigned)chunk * 16u;
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-(function, device) property: remember it per device, so a
// process that drives several GPUs (not the one-process-per-GPU design, but legal) opts in on each of them.
hipError_t opt_in_lds(const void* fn, int bytes) {
    constexpr int kMaxDev = 64, kMaxFn = 64;
    static const void* fns[kMaxFn];
    static unsigned long long done[kMaxFn];  // bit d: set on device d
    static std::mutex mu;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    int slot = -1;
    for (int i = 0; i < kMaxFn; ++i) {
        if (fns[i] == fn) { slot = i; break; }
        if (fns[i] == nullptr) { fns[i] = fn; slot = i; break; }
    }
    if (slot >= 0 && dev < kMaxDev && (done[slot] >> dev & 1ull)) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && slot >= 0 && dev < kMaxDev) done[slot] |= 1ull << dev;
    return e;
}

}  // namespace

// Per-workgroup occupancy trace -- DIAGNOSTIC BUILD ONLY (-DGMD_WG_TRACE, tools/dbg/build_wgtrace.sh -> tools/dbg/libgmd_wgtrace.so).
// In the product build GMD_WG_TRACE_SCOPE() expands to nothing and no kernel carries a stamp.
//
// Every workgroup of every kernel appends ONE 32-byte record to its CU's array in global memory when its wave 0 leaves the kernel:
//   u64 t0, t1   the device's constant 100 MHz counter (s_memrealtime: common to all XCDs and both streams) at entry / exit of wave 0
//   u32 hw_id    HW_REG_HW_ID: wave [3:0] simd [5:4] pipe [7:6] cu [11:8] sh [12] se [15:13] ... queue [26:24] ... me [31:30]
//   u32 tag      xcc [3:0] (HW_REG_XCC_ID) | kind << 8 | (threads / 64) << 16
//   u32 block    linear workgroup id in its grid
//   u32 grid     workgroups of the launch
// tools/cu_occupancy.py folds the ring of ONE loop iteration of the shipped two-stream graph path into profiles/r05_cu_occupancy.txt:
// per CU (xcc, se, sh, cu) and per hardware queue (= stream) the resident workgroups over time.
// Cost per workgroup: two scalar clock reads, one uncontended L2 atomic add (per-CU counter), one 32-byte store by one lane.
#pragma once
#ifdef GMD_WG_TRACE
#include <hip/hip_runtime.h>
struct GmdWgTraceHeader {
    unsigned long long shards;       // 2048: one per possible CU id ((xcc * 8 + se) * 2 + sh) * 16 + cu
    unsigned long long capacity;     // records per shard (the surplus of a shard is dropped)
    unsigned long long pad[14];      // header = 128 bytes; then `shards` counters, one per 128-byte line; then the shards' record arrays
};
// one copy per translation unit (no relocatable device code): each .hip that includes this header defines a setter with
// GMD_WG_TRACE_SETTER(name), and gmd_wg_trace_enable() (elementwise.hip, diagnostic build only) calls them all
static __device__ GmdWgTraceHeader* g_wg_trace = nullptr;

struct GmdWgTraceScope {
    unsigned long long t0;
    unsigned kind;
    __device__ __forceinline__ explicit GmdWgTraceScope(unsigned k) : kind(k) {
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    }
    __device__ __forceinline__ ~GmdWgTraceScope() {
        if (threadIdx.x == 0 && threadIdx.y == 0 && threadIdx.z == 0) {
            GmdWgTraceHeader* h = g_wg_trace;
            if (h != nullptr) {
                unsigned long long t1;
                asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
                const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID, all 32 bits
                const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
                // One counter per CU: a single counter for the chip is a ~68 ns serial point per workgroup (measured: 5.0 M workgroups in
                // 340 ms -- the traced loop ran at HALF speed).  A CU's counter is only ever touched from its own XCD, whose L2 executes
                // the atomic, so workgroup scope (no sc1: the add stays in that L2) is enough and costs a few hundred ns, uncontended.
                const unsigned shard = ((((xcc & 7u) * 8u + ((hw >> 13) & 7u)) * 2u + ((hw >> 12) & 1u)) * 16u) + ((hw >> 8) & 15u);
                unsigned long long* counters = reinterpret_cast<unsigned long long*>(h + 1);
                const unsigned long long i = __hip_atomic_fetch_add(counters + 16ull * shard, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (i < h->capacity) {
                    const unsigned nthreads = blockDim.x * blockDim.y * blockDim.z;
                    const unsigned block = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
                    const unsigned grid = gridDim.x * gridDim.y * gridDim.z;
                    uint4* rec = reinterpret_cast<uint4*>(counters + 16ull * h->shards) + 2ull * (shard * h->capacity + i);
                    rec[0] = make_uint4((unsigned)t0, (unsigned)(t0 >> 32), (unsigned)t1, (unsigned)(t1 >> 32));
                    rec[1] = make_uint4(hw, (xcc & 15u) | (kind << 8) | ((nthreads >> 6) << 16), block, grid);
                }
            }
        }
    }
};
#define GMD_WG_TRACE_SCOPE(kind) GmdWgTraceScope gmd_wg_trace_scope_(kind)
#define GMD_WG_TRACE_SETTER(name)                                                                          \
    extern "C" int gmd_wg_trace_set_##name(void* ring) {                                                   \
        return hipMemcpyToSymbol(HIP_SYMBOL(g_wg_trace), &ring, sizeof(ring)) == hipSuccess ? 0 : 1;       \
    }
#else
#define GMD_WG_TRACE_SCOPE(kind)
#define GMD_WG_TRACE_SETTER(name)
#endif

// kernel kinds (bit 6 set: the convolution instantiation of a GEMM kernel)
enum {
    WGK_GEMM64 = 1, WGK_RING = 2, WGK_PP = 3, WGK_LC = 4, WGK_PATCH = 5, WGK_PATCH_CONT = 6, WGK_SPLITK_REDUCE = 7, WGK_FF_FUSED = 8,
    WGK_ATTN = 9, WGK_ATTN40 = 10, WGK_GN_PARTIAL = 11, WGK_GN_APPLY_WS = 12, WGK_GN_APPLY_CS = 13, WGK_GN_FUSED = 14,
    WGK_GN_FUSED_REG = 15, WGK_GN_SLAB = 16, WGK_LN = 17, WGK_LN_PACKED = 18, WGK_CONCAT = 19, WGK_DUP = 20, WGK_PACK = 21,
    WGK_UNPACK = 22, WGK_LATENT_STEP = 23, WGK_TEMB = 24, WGK_CAST = 25, WGK_STAMP = 26, WGK_GN_FINALIZE = 27, WGK_GN_APPLY = 28,
    WGK_CFG_RATIO = 29, WGK_GEMM_F32 = 30, WGK_OTHER = 31, WGK_CONV_BIT = 64
};

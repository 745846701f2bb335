// Sustained matrix-core rate of register-resident MFMA streams on the whole chip, by instruction shape and operand data:
// is the ~1.0-1.1 PFLOP/s the GEMM / conv kernels reach on random data (profiles/r04_zero_data_probe.txt) a property of the
// kernels or of the chip's power management, and does the 32x32x16 shape sustain more than 16x16x32?   (measurement tool only)
//   hipcc -O3 --offload-arch=gfx950 -o mfma_power_probe mfma_power_probe.hip && ./mfma_power_probe [iterations [data 0|1|2 [type 0|1]]]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <cstring>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: 16x16x32 bf16, 4x4 accumulators (64 x 64 wave tile); 1: 32x32x16 bf16, 2x2 accumulators (64 x 64 wave tile);
// 2: 16x16x32 f16; 3: 32x32x16 f16
template <int MODE>
__global__ __launch_bounds__(512) void probe(const u32x4* __restrict__ src, float* __restrict__ out, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    u32x4 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = src[(size_t)tid * 8 + i];
        b[i] = src[(size_t)tid * 8 + 4 + i];
    }
    float r = 0.f;
    if constexpr (MODE == 0 || MODE == 2) {
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (MODE == 0)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
                    else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[i]), __builtin_bit_cast(f16x8, b[j]), acc[i][j], 0, 0, 0);
                }
            asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) r += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    } else {
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)  // two k-halves: the same 64 x 64 x 32 products per iteration as MODE 0
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if constexpr (MODE == 1)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[2 * kk + i]), __builtin_bit_cast(bf16x8, b[2 * kk + j]), acc[i][j], 0, 0, 0);
                        else
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[2 * kk + i]), __builtin_bit_cast(f16x8, b[2 * kk + j]), acc[i][j], 0, 0, 0);
                    }
            asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) r += acc[i][j][e];
    }
    if (r == 12345.678f) out[tid] = r;  // keeps the accumulators alive
}

static unsigned short bf16_bits(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }
static unsigned short f16_bits(float f) { _Float16 h = (_Float16)f; unsigned short s; memcpy(&s, &h, 2); return s; }

template <int MODE>
static void run(const char* name, const u32x4* d, float* out, int wgs, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(probe<MODE>, dim3(wgs), dim3(512), 0, 0, d, out, iters);
    hipDeviceSynchronize();
    const int reps = 6;
    float best = 1e30f, sum = 0.f;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<MODE>, dim3(wgs), dim3(512), 0, 0, d, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
        sum += ms;
    }
    const double flops = 2.0 * 64 * 64 * 32 * (double)iters * wgs * 8;  // per wave and iteration: 64 x 64 x 32 MACs
    printf("%-34s %8.2f ms avg %8.2f ms best  -> %7.0f TFLOP/s avg, %7.0f best\n", name, sum / reps, best, flops / (sum / reps) * 1e-9, flops / best * 1e-9);
}

int main(int argc, char** argv) {
    const int wgs = 256, iters = argc > 1 ? atoi(argv[1]) : 400000;
    const int only_data = argc > 2 ? atoi(argv[2]) : -1, only_half = argc > 3 ? atoi(argv[3]) : -1;  // restrict to one data pattern / type
    const size_t n = (size_t)wgs * 512 * 8;  // u32x4 per thread: 8
    std::vector<u32x4> h(n);
    u32x4* d; float* out;
    hipMalloc(&d, n * sizeof(u32x4));
    hipMalloc(&out, (size_t)wgs * 512 * 4);
    for (int data = 0; data < 3; ++data) {  // 0: zeros, 1: random normal bf16 / f16, 2: all ones
        for (int half = 0; half < 2; ++half) {
            if ((only_data >= 0 && data != only_data) || (only_half >= 0 && half != only_half)) continue;
            srand(1);
            for (size_t i = 0; i < n; ++i) {
                unsigned short v[8];
                for (int e = 0; e < 8; ++e) {
                    float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = rand() / (float)RAND_MAX;
                    float x = data == 0 ? 0.f : data == 2 ? 1.f : sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2) * 0.05f;
                    v[e] = half ? f16_bits(x) : bf16_bits(x);
                }
                memcpy(&h[i], v, 16);
            }
            hipMemcpy(d, h.data(), n * sizeof(u32x4), hipMemcpyHostToDevice);
            const char* dn = data == 0 ? "zeros" : data == 1 ? "random" : "ones";
            char name[64];
            if (!half) {
                snprintf(name, 64, "bf16 16x16x32  %s", dn); run<0>(name, d, out, wgs, iters);
                snprintf(name, 64, "bf16 32x32x16  %s", dn); run<1>(name, d, out, wgs, iters);
            } else {
                snprintf(name, 64, "f16  16x16x32  %s", dn); run<2>(name, d, out, wgs, iters);
                snprintf(name, 64, "f16  32x32x16  %s", dn); run<3>(name, d, out, wgs, iters);
            }
        }
    }
    return 0;
}

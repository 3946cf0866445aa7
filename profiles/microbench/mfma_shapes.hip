// mfma_shapes.hip - does the f16 MFMA SHAPE matter at the package power cap?  The same FLOP count per wave and iteration on
// v_mfma_f32_32x32x16_f16 (what the SA / GA kernels use) and on v_mfma_f32_16x16x32_f16, random and all-zero operands,
// 2 and 4 waves per SIMD.      hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_shapes profiles/microbench/mfma_shapes.hip && /tmp/mfma_shapes
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int ZERO>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    half8 a[4], b[4];
    uint32_t st = 0x9E3779B9u * (threadIdx.x + 1) + blockIdx.x;
    for (int i = 0; i < 4; i++)
        for (int e = 0; e < 8; e++) {
            st = st * 1664525u + 1013904223u;
            a[i][e] = ZERO ? (_Float16)0.f : (_Float16)(((int)(st >> 9) & 0xFFFF) * (1.0f / 32768.f) - 1.0f);
            st = st * 1664525u + 1013904223u;
            b[i][e] = ZERO ? (_Float16)0.f : (_Float16)(((int)(st >> 9) & 0xFFFF) * (1.0f / 32768.f) - 1.0f);
        }
    float s = 0.f;
    if constexpr (SHAPE == 0) {          // 8 x 32x32x16 per trip = 8 x 32768 FLOP
        f32x16 acc[8];
        for (int i = 0; i < 8; i++)
            for (int e = 0; e < 16; e++) acc[i][e] = 0.f;
        for (int it = 0; it < iters; it += 4) {
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 3], b[(i + r) & 3], acc[i], 0, 0, 0);
            if (ZERO) asm volatile("" : "+v"(a[0]), "+v"(b[0]));
        }
        for (int i = 0; i < 8; i++)
            for (int e = 0; e < 16; e++) s += acc[i][e];
    } else {                              // 16 x 16x16x32 per trip = 16 x 16384 FLOP (same FLOPs, same number of accumulator registers: 64)
        f32x4 acc[16];
        for (int i = 0; i < 16; i++)
            for (int e = 0; e < 4; e++) acc[i][e] = 0.f;
        for (int it = 0; it < iters; it += 4) {
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int i = 0; i < 16; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[(i + r) & 3], acc[i], 0, 0, 0);
            if (ZERO) asm volatile("" : "+v"(a[0]), "+v"(b[0]));
        }
        for (int i = 0; i < 16; i++)
            for (int e = 0; e < 4; e++) s += acc[i][e];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int SHAPE, int ZERO>
void run(float* out, int waves_per_simd) {
    const int blocks = 256 * waves_per_simd, iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SHAPE, ZERO>), dim3(blocks), dim3(256), 0, 0, out, 1000);
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<SHAPE, ZERO>), dim3(blocks), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double flop = (double)blocks * 4 * iters * 8 * (2.0 * 32 * 32 * 16);
    printf("%-10s %-6s operands, %d waves/SIMD: %7.1f TFLOP/s\n", SHAPE ? "16x16x32" : "32x32x16", ZERO ? "zero" : "random", waves_per_simd, flop / (best * 1e-3) / 1e12);
}

int main() {
    float* out;
    (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int w : {2, 4}) {
        run<0, 0>(out, w);
        run<1, 0>(out, w);
        run<0, 1>(out, w);
        run<1, 1>(out, w);
    }
    return 0;
}

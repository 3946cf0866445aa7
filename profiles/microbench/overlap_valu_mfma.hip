#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
#define SB() __builtin_amdgcn_sched_barrier(0)
// MODE 0: MFMA only, 1: fillers only, 2: MFMA + NF fillers per group placed in one gap, 3: fillers spread after each MFMA
template <int NT, int NF, int MODE>
__global__ __launch_bounds__(NT, 1) void k(const float* __restrict__ in, float* __restrict__ out, int iters) {
    extern __shared__ float lds[];
    half8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)in[threadIdx.x + i]; b[i] = (_Float16)in[threadIdx.x + 64 + i]; }
    f32x16 acc, accx;
    for (int e = 0; e < 16; e++) { acc[e] = 0.f; accx[e] = 0.f; }
    float f[8];
    for (int i = 0; i < 8; i++) f[i] = in[threadIdx.x + i * 64];
    const float c0 = in[1], c1 = in[2];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int g = 0; g < 16; g++) {
            SB();
            if (MODE != 1) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
                if (MODE == 3) { SB();
#pragma unroll
                    for (int i = 0; i < NF / 3; i++) f[i % 8] = fmaf(f[i % 8], c0, c1);
                    SB(); }
                accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, accx, 0, 0, 0);
            }
            SB();
            if (MODE == 1 || MODE == 2) {
#pragma unroll
                for (int i = 0; i < NF; i++) f[i % 8] = fmaf(f[i % 8], c0, c1);
            }
            if (MODE == 3) {
#pragma unroll
                for (int i = 0; i < NF / 3; i++) f[i % 8] = fmaf(f[i % 8], c0, c1);
            }
            SB();
            if (MODE != 1) accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, accx, 0, 0, 0);
            if (MODE == 3) { SB();
#pragma unroll
                for (int i = 0; i < NF - 2 * (NF / 3); i++) f[i % 8] = fmaf(f[i % 8], c0, c1);
            }
        }
    }
    float s = 0;
    for (int e = 0; e < 16; e++) s += acc[e] + accx[e];
    for (int i = 0; i < 8; i++) s += f[i];
    out[blockIdx.x * NT + threadIdx.x] = s;
}
template <int NT, int NF, int MODE>
void run(const float* in, float* out, const char* name) {
    const int iters = 4000;
    auto kern = k<NT, NF, MODE>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(256), dim3(NT), 100 * 1024, 0, in, out, iters); hipDeviceSynchronize();
    hipEventRecord(a); hipLaunchKernelGGL(kern, dim3(256), dim3(NT), 100 * 1024, 0, in, out, iters); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-44s NT=%d NF=%2d mode=%d  %7.3f ms  %7.1f ns/iter (16 groups of 3 MFMA)  -> %.0f cyc/iter @2.4GHz\n", name, NT, NF, MODE, ms, ms * 1e6 / iters, ms * 1e6 / iters * 2.4);
}
int main() {
    float *in, *out; hipMalloc(&in, 1 << 22); hipMalloc(&out, 1 << 22); hipMemset(in, 0, 1 << 22);
    run<256, 0, 0>(in, out, "1 wave/SIMD mfma only");
    run<512, 0, 0>(in, out, "2 waves/SIMD mfma only");
    run<256, 15, 1>(in, out, "1 wave/SIMD fillers only");
    run<512, 15, 1>(in, out, "2 waves/SIMD fillers only");
    run<256, 15, 2>(in, out, "1 wave/SIMD both, one gap");
    run<512, 15, 2>(in, out, "2 waves/SIMD both, one gap");
    run<256, 15, 3>(in, out, "1 wave/SIMD both, spread");
    run<512, 15, 3>(in, out, "2 waves/SIMD both, spread");
    run<512, 6, 2>(in, out, "2 waves/SIMD both, one gap");
    run<512, 6, 3>(in, out, "2 waves/SIMD both, spread");
    run<512, 30, 2>(in, out, "2 waves/SIMD both, one gap");
    run<512, 30, 3>(in, out, "2 waves/SIMD both, spread");
    run<512, 30, 1>(in, out, "2 waves/SIMD fillers only");
    return 0;
}

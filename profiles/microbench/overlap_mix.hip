#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
#define SB() __builtin_amdgcn_sched_barrier(0)
// per group of 3 MFMAs: NV VALU fmas, NS SALU adds, NL ds_read_b128, NA LDS atomic max
template <int NT, int NV, int NS, int NL, int NA, int MF>
__global__ __launch_bounds__(NT, 2) void k(const float* __restrict__ in, float* __restrict__ out, int iters) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 16384; i += NT) lds[i] = in[i & 1023];
    __syncthreads();
    half8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)in[threadIdx.x + i]; b[i] = (_Float16)in[threadIdx.x + 64 + i]; }
    f32x16 acc, accx;
    for (int e = 0; e < 16; e++) { acc[e] = 0.f; accx[e] = 0.f; }
    float f[8];
    for (int i = 0; i < 8; i++) f[i] = in[threadIdx.x + i * 64];
    const float c0 = in[1], c1 = in[2];
    int sacc = iters;
    f32x4 l[4] = {};
    const f32x4* lp = (const f32x4*)lds + (threadIdx.x & 63) * 2;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int g = 0; g < 16; g++) {
            SB();
            if (MF) { acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
                      accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, accx, 0, 0, 0); }
            SB();
#pragma unroll
            for (int i = 0; i < NL; i++) l[i] = lp[(g * 4 + i) * 130];
#pragma unroll
            for (int i = 0; i < NV; i++) f[i % 8] = fmaf(f[i % 8], c0, c1);
#pragma unroll
            for (int i = 0; i < NS; i++) asm volatile("s_add_u32 %0, %0, 7" : "+s"(sacc));
#pragma unroll
            for (int i = 0; i < NA; i++) atomicMax((int*)lds + 8192 + ((threadIdx.x * 1 + i * 67 + g) & 4095), (int)threadIdx.x);
            SB();
            if (MF) accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, accx, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NL; i++) f[i] += l[i][0];
        }
    }
    float s = (float)sacc;
    for (int e = 0; e < 16; e++) s += acc[e] + accx[e];
    for (int i = 0; i < 8; i++) s += f[i];
    out[blockIdx.x * NT + threadIdx.x] = s;
}
template <int NT, int NV, int NS, int NL, int NA, int MF>
void run(const float* in, float* out) {
    const int iters = 3000;
    auto kern = k<NT, NV, NS, NL, NA, MF>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(256), dim3(NT), 100 * 1024, 0, in, out, iters); hipDeviceSynchronize();
    hipEventRecord(a); hipLaunchKernelGGL(kern, dim3(256), dim3(NT), 100 * 1024, 0, in, out, iters); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("waves/SIMD=%d mfma=%d  per group: VALU %2d SALU %2d ds_read_b128 %d lds_atomic %d  -> %6.0f ns/iter = %5.0f cyc@2.4GHz (16 groups)\n", NT / 256, MF * 3, NV, NS, NL, NA, ms * 1e6 / iters, ms * 1e6 / iters * 2.4);
}
int main() {
    float *in, *out; hipMalloc(&in, 1 << 22); hipMalloc(&out, 1 << 22); hipMemset(in, 0, 1 << 22);
    run<512, 0, 0, 0, 0, 1>(in, out);
    run<512, 15, 0, 0, 0, 1>(in, out);
    run<512, 0, 15, 0, 0, 1>(in, out);
    run<512, 0, 0, 2, 0, 1>(in, out);
    run<512, 0, 0, 4, 0, 1>(in, out);
    run<512, 0, 0, 0, 1, 1>(in, out);
    run<512, 0, 0, 0, 2, 1>(in, out);
    run<512, 15, 8, 2, 1, 1>(in, out);
    run<512, 15, 8, 2, 1, 0>(in, out);
    run<512, 0, 15, 0, 0, 0>(in, out);
    run<512, 0, 0, 4, 0, 0>(in, out);
    run<512, 0, 0, 0, 2, 0>(in, out);
    return 0;
}

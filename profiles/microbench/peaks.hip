// On-box peaks that bench.py prints next to the nominal ones ("re-measure on the box and state the values used").
//   t2p_peak_mfma_f16:  dense v_mfma_f32_32x32x16_f16 rate, every SIMD of the chip busy (4 waves per CU-SIMD, 8 independent
//                       accumulators per wave), TFLOP/s
//   t2p_peak_mfma_f16_random: the same loop with eight operand pairs of pseudo-random values that rotate from MFMA to MFMA: the
//                       chip is power-limited under matrix load, and operands that never change (or are zero) toggle fewer wires
//                       than real data, so the constant-operand figure above is the OPTIMISTIC peak
//   t2p_peak_copy:      float4 copy of a buffer far larger than the 256 MB Infinity Cache, GB/s (read + written bytes)
// Built by text2pos-cvpr2022_amd/build.py into profiles/microbench/libt2p_peaks.so; measurement code, not part of the product.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_mfma_peak(float* out, int iters) {
    half8 a, b;
    for (int e = 0; e < 8; e++) {
        a[e] = (_Float16)(0.001f * (threadIdx.x + e));
        b[e] = (_Float16)(0.002f * (threadIdx.x % 7 + e));
    }
    f32x16 acc[8];
    for (int i = 0; i < 8; i++)
        for (int e = 0; e < 16; e++) acc[i][e] = 0.f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; i++)
        for (int e = 0; e < 16; e++) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// MODE 0: random values, 1: the constant values of k_mfma_peak, 2: one operand pair for all MFMAs (= k_mfma_peak when NACC = 8),
// 3: ALL-ZERO operands (nothing toggles: the rate the data sheet - and the micro-architecture guide's 2,495 TFLOP/s - describe)
template <int NACC, int MODE>
__global__ __launch_bounds__(256) void k_mfma_peak_random(float* out, int iters) {
    half8 a[4], b[4];
    uint32_t st = 0x9E3779B9u * (threadIdx.x + 1) + blockIdx.x;
    for (int i = 0; i < 4; i++)
        for (int e = 0; e < 8; e++) {
            st = st * 1664525u + 1013904223u;
            a[i][e] = MODE == 3 ? (_Float16)0.f : MODE ? (_Float16)(0.001f * (threadIdx.x + e)) : (_Float16)(((int)(st >> 9) & 0xFFFF) * (1.0f / 32768.f) - 1.0f);
            st = st * 1664525u + 1013904223u;
            b[i][e] = MODE == 3 ? (_Float16)0.f : MODE ? (_Float16)(0.002f * (threadIdx.x % 7 + e)) : (_Float16)(((int)(st >> 9) & 0xFFFF) * (1.0f / 32768.f) - 1.0f);
        }
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; i++)
        for (int e = 0; e < 16; e++) acc[i][e] = 0.f;
    for (int it = 0; it < iters; it += 2) {      // 16 MFMAs per trip = 2 x the 8 of k_mfma_peak
#pragma unroll
        for (int r = 0; r < 16 / NACC; r++)
#pragma unroll
            for (int i = 0; i < NACC; i++)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[MODE == 2 ? 0 : (i + r) & 3], b[MODE == 2 ? 0 : (i / 4 + i + 2 * r + 1) & 3], acc[i], 0, 0, 0);
        if constexpr (MODE == 3) asm volatile("" : "+v"(a[0]), "+v"(b[0]));   // (keeps the optimiser from folding 0 x 0)
    }
    float s = 0.f;
    for (int i = 0; i < NACC; i++)
        for (int e = 0; e < 16; e++) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_copy(const f32x4* __restrict__ src, f32x4* __restrict__ dst, size_t n) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

// the same copy as ONE 1024-thread workgroup per CU (x 2), four 16-byte loads in flight per thread before the stores
__global__ __launch_bounds__(1024) void k_copy_wide(const f32x4* __restrict__ src, f32x4* __restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * 1024;
    size_t i = blockIdx.x * (size_t)1024 + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const f32x4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a;
        dst[i + stride] = b;
        dst[i + 2 * stride] = c;
        dst[i + 3 * stride] = d;
    }
    for (; i < n; i += stride) dst[i] = src[i];
}

static double time_ms(hipEvent_t a, hipEvent_t b) {
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

static int peak_mfma(double* tflops, int mode);
extern "C" int t2p_peak_mfma_f16(double* tflops) { return peak_mfma(tflops, 0); }
extern "C" int t2p_peak_mfma_f16_random(double* tflops) { return peak_mfma(tflops, 1); }
// control: the rotating-register loop of the random variant with the constant values of the first (same instruction stream,
// other data)
extern "C" int t2p_peak_mfma_f16_rotating_constant(double* tflops) { return peak_mfma(tflops, 2); }
// control of the control: the four-accumulator loop with ONE operand pair
extern "C" int t2p_peak_mfma_f16_four_acc(double* tflops) { return peak_mfma(tflops, 3); }
// the rotating-register loop on ALL-ZERO operands: the chip's un-throttled rate (no operand bit ever toggles)
extern "C" int t2p_peak_mfma_f16_zero(double* tflops) { return peak_mfma(tflops, 4); }
static int peak_mfma(double* tflops, int mode) {
    auto kern = mode == 1 ? k_mfma_peak_random<8, 0> : (mode == 2 ? k_mfma_peak_random<8, 1> : (mode == 3 ? k_mfma_peak_random<4, 2> : (mode == 4 ? k_mfma_peak_random<8, 3> : k_mfma_peak)));
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 1;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int blocks = cus * 4, iters = 20000;   // 4 blocks x 4 waves per CU = 4 waves per SIMD
    float* out = nullptr;
    if (hipMalloc(&out, (size_t)blocks * 256 * 4) != hipSuccess) return 2;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1000);   // warm-up
    double best = 1e30;
    for (int r = 0; r < 3; r++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        const double ms = time_ms(e0, e1);
        best = ms < best ? ms : best;
    }
    const double flop = (double)blocks * 4 /*waves*/ * iters * 8 * (2.0 * 32 * 32 * 16);
    *tflops = flop / (best * 1e-3) / 1e12;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipFree(out);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

extern "C" int t2p_peak_copy(double* gbps) {
    const size_t bytes = (size_t)2 << 30;   // 2 GiB source + 2 GiB destination
    f32x4 *src = nullptr, *dst = nullptr;
    if (hipMalloc(&src, bytes) != hipSuccess || hipMalloc(&dst, bytes) != hipSuccess) {
        if (src) hipFree(src);
        return 2;
    }
    hipMemset(src, 1, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const size_t n = bytes / 16;
    hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, src, dst, n);
    double best = 1e30;
    for (int r = 0; r < 3; r++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, src, dst, n);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        const double ms = time_ms(e0, e1);
        best = ms < best ? ms : best;
    }
    {   // second form: 1024-thread workgroups, 2 per CU, 4 loads in flight per thread; the better of the two is reported
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        hipLaunchKernelGGL(k_copy_wide, dim3(2 * cus), dim3(1024), 0, 0, src, dst, n);
        for (int r = 0; r < 3; r++) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_copy_wide, dim3(2 * cus), dim3(1024), 0, 0, src, dst, n);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            const double ms = time_ms(e0, e1);
            best = ms < best ? ms : best;
        }
    }
    *gbps = 2.0 * bytes / (best * 1e-3) / 1e9;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipFree(src);
    hipFree(dst);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

// MFMA rate against the number of INDEPENDENT accumulator chains per wave and the waves per SIMD (one constant operand pair):
// mode = NACC in {1, 2, 4, 8}, waves_per_simd = workgroups of 256 threads per CU
extern "C" int t2p_peak_mfma_chains(int nacc, int waves_per_simd, double* tflops) {
    auto kern = nacc == 1 ? k_mfma_peak_random<1, 2> : nacc == 2 ? k_mfma_peak_random<2, 2> : nacc == 4 ? k_mfma_peak_random<4, 2> : k_mfma_peak_random<8, 2>;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 1;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int blocks = cus * waves_per_simd, iters = 20000;
    float* out = nullptr;
    if (hipMalloc(&out, (size_t)blocks * 256 * 4) != hipSuccess) return 2;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1000);
    double best = 1e30;
    for (int r = 0; r < 3; r++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        const double ms = time_ms(e0, e1);
        best = ms < best ? ms : best;
    }
    const double flop = (double)blocks * 4 * iters * 8 * (2.0 * 32 * 32 * 16);
    *tflops = flop / (best * 1e-3) / 1e12;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipFree(out);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

// mix2.hip - what a VALU / LDS / SALU instruction costs next to f16 MFMAs on one SIMD, by WHERE THE ACCUMULATOR LIVES.
//
// hipcc selects the "VGPR form" of v_mfma (vdst / srcC in architectural VGPRs) for every kernel of this repository that runs two
// waves per SIMD (<= 256 registers: NumAgprs 0 in all of them) - and all earlier micro-benchmarks of this directory inherited
// that choice.  This one issues the SAME instruction mixes with the accumulators in AGPRs (inline asm, "+a" constraints): the
// MFMA's 16-register C read / D write then goes through the accumulator file instead of competing with the VALU operand reads.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mix2 profiles/microbench/mix2.hip && /tmp/mix2
//
// One workgroup per CU, 16 groups of 3 v_mfma_f32_32x32x16_f16 per iteration (two accumulators, the third MFMA depends on the
// second: the SA edge kernels' shape); fillers per group between the MFMAs.  Cycles per iteration at a nominal 2.4 GHz.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
#define SB() __builtin_amdgcn_sched_barrier(0)

template <int AG>
__device__ __forceinline__ void mfma(f32x16& acc, const half8& a, const half8& b) {
    if constexpr (AG) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    else acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
}

// per group of 3 MFMAs: NV VALU (VT 0: v_fma_f32 chains; 1: the staging mix sub / max / cvt_pk / fma_mix), NS SALU adds,
// NL ds_read_b128, NA LDS float atomic max (fed from the accumulator), NW ds_write_b64
template <int NT, int WPS, int AG, int MF, int NV, int VT, int NS, int NL, int NA, int NW>
__global__ __launch_bounds__(NT, WPS) void k(const float* __restrict__ in, float* __restrict__ out, int iters) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 16384; i += NT) lds[i] = in[i & 1023];
    __syncthreads();
    half8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)in[threadIdx.x + i]; b[i] = (_Float16)in[threadIdx.x + 64 + i]; }
    f32x16 acc, accx;
    for (int e = 0; e < 16; e++) { acc[e] = in[e]; accx[e] = in[e + 16]; }
    float f[8];
    for (int i = 0; i < 8; i++) f[i] = in[threadIdx.x + i * 64];
    const float c0 = in[1], c1 = in[2];
    int sacc = iters;
    f32x4 l[4] = {};
    const f32x4* lp = (const f32x4*)lds + (threadIdx.x & 63) * 2;
    float* accl = lds + 8192 + (threadIdx.x & 63);
    uint2* wl = (uint2*)(lds + 12288) + threadIdx.x;
    const uint32_t lp_a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const f32x4*)lp;
    const uint32_t wl_a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint2*)wl;
    uint2 wv = {threadIdx.x, threadIdx.x * 3u};
    uint32_t keep = 0;
    const uint32_t accl_a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)accl;   // LDS byte address
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int g = 0; g < 16; g++) {
            SB();
            if (MF) { mfma<AG>(acc, a, b); mfma<AG>(accx, a, b); }
            SB();
            // every filler is ONE asm statement: exact instructions, exact counts (plain C++ fmas were SLP-packed into v_pk_fma_f32
            // and half of the LDS reads narrowed by the optimiser in overlap_mix.hip)
#pragma unroll
            for (int i = 0; i < NL; i++)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(l[i]) : "v"(lp_a), "n"(((0 * 4 + i) * 130 * 16) & 0xFFF0) : "memory");
            if constexpr (VT == 0) {
#pragma unroll
                for (int i = 0; i < NV; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i % 8]) : "v"(c0), "v"(c1));
            } else {
                // staging of 4 elements = 16 VALU: 4 sub, 4 max, 2 cvt_pk, 4 fma_mix, 2 cvt_pk  (NV / 16 such blocks)
#pragma unroll
                for (int i = 0; i < NV / 16; i++) {
                    float v[4];
                    uint32_t h01, h23, l01, l23;
#pragma unroll
                    for (int e = 0; e < 4; e++) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(v[e]) : "v"(f[e]), "v"(f[4 + e]));
#pragma unroll
                    for (int e = 0; e < 4; e++) asm volatile("v_max_f32 %0, %0, 0" : "+v"(v[e]));
                    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h01) : "v"(v[0]), "v"(v[1]));
                    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h23) : "v"(v[2]), "v"(v[3]));
                    asm volatile("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(v[0]) : "v"(h01));
                    asm volatile("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(v[1]) : "v"(h01));
                    asm volatile("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(v[2]) : "v"(h23));
                    asm volatile("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(v[3]) : "v"(h23));
                    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(l01) : "v"(v[0]), "v"(v[1]));
                    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(l23) : "v"(v[2]), "v"(v[3]));
                    keep ^= h01 ^ h23 ^ l01 ^ l23;   // (2 v_xor3: bookkeeping so the results stay live)
                }
            }
#pragma unroll
            for (int i = 0; i < NS; i++) asm volatile("s_add_u32 %0, %0, 7" : "+s"(sacc));
#pragma unroll
            for (int i = 0; i < NW; i++)
                asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(wl_a), "v"(wv), "n"(i * 4096) : "memory");
            if constexpr (NA > 0) {
                if constexpr (AG) {
#pragma unroll
                    for (int i = 0; i < NA; i++)
                        asm volatile("ds_max_f32 %0, %1 offset:%2" ::"v"(accl_a), "a"(acc[(g + i) & 15]), "n"(256 * ((0 + i) & 15)) : "memory");
                } else {
#pragma unroll
                    for (int i = 0; i < NA; i++)
                        (void)__hip_atomic_fetch_max(accl + 64 * ((g + i) & 15), acc[(g + i) & 15], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            SB();
            if (MF) mfma<AG>(accx, b, a);
            if constexpr (NL > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the real kernel's operand wait)
        }
    }
    float s = (float)sacc;
    for (int e = 0; e < 16; e++) s += acc[e] + accx[e];
    for (int i = 0; i < 8; i++) s += f[i];
    for (int i = 0; i < 4; i++) s += l[i][0] + l[i][3];
    out[blockIdx.x * NT + threadIdx.x] = s + (float)keep;
}

template <int NT, int WPS, int AG, int MF, int NV, int VT, int NS, int NL, int NA, int NW>
void run(const float* in, float* out, const char* what) {
    const int iters = 2000;
    auto kern = k<NT, WPS, AG, MF, NV, VT, NS, NL, NA, NW>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(256), dim3(NT), 100 * 1024, 0, in, out, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(256), dim3(NT), 100 * 1024, 0, in, out, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    printf("%-4s waves/SIMD=%d mfma/grp=%d | VALU %2d%s SALU %2d ds_read_b128 %d atomic %d ds_write_b64 %d | %6.0f ns/iter = %5.0f cyc@2.4GHz  %s\n",
           AG ? "AGPR" : "VGPR", NT / 256, MF * 3, NV, VT ? "(stage)" : "(fma)  ", NS, NL, NA, NW, ms * 1e6 / iters, ms * 1e6 / iters * 2.4, what);
}

#define BOTH(NT, WPS, MF, NV, VT, NS, NL, NA, NW, what)            \
    run<NT, WPS, 0, MF, NV, VT, NS, NL, NA, NW>(in, out, what);   \
    run<NT, WPS, 1, MF, NV, VT, NS, NL, NA, NW>(in, out, what)

int main() {
    float *in, *out;
    (void)hipMalloc(&in, 1 << 22);
    (void)hipMalloc(&out, 1 << 22);
    {   // random-ish operand bits (the chip is power-limited under matrix load: constant operands flatter the clock)
        float* h = (float*)malloc(1 << 22);
        unsigned s = 12345u;
        for (int i = 0; i < (1 << 20); i++) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
        (void)hipMemcpy(in, h, 1 << 22, hipMemcpyHostToDevice);
        free(h);
    }
#ifdef MIX2_LONG   // power probe (profiles/power_probe.sh): the bare MFMA loop, back to back, for several seconds
    for (int rep = 0; rep < 1500; rep++) run<512, 2, 1, 1, 0, 0, 0, 0, 0, 0>(in, out, "MFMA only (long)");
    return 0;
#endif
    BOTH(512, 2, 1, 0, 0, 0, 0, 0, 0, "MFMA only");
    BOTH(512, 2, 1, 6, 0, 0, 0, 0, 0, "+ 6 v_fma");
    BOTH(512, 2, 1, 15, 0, 0, 0, 0, 0, "+ 15 v_fma");
    BOTH(512, 2, 1, 16, 1, 0, 0, 0, 0, "+ 16 staging VALU");
    BOTH(512, 2, 1, 0, 0, 0, 2, 0, 0, "+ 2 ds_read_b128");
    BOTH(512, 2, 1, 0, 0, 0, 0, 1, 0, "+ 1 atomic");
    BOTH(512, 2, 1, 15, 0, 8, 2, 1, 0, "round-3 SA3 mix (15 VALU 8 SALU 2 reads 1 atomic)");
    BOTH(512, 2, 1, 16, 1, 8, 2, 1, 1, "round-3 SA3 mix, staging VALU + write");
    BOTH(512, 2, 1, 6, 0, 6, 2, 1, 1, "scalarised SA3 mix (6 VALU 6 SALU 2 reads 1 atomic 1 write per 3 MFMA)");
    BOTH(512, 2, 1, 4, 0, 6, 2, 1, 1, "4 VALU");
    BOTH(512, 2, 1, 8, 0, 6, 2, 1, 1, "8 VALU");
    BOTH(512, 2, 1, 10, 0, 6, 2, 1, 1, "10 VALU");
    BOTH(512, 2, 0, 15, 0, 8, 2, 1, 0, "round-3 mix without MFMAs");
    BOTH(256, 1, 1, 0, 0, 0, 0, 0, 0, "1 wave/SIMD, MFMA only");
    BOTH(256, 1, 1, 15, 0, 0, 0, 0, 0, "1 wave/SIMD + 15 v_fma");
    BOTH(256, 1, 1, 6, 0, 6, 2, 1, 1, "1 wave/SIMD scalarised mix");
    // the same bare MFMA loop on ALL-ZERO operands: if the chip were not power-limited the operand values would not matter
    (void)hipMemset(in, 0, 1 << 22);
    BOTH(512, 2, 1, 0, 0, 0, 0, 0, 0, "MFMA only, ZERO operands");
    BOTH(256, 1, 1, 0, 0, 0, 0, 0, 0, "1 wave/SIMD, MFMA only, ZERO operands");
    return 0;
}

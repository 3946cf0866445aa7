// Does a float -> fp16 conversion that overflows leave a sticky mark in TRAPSTS.EXCP on gfx950 (no traps enabled)?
// If yes, a kernel can detect "an activation left fp16's range" with one s_getreg at its end instead of a compare per
// element.  Run on the MI355X box: hipcc --offload-arch=gfx950 -O3 trapsts_probe.hip -o trapsts_probe && ./trapsts_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));

// mode 0: cvt_pkrtz of in-range values, 1: cvt_pkrtz of 1e6, 2: v_cvt_f16_f32 (RNE) of 1e6, 3: fp32 multiply overflow,
// 4: 0 * inf (invalid), 5: v_fma_mix of an out-of-range difference, 6: cvt_pkrtz of a denormal-range value (underflow)
__global__ void probe(const float* in, unsigned* out, float* sink, int mode) {
    const float big = in[0], small = in[1], tiny = in[2];
    unsigned before, after;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_TRAPSTS)" : "=s"(before));
    float r = 0.f;
    if (mode == 0) { fp16x2 h = __builtin_amdgcn_cvt_pkrtz(small, small * 2.f); r = (float)h[0] + (float)h[1]; }
    if (mode == 1) { fp16x2 h = __builtin_amdgcn_cvt_pkrtz(big, small); r = (float)h[0] + (float)h[1]; }
    if (mode == 2) { _Float16 h = (_Float16)big; r = (float)h; }
    if (mode == 3) { r = big * big * big * big * big * big * big; }
    if (mode == 4) { float inf = big * big * big * big * big * big * big; r = inf * 0.f * tiny; }
    if (mode == 6) { fp16x2 h = __builtin_amdgcn_cvt_pkrtz(tiny, tiny); r = (float)h[0]; }
    sink[threadIdx.x] = r;
    asm volatile("s_nop 7\n s_nop 7\n s_getreg_b32 %0, hwreg(HW_REG_TRAPSTS)" : "=s"(after) : "v"(r));
    if (threadIdx.x == 0) { out[2 * mode] = before; out[2 * mode + 1] = after; }
}

int main() {
    float h_in[3] = {1.0e6f, 3.0f, 1.0e-7f};
    float *in, *sink; unsigned* out;
    hipMalloc(&in, sizeof(h_in)); hipMalloc(&sink, 64 * 4); hipMalloc(&out, 64);
    hipMemcpy(in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
    hipMemset(out, 0, 64);
    for (int m = 0; m < 7; m++) hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, in, out, sink, m);
    unsigned h[16]; float hs[64];
    hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
    hipMemcpy(hs, sink, 256, hipMemcpyDeviceToHost);
    const char* names[] = {"cvt_pkrtz in range", "cvt_pkrtz(1e6)", "cvt_f16_f32(1e6)", "fp32 mul overflow", "0*inf", "-", "cvt_pkrtz(1e-7)"};
    for (int m = 0; m < 7; m++)
        printf("mode %d %-22s TRAPSTS before %08x after %08x  excp bits %03x -> %03x\n", m, names[m], h[2 * m], h[2 * m + 1],
               h[2 * m] & 0x1ff, h[2 * m + 1] & 0x1ff);
    printf("last sink[0] = %g\n", hs[0]);
    return 0;
}

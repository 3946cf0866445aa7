#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// each workgroup walks its own objects (table of ROWS rows x 1 KB per object); per batch every thread gathers NL rows' 16 B
template <int NL>
__global__ __launch_bounds__(512, 2) void k(const float* __restrict__ tab, float* __restrict__ out, int rows, int n_obj_per_wg,
                                             int batches_per_obj) {
    const int tid = threadIdx.x, c4 = tid & 63, rg = tid >> 6;
    f32x4 acc = {0, 0, 0, 0};
    unsigned h = tid * 2654435761u;
    for (int o = 0; o < n_obj_per_wg; o++) {
        const float* base = tab + ((size_t)blockIdx.x * n_obj_per_wg + o) * rows * 256;
        for (int b = 0; b < batches_per_obj; b++) {
            f32x4 v[NL];
#pragma unroll
            for (int i = 0; i < NL; i++) {
                h = h * 1664525u + 1013904223u;
                const unsigned r = __builtin_amdgcn_readfirstlane(h >> 8) % (unsigned)rows;  // wave-uniform row: a 1 KB row per wave-load
                v[i] = *(const f32x4*)(base + (size_t)((r + rg * 7 + i) % rows) * 256 + c4 * 4);
            }
#pragma unroll
            for (int i = 0; i < NL; i++) acc += v[i];
            __syncthreads();
        }
    }
    out[blockIdx.x * 512 + tid] = acc[0] + acc[1] + acc[2] + acc[3];
}
template <int NL>
void run(const float* tab, float* out, int rows, int nobj, int bpo, const char* name) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<NL>, dim3(256), dim3(512), 0, 0, tab, out, rows, nobj, bpo); hipDeviceSynchronize();
    hipEventRecord(a); hipLaunchKernelGGL(k<NL>, dim3(256), dim3(512), 0, 0, tab, out, rows, nobj, bpo); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = 256.0 * nobj * bpo * 512 * NL * 16;
    printf("%-40s rows/object %3d  %d loads/thread/batch: %7.3f ms  %6.2f TB/s  %5.1f B/clk/CU @2.4GHz, %6.0f cyc/batch\n", name, rows, NL, ms,
           bytes / ms / 1e9, bytes / ms / 1e-3 / 256 / 2.4e9, ms * 1e-3 * 2.4e9 / (nobj * bpo));
}
int main() {
    const size_t bytes = (size_t)3 << 30;
    float *tab, *out; hipMalloc(&tab, bytes); hipMalloc(&out, 1 << 22); hipMemset(tab, 0, bytes);
    // SA3-like: 64 rows x 1 KB per object (A) ; 13 batches per object, 8 loads per thread per batch (4 A + 4 B)
    run<8>(tab, out, 64, 128, 13, "SA3-like (64 KB table, 13 batches)");
    run<4>(tab, out, 64, 128, 13, "half the loads");
    run<8>(tab, out, 64, 128, 1, "streaming (1 batch per object)");
    run<8>(tab, out, 64, 16, 104, "hot (104 batches per object)");
    run<16>(tab, out, 64, 128, 13, "16 loads in flight");
    return 0;
}

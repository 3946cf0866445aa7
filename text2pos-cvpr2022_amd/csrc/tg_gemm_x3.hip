// LDS-tiled f16x3 GEMM with fused bias / ReLU epilogue for the big dense layers behind the PointNet++ trunk:
//   lin1 (1024 -> 512), lin2 (512 -> 256)   models/pointcloud/pointnet2.py:89-90
//   mlp_merge (3D -> D)                       models/object_encoder.py:137-138
// C[M,N] = act(A[M,K] W[K,N] + bias) with A fp32 in HBM and W given as the scaled split image of
// packing.py::pack_gemm_x3: w' = s w (s a power of two), hi = fp16(w'), lo = fp16(w' - hi), stored [plane][n][k]
// (k contiguous) so that a 16-byte load is one MFMA B-operand fragment.  A is split on the fly (hi = fp16 to nearest,
// lo = fp16(a - hi)); hi.hi + hi.lo + lo.hi share one fp32 accumulator (the matrix cores honour fp16 denormals), the
// epilogue divides by s.  Same error class as an fp32 fma chain (~5e-7), 16/3 x the fp32-MFMA rate.
// 128 x 128 x 32 tile (64 x 64 x 32 for calls of few rows), 4 waves x (2 x 2 | 1) v_mfma_f32_32x32x16_f16 blocks, double-buffered LDS,
// register-prefetched loads.
#include "t2p_common.h"

namespace t2p {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));

constexpr int BK = 32;
constexpr int LDT = BK + 8;               // halves per LDS row: 80 B keeps ds_read_b128 conflict-free
// TS = tile side (BM = BN): 128 (4 waves x 2 x 2 MFMA blocks) or 64 (4 waves x 1 block) - the small tile for calls with so few rows
// that 128 x 128 tiles would leave most of the chip idle (a 64-cell call has ~1,000 object rows: 8 x 2..4 tiles on 256 CUs, each
// walking K alone: 32 us per head layer; the reference's callers use batch_size 64).  Every output element sees the same
// products in the same order with either tile: bit-identical results.
template <int TS>
struct X3Cfg {
    static constexpr int PLANE = TS * LDT;          // halves per plane of one operand tile
    static constexpr size_t kLds = (size_t)2 /*buffers*/ * 2 /*A, W*/ * 2 /*hi, lo*/ * PLANE * sizeof(_Float16);
    static constexpr int TPR = 256 / TS;            // threads per staged row (2 / 4)
    static constexpr int KPT = BK / TPR;            // k per thread and tile (16 / 8)
    static constexpr int MI = TS / 64;              // 32 x 32 blocks per wave and dimension (2 / 1)
};

template <int SEL>
__device__ __forceinline__ float sub_half(float v, fp16x2 h) {
    float r;
    if constexpr (SEL == 0)
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(v));
    else
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(v));
    return r;
}

template <int TS>
__global__ __launch_bounds__(256, 2) void k_gemm_x3(const float* __restrict__ A, int lda, const _Float16* __restrict__ Wx,
                                                    int kp /*padded K of the image*/, float inv_scale,
                                                    const float* __restrict__ bias, float* C, int ldc, int c0, int64_t M,
                                                    int K, int N, int relu, const float* R, int ldr, uint32_t* amax_in) {
    using Cf = X3Cfg<TS>;
    constexpr int BM = TS, BN = TS, PLANE = Cf::PLANE, KPT = Cf::KPT, MI = Cf::MI;
    extern __shared__ __attribute__((aligned(16))) _Float16 sm[];
    // buffer b: [A hi | A lo | W hi | W lo], each [TS][LDT]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, h = lane >> 5, l31 = lane & 31;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const _Float16* Whi = Wx;
    const _Float16* Wlo = Wx + (size_t)N * kp;

    // staging assignment: A: thread -> row tid/2, 16 consecutive k (4 x f32x4); W: thread -> column n = tid/2,
    // 16 consecutive k of each plane (2 x 16 B per plane)
    const int s_row = tid / Cf::TPR, s_k = (tid % Cf::TPR) * KPT;
    const bool a_ok = (m0 + s_row) < M;
    const bool w_ok = (n0 + s_row) < N;
    const float* a_ptr = A + (m0 + s_row) * (int64_t)lda + s_k;
    const _Float16* wh_ptr = Whi + (size_t)(n0 + s_row) * kp + s_k;
    const _Float16* wl_ptr = Wlo + (size_t)(n0 + s_row) * kp + s_k;
    f32x4 ra[KPT / 4];
    uint4 rwh[KPT / 8], rwl[KPT / 8];
    float gmax = 0.f;  // fp16-range guard: largest |a| split to fp16 by this thread

    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < KPT / 4; i++) {
            const int k = k0 + s_k + 4 * i;
            ra[i] = (a_ok && k < K) ? *(const f32x4*)(a_ptr + k0 + 4 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < KPT / 8; i++) {
            rwh[i] = w_ok ? *(const uint4*)(wh_ptr + k0 + 8 * i) : uint4{0, 0, 0, 0};   // the image is zero-padded in k
            rwl[i] = w_ok ? *(const uint4*)(wl_ptr + k0 + 8 * i) : uint4{0, 0, 0, 0};
        }
    };
    auto store_tile = [&](int buf) {
        _Float16* base = sm + (size_t)buf * 4 * PLANE;
#pragma unroll
        for (int i = 0; i < KPT / 4; i++) {
            const f32x4 v = ra[i];
            gmax = fmaxf(fmaxf(gmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
            const fp16x2 h01 = cvt_pk_f16(v[0], v[1]), h23 = cvt_pk_f16(v[2], v[3]);
            const fp16x2 l01 = cvt_pk_f16(sub_half<0>(v[0], h01), sub_half<1>(v[1], h01));
            const fp16x2 l23 = cvt_pk_f16(sub_half<0>(v[2], h23), sub_half<1>(v[3], h23));
            uint2 ph, pl;
            ph.x = __builtin_bit_cast(uint32_t, h01); ph.y = __builtin_bit_cast(uint32_t, h23);
            pl.x = __builtin_bit_cast(uint32_t, l01); pl.y = __builtin_bit_cast(uint32_t, l23);
            *(uint2*)(base + s_row * LDT + s_k + 4 * i) = ph;
            *(uint2*)(base + PLANE + s_row * LDT + s_k + 4 * i) = pl;
        }
#pragma unroll
        for (int i = 0; i < KPT / 8; i++) {
            *(uint4*)(base + 2 * PLANE + s_row * LDT + s_k + 8 * i) = rwh[i];
            *(uint4*)(base + 3 * PLANE + s_row * LDT + s_k + 8 * i) = rwl[i];
        }
    };

    f32x16 acc[MI][MI];
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < MI; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    load_tile(0);
    store_tile(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < K; k0 += BK, buf ^= 1) {
        const bool more = k0 + BK < K;
        if (more) load_tile(k0 + BK);
        const _Float16* base = sm + (size_t)buf * 4 * PLANE;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ks++) {
            half8 a_hi[MI], a_lo[MI], b_hi[MI], b_lo[MI];
#pragma unroll
            for (int i = 0; i < MI; i++) {
                const _Float16* p = base + (wr * (TS / 2) + i * 32 + l31) * LDT + ks * 16 + h * 8;
                a_hi[i] = *(const half8*)p;
                a_lo[i] = *(const half8*)(p + PLANE);
                const _Float16* q = base + 2 * PLANE + (wc * (TS / 2) + i * 32 + l31) * LDT + ks * 16 + h * 8;
                b_hi[i] = *(const half8*)q;
                b_lo[i] = *(const half8*)(q + PLANE);
            }
#pragma unroll
            for (int i = 0; i < MI; i++)
#pragma unroll
                for (int j = 0; j < MI; j++) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi[i], b_hi[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi[i], b_lo[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo[i], b_hi[j], acc[i][j], 0, 0, 0);
                }
        }
        if (more) store_tile(buf ^ 1);
        __syncthreads();
    }

    if (blockIdx.y == 0) guard_publish(amax_in, gmax);  // every column block stages the same rows
#pragma unroll
    for (int j = 0; j < MI; j++) {
        const int col = n0 + wc * (TS / 2) + j * 32 + l31;
        if (col >= N) continue;
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < MI; i++) {
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int64_t row = m0 + wr * (TS / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (row < M) {
                    float v = fmaf(acc[i][j][e], inv_scale, bv);
                    if (relu) v = fmaxf(v, 0.f);
                    if (R) v += R[row * (int64_t)ldr + col];
                    C[row * (int64_t)ldc + c0 + col] = v;
                }
            }
        }
    }
}

}  // namespace

// Wx: image of pack_gemm_x3 ([2][N][kp] fp16, kp = K rounded up to 32, zero padded), scale = its power-of-two factor.
int launch_gemm_x3(const float* A, int lda, const void* Wx, float scale, const float* bias, float* C, int ldc, int c0,
                   int64_t M, int K, int N, int relu, hipStream_t st, const float* resid, int ldr, uint32_t* amax_in) {
    T2P_CHECK_ARG(K % 4 == 0 && N % 8 == 0 && lda % 4 == 0 && scale > 0.f, "gemm_x3: K=%d %% 4, N=%d %% 8, lda=%d %% 4", K, N, lda);
    T2P_CHECK_ARG((((uintptr_t)A) & 15) == 0 && (((uintptr_t)Wx) & 15) == 0, "gemm_x3: A and W must be 16-byte aligned");
    if (M == 0) return 0;
    const int kp = (K + 31) / 32 * 32;
    ProfScope ps_("tg_gemm_x3", st);
    const int64_t tiles128 = ((M + 127) / 128) * ((N + 127) / 128);
    if (tiles128 * 2 <= num_cus()) {                      // too few big tiles to fill the chip: 64 x 64 tiles (same bits out)
        T2P_TRY(reserve_lds((const void*)k_gemm_x3<64>, X3Cfg<64>::kLds, "gemm_x3"));
        dim3 grid((unsigned)((M + 63) / 64), (unsigned)((N + 63) / 64));
        hipLaunchKernelGGL(k_gemm_x3<64>, grid, dim3(256), X3Cfg<64>::kLds, st, A, lda, (const _Float16*)Wx, kp, 1.0f / scale, bias, C, ldc,
                           c0, M, K, N, relu, resid, ldr, amax_in);
    } else {
        T2P_TRY(reserve_lds((const void*)k_gemm_x3<128>, X3Cfg<128>::kLds, "gemm_x3"));
        dim3 grid((unsigned)((M + 127) / 128), (unsigned)((N + 127) / 128));
        hipLaunchKernelGGL(k_gemm_x3<128>, grid, dim3(256), X3Cfg<128>::kLds, st, A, lda, (const _Float16*)Wx, kp, 1.0f / scale, bias, C, ldc,
                           c0, M, K, N, relu, resid, ldr, amax_in);
    }
    T2P_CHECK_LAUNCH("gemm_x3");
    return 0;
}

}  // namespace t2p

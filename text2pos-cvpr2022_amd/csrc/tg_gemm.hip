// Generic LDS-tiled fp32-MFMA GEMM with fused bias / ReLU epilogue, for the small dense layers of the path:
//   lin1, lin2                    models/pointcloud/pointnet2.py:89-90
//   mlp_pointnet, color/pos MLPs, mlp_merge   models/object_encoder.py:98,124-138
//   DynamicEdgeConv layer-1 tables (P, Q) and the cell `lin` MLP   models/cell_retrieval.py:46-49,97-99
//   LSTM input projection of the vocabulary (gate table)            models/modules.py:77,89
// C[M,N] = act(A[M,K] W[K,N] + bias).  128x128x16 tile (64x64x16 for products of few rows), 4 waves x (2x2 | 1) v_mfma_f32_32x32x2_f32 tiles,
// register-prefetched global loads (issue next chunk before the MFMAs of the current one).
#include "t2p_common.h"

namespace t2p {
namespace {

constexpr int BK = 16;
// TS = tile side (BM = BN): 128, or 64 for products of so few rows that the big tiles would leave most of the chip idle (head
// layers of a 64-cell call, the training path's per-object layers); every output element sees the same fma chain either way.
template <int TS>
__global__ __launch_bounds__(256) void k_gemm(const float* __restrict__ A, int lda, const float* __restrict__ W,
                                              const float* __restrict__ bias, float* C, int ldc, int c0,
                                              int64_t M, int K, int N, int relu, const float* R, int ldr) {
    constexpr int BM = TS, BN = TS, MI = TS / 64;
    constexpr int LDA_S = BM + 2;  // As[k][m], +2 keeps the transposing ds_write_b32 conflict-free
    constexpr int LDB_S = BN + 4;  // Ws[k][n]
    constexpr int TPR = 256 / TS;  // threads per staged row of A (2 / 4): KA = 16 / TPR consecutive k each
    constexpr int KA = BK / TPR;   // 8 / 4
    constexpr int NW = TS / 16;    // consecutive n per thread of one k row of W (8 / 4)
    __shared__ float As[BK * LDA_S];
    __shared__ __attribute__((aligned(16))) float Ws[BK * LDB_S];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, h = lane >> 5, l31 = lane & 31;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    // global -> register staging assignments
    const int a_row = tid / TPR, a_k = (tid % TPR) * KA;  // KA consecutive k of one row
    const int w_k = tid >> 4, w_n = (tid & 15) * NW;      // NW consecutive n of one k
    const bool a_row_ok = (m0 + a_row) < M;
    const float* a_ptr = A + (m0 + a_row) * (int64_t)lda + a_k;
    f32x4 ra[KA / 4], rw[NW / 4];

    auto load_chunk = [&](int k0) {
#pragma unroll
        for (int i = 0; i < KA / 4; i++) {
            const int k = k0 + a_k + 4 * i;
            ra[i] = (a_row_ok && k < K) ? *(const f32x4*)(a_ptr + k0 + 4 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < NW / 4; i++) {
            const int kk = k0 + w_k, n = n0 + w_n + 4 * i;
            rw[i] = (kk < K && n < N) ? *(const f32x4*)(W + (int64_t)kk * N + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < KA / 4; i++)
#pragma unroll
            for (int e = 0; e < 4; e++) As[(a_k + 4 * i + e) * LDA_S + a_row] = ra[i][e];
#pragma unroll
        for (int i = 0; i < NW / 4; i++) *(f32x4*)(Ws + w_k * LDB_S + w_n + 4 * i) = rw[i];
    };

    f32x16 acc[MI][MI];
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < MI; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    load_chunk(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        store_chunk();
        __syncthreads();
        if (k0 + BK < K) load_chunk(k0 + BK);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a[MI], b[MI];
#pragma unroll
            for (int i = 0; i < MI; i++) a[i] = As[(kk + h) * LDA_S + wr * (TS / 2) + i * 32 + l31];
#pragma unroll
            for (int j = 0; j < MI; j++) b[j] = Ws[(kk + h) * LDB_S + wc * (TS / 2) + j * 32 + l31];
#pragma unroll
            for (int i = 0; i < MI; i++)
#pragma unroll
                for (int j = 0; j < MI; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

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
                    float v = acc[i][j][e] + bv;
                    if (relu) v = fmaxf(v, 0.f);
                    if (R) v += R[row * (int64_t)ldr + col];  // residual (may alias C: same element, same thread)
                    C[row * (int64_t)ldc + c0 + col] = v;
                }
            }
        }
    }
}

// ---- skinny product: few rows, long K -------------------------------------------------------------------------------------
// The per-step products of the training-mode LSTM (h_{s-1} W_hh^T: 64 x 256 x 1024; d_pre W_hh: 64 x 1024 x 256) put 2-8
// workgroups of the tiled kernel above on the chip, each walking its whole K alone behind two barriers per 16 k: 31-47 us per
// launch, 2 x T launches per direction.  Here a workgroup of 16 waves owns 64 rows x 32 columns: wave (rt, ks) multiplies row
// tile rt by the ks-th eighth of K straight from global memory (no LDS staging: every operand is used once per wave), the eight
// partial tiles are added in a fixed order through LDS by the ks = 0 waves.  k is permuted inside groups of 8 (lane half h owns
// k0 + 4 h .. + 3, so that A arrives in 16-byte loads): fp32 sums in another order than k_gemm, deterministic.
__global__ __launch_bounds__(1024) void k_gemm_skinny(const float* __restrict__ A, int lda, const float* __restrict__ W,
                                                       float* __restrict__ C, int ldc, int64_t M, int K, int N) {
    __shared__ float part[14 * 16 * 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l31 = lane & 31;
    const int rt = wave & 1, ks = wave >> 1;
    const int col = blockIdx.x * 32 + l31;
    const int64_t m0 = (int64_t)blockIdx.y * 64 + rt * 32;
    const int64_t row = m0 + l31;
    const int per = (((K + 7) / 8 + 7) / 8) * 8;
    const int k_begin = ks * per < K ? ks * per : K;
    const int k_end = k_begin + per < K ? k_begin + per : K;
    const bool row_ok = row < M, col_ok = col < N;
    const float* ap = A + (row_ok ? row : 0) * (int64_t)lda + 4 * h;
    const float* wp = W + (int64_t)(4 * h) * N + (col_ok ? col : 0);
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = 0.f;
    if (m0 < M) {
#pragma unroll 4
        for (int k = k_begin; k < k_end; k += 8) {
            const bool k_ok = k + 4 * h < K;          // (K = 8 i + 4: the upper half of the last group is past the end)
            const int ks_ = k_ok ? k : 0;
            f32x4 a = *(const f32x4*)(ap + ks_);
            float b[4];
#pragma unroll
            for (int j = 0; j < 4; j++) b[j] = wp[(int64_t)(ks_ + j) * N];
            const bool a_ok = row_ok && k_ok, b_ok = col_ok && k_ok;
#pragma unroll
            for (int j = 0; j < 4; j++)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_ok ? a[j] : 0.f, b_ok ? b[j] : 0.f, acc, 0, 0, 0);
        }
    }
    if (ks > 0) {
#pragma unroll
        for (int e = 0; e < 16; e++) part[(((ks - 1) * 2 + rt) * 16 + e) * 64 + lane] = acc[e];
    }
    __syncthreads();
    if (ks == 0 && m0 < M) {
#pragma unroll
        for (int s = 1; s < 8; s++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[e] += part[(((s - 1) * 2 + rt) * 16 + e) * 64 + lane];
        if (col_ok) {
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int64_t r = m0 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (r < M) C[r * (int64_t)ldc + col] = acc[e];
            }
        }
    }
}

}  // namespace

int launch_gemm_skinny(const float* A, int lda, const float* W, float* C, int ldc, int64_t M, int K, int N, hipStream_t st) {
    T2P_CHECK_ARG(K % 4 == 0 && lda % 4 == 0 && (((uintptr_t)A) & 15) == 0, "gemm_skinny: K=%d and lda=%d must be multiples of 4, A 16-byte aligned",
                  K, lda);
    if (M == 0 || N == 0) return 0;
    ProfScope ps_("tg_gemm_skinny", st);
    hipLaunchKernelGGL(k_gemm_skinny, dim3((unsigned)((N + 31) / 32), (unsigned)((M + 63) / 64)), dim3(1024), 0, st, A, lda, W, C, ldc,
                       M, K, N);
    T2P_CHECK_LAUNCH("gemm_skinny");
    return 0;
}

int launch_gemm(const float* A, int lda, const float* W, const float* bias, float* C, int ldc, int c0, int64_t M,
                int K, int N, int relu, hipStream_t st, const float* resid, int ldr) {
    T2P_CHECK_ARG(K % 4 == 0 && N % 8 == 0 && lda % 4 == 0, "gemm: K=%d must be a multiple of 4, N=%d of 8, lda=%d of 4",
                  K, N, lda);
    T2P_CHECK_ARG((((uintptr_t)A) & 15) == 0 && (((uintptr_t)W) & 15) == 0, "gemm: A and W must be 16-byte aligned");
    if (M == 0) return 0;
    ProfScope ps_("tg_gemm", st);
    const int64_t tiles128 = ((M + 127) / 128) * ((N + 127) / 128);
    if (tiles128 * 2 <= num_cus()) {                      // too few big tiles to fill the chip: 64 x 64 tiles (same bits out)
        dim3 grid((unsigned)((M + 63) / 64), (unsigned)((N + 63) / 64));
        hipLaunchKernelGGL(k_gemm<64>, grid, dim3(256), 0, st, A, lda, W, bias, C, ldc, c0, M, K, N, relu, resid, ldr);
    } else {
        dim3 grid((unsigned)((M + 127) / 128), (unsigned)((N + 127) / 128));
        hipLaunchKernelGGL(k_gemm<128>, grid, dim3(256), 0, st, A, lda, W, bias, C, ldc, c0, M, K, N, relu, resid, ldr);
    }
    T2P_CHECK_LAUNCH("gemm");
    return 0;
}

}  // namespace t2p

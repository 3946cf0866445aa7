// C[K1, N] = A[M, K1]^T  B[M, N]: the weight-gradient / one-hot / embedding-gradient products of the training-mode path
// (dW = dY^T X of every Linear, d W_hh = H^T dPre, the gate-table gradient; training/coarse.py:31-62 through autograd).
// The reduction runs over the ROWS (M = edges or points of a batch: 1e4..1e6) while the output is small (<= 1024 x 1024), so
// the rows are split over the grid: block (tile, s) forms the partial product of row range s for one 64 x 64 output tile,
// a second kernel adds the partials in a fixed order (deterministic: no float atomics).
//
// v_mfma_f32_32x32x2_f32 takes A[i][kk] and B[kk][j] with kk = lane >> 5: for this product that is A[m0 + kk][k0 + i] and
// B[m0 + kk][n0 + j], i.e. each half-wave reads 32 CONSECUTIVE floats of one row of A and of B - coalesced 128-byte reads
// straight from global memory, no LDS transposition.  Four waves per block, one 32 x 32 accumulator each; the four waves
// re-read the block's two row slices through L1.  fp32 MFMA: exact fp32 fma chains.
#include "t2p_common.h"

namespace t2p {
namespace {

constexpr int TN_ROWS = 16;   // rows (m pairs x 2) fetched ahead per step: 8 MFMAs per wave between waits

__global__ __launch_bounds__(256) void k_gemm_tn(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                 float* __restrict__ part, int64_t M, int K1, int N, int tiles_n, int splits) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    const int tile = blockIdx.x, s = blockIdx.y;
    const int k0 = (tile / tiles_n) * 64 + (wave >> 1) * 32, n0 = (tile % tiles_n) * 64 + (wave & 1) * 32;
    // row range of this split: multiples of TN_ROWS
    const int64_t per = ((M + splits - 1) / splits + TN_ROWS - 1) / TN_ROWS * TN_ROWS;
    const int64_t m_lo = (int64_t)s * per, m_hi = (m_lo + per) < M ? (m_lo + per) : M;
    const bool a_ok = (k0 + l31) < K1, b_ok = (n0 + l31) < N;
    const float* ap = A + (k0 + l31);
    const float* bp = B + (n0 + l31);
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = 0.f;
    for (int64_t m = m_lo; m < m_hi; m += TN_ROWS) {
        float a[TN_ROWS / 2], b[TN_ROWS / 2];
#pragma unroll
        for (int i = 0; i < TN_ROWS / 2; i++) {
            const int64_t r = m + 2 * i + h;
            const bool ok = r < m_hi;
            a[i] = (ok && a_ok) ? ap[r * lda] : 0.f;
            b[i] = (ok && b_ok) ? bp[r * ldb] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < TN_ROWS / 2; i++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc, 0, 0, 0);
    }
    // partial [s][K1][N]
    float* out = part + (int64_t)s * K1 * N;
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const int k = k0 + (e & 3) + 8 * (e >> 2) + 4 * h, n = n0 + l31;
        if (k < K1 && n < N) out[(int64_t)k * N + n] = acc[e];
    }
}

__global__ __launch_bounds__(256) void k_gemm_tn_reduce(const float* __restrict__ part, float* __restrict__ C, int ldc,
                                                        int K1, int N, int splits) {
    const int64_t total = (int64_t)K1 * N;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        float sum = 0.f;
        for (int s = 0; s < splits; s++) sum += part[(int64_t)s * total + i];   // fixed order
        C[(i / N) * ldc + i % N] = sum;
    }
}

int tn_splits(int64_t M, int K1, int N) {
    const int tiles = ((K1 + 63) / 64) * ((N + 63) / 64);
    int64_t want = (4 * (int64_t)num_cus() + tiles - 1) / tiles;   // ~4 blocks per CU in total
    const int64_t max_by_rows = (M + 4 * TN_ROWS - 1) / (4 * TN_ROWS);
    if (want > max_by_rows) want = max_by_rows;
    if (want > 4096) want = 4096;
    return want < 1 ? 1 : (int)want;
}

}  // namespace

size_t gemm_tn_workspace_bytes(int64_t M, int K1, int N) { return (size_t)tn_splits(M, K1, N) * K1 * N * sizeof(float) + 256; }

int launch_gemm_tn(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int64_t M, int K1, int N, void* ws,
                   size_t ws_bytes, hipStream_t st) {
    T2P_CHECK_ARG(A && B && C && M >= 0 && K1 >= 1 && N >= 1 && lda >= K1 && ldb >= N && ldc >= N, "gemm_tn: bad arguments");
    const int splits = tn_splits(M, K1, N);
    if (ws == nullptr || ws_bytes < gemm_tn_workspace_bytes(M, K1, N)) {
        set_error("gemm_tn: workspace %zu B < required %zu B", ws_bytes, gemm_tn_workspace_bytes(M, K1, N));
        return T2P_E_WORKSPACE;
    }
    const int tiles_n = (N + 63) / 64, tiles = ((K1 + 63) / 64) * tiles_n;
    {
        ProfScope ps_("gemm_tn", st);
        hipLaunchKernelGGL(k_gemm_tn, dim3(tiles, splits), dim3(256), 0, st, A, lda, B, ldb, (float*)ws, M, K1, N, tiles_n, splits);
        T2P_CHECK_LAUNCH("gemm_tn");
    }
    const int64_t total = (int64_t)K1 * N;
    const unsigned grid = (unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_gemm_tn_reduce, dim3(grid), dim3(256), 0, st, (const float*)ws, C, ldc, K1, N, splits);
    T2P_CHECK_LAUNCH("gemm_tn_reduce");
    return 0;
}

}  // namespace t2p

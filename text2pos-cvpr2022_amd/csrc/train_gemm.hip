// Weight and bias gradient of every nn.Linear in the training-mode path (`get_mlp` blocks under model.train(),
// models/modules.py:11-36, reached from training/coarse.py:31-62 through autograd), exact fp32 (v_mfma_f32_32x32x2_f32):
//
//   k_wgrad_f32  dW[K1][N] = dY[M][K1]^T X[M][N], db[K1] = column sums of dY
//
// M is the number of EDGES of a batch (1e5 .. 1.4e6) while K1 and N are layer widths (8 .. 1024).  The contraction runs over the
// rows, so both operands are read as "row m + h, 32 consecutive columns" straight from a row-major LDS tile - no transposition
// anywhere.  A workgroup owns an output block of up to 256 x 256 (8 x 8 tiles of 32 x 32, up to 8 per wave = 128 accumulator
// registers) for one row range, so dY and X are read ONCE per row range instead of once per 64 x 64 output tile (k_gemm_tn of
// tg_gemm_tn.hip: 4 x 4 times for a 256 x 256 gradient); narrow operands are staged hundreds of rows at a time (a barrier per 32
// rows would be all the kernel does).  The row ranges' partial blocks are added in a fixed order by a second kernel
// (deterministic, no float atomics).  The column sums of dY (the bias gradient: an `aten::sum` over the whole tensor before) ride
// along in the lanes that hold dY anyway.
// Measured on the 64-cell step's shapes against t2p_gemm_tn + sum (profiles/microbench/train_gemm_shapes.py): see the notebook.
// (A weight-stationary fp32 kernel for the forward / input-gradient products was built beside it and dropped: the tiled k_gemm
// already runs the large layers at 95-100 TFLOP/s - 62 % of the fp32 matrix peak - and the new kernel only matched that.)
#include "t2p_common.h"

namespace t2p {
namespace {

// ---- weight gradient ------------------------------------------------------------------------------------------------------

template <int TPW>
__global__ __launch_bounds__(512) void k_wgrad_f32(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                   float* __restrict__ part, float* __restrict__ psum, int64_t M, int K1, int N,
                                                   int n_blocks_n, int64_t rows_per_split, int n_phase, int ktp, int ntp, int rows_chunk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [2][rows_chunk][kw + nw]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l31 = lane & 31;
    const int kb = (int)blockIdx.y / n_blocks_n, nb = (int)blockIdx.y % n_blocks_n;
    const int k_base = kb * 256, n_base = nb * 256;
    const int kw = (K1 - k_base) < 256 ? (K1 - k_base) : 256, nw = (N - n_base) < 256 ? (N - n_base) : 256;   // block extent
    const int kw4 = (kw + 3) & ~3, nw4 = (nw + 3) & ~3;
    const int ldl = kw4 + nw4;
    const int kt_n = (kw + 31) / 32, nt_n = (nw + 31) / 32, n_tiles = kt_n * nt_n;
    const int64_t m_lo = (int64_t)blockIdx.x * rows_per_split;
    const int64_t m_hi = (m_lo + rows_per_split) < M ? (m_lo + rows_per_split) : M;

    // Tiles of this wave.  The eight waves form ktp row groups (ktp = 1, 2, 4, 8 >= the block's tile rows): wave w works in tile
    // row kt = w % ktp, so ONE read of dY per row pair serves all its tiles; the G = 8 / ktp waves of a tile row share its nt_n
    // tiles: wave g of them takes nt = g, g + G, ...  With fewer tile columns than waves (ntp = pow2 >= nt_n < G) the surplus
    // waves split the row pairs instead: phase = g / ntp of n_phase = G / ntp (their partial blocks are separate summands).
    const int kt = wave % ktp, grp = wave / ktp, G = 8 / ktp;
    const int phase = (n_phase > 1) ? grp / ntp : 0;
    const bool a_live = kt < kt_n;
    int boff[TPW];                                        // LDS column of tile i's X operand (0 for a tile that does not exist)
    int tn[TPW];
    bool live[TPW];
#pragma unroll
    for (int i = 0; i < TPW; i++) {
        const int nt = (n_phase > 1) ? grp % ntp : grp + G * i;
        live[i] = a_live && nt < nt_n && (n_phase == 1 || i == 0);
        tn[i] = live[i] ? nt : 0;
        boff[i] = kw4 + tn[i] * 32 + l31;
    }
    const int aoff = (a_live ? kt : 0) * 32 + l31;
    const bool sums_here = a_live && tn[0] == 0 && live[0];   // this wave holds tile (kt, 0): it adds dY's column sums of its rows
    (void)n_tiles;
    f32x16 acc[TPW];
#pragma unroll
    for (int i = 0; i < TPW; i++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[i][e] = 0.f;
    float colsum = 0.f;

    // staging plan of this thread: up to 8 float4 pieces of a chunk; (row in chunk, source column) do not change from chunk to
    // chunk and are kept as ONE packed word per piece (the accumulators leave no room for more):
    //   bits 0-9 row in chunk | 10-16 float4 column inside the operand | 17-19 live elements (0 = no piece, 4 = whole) | 20 operand X
    const int pieces_row = ldl / 4;                       // float4 pieces per staged row
    const int n_pieces = rows_chunk * pieces_row;         // <= 4096 (the host's choice of rows_chunk)
    constexpr int MAXP = 8;
    uint32_t plan[MAXP];
#pragma unroll
    for (int p = 0; p < MAXP; p++) {
        const int idx = p * 512 + tid;
        plan[p] = 0u;
        if (idx < n_pieces) {
            const int r = idx / pieces_row, q = idx % pieces_row;
            if (q * 4 < kw4) {                            // (a piece never straddles the two operands: kw4 is a multiple of 4)
                const int cnt = (kw - q * 4) < 4 ? (kw - q * 4) : 4;
                plan[p] = (uint32_t)r | ((uint32_t)q << 10) | ((uint32_t)cnt << 17);
            } else {
                const int qb = q - kw4 / 4;
                const int cnt = (nw - qb * 4) < 4 ? (nw - qb * 4) : 4;
                plan[p] = (uint32_t)r | ((uint32_t)qb << 10) | ((uint32_t)cnt << 17) | (1u << 20);
            }
        }
    }
    const float* const a_blk = A + k_base;
    const float* const b_blk = B + n_base;
    f32x4 stage[MAXP];
    auto load = [&](int64_t m) {
#pragma unroll
        for (int p = 0; p < MAXP; p++) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            const uint32_t w = plan[p];
            const int cnt = (int)((w >> 17) & 7u);
            const int64_t row = m + (int)(w & 1023u);
            if (cnt > 0 && row < m_hi) {
                const bool is_b = (w >> 20) & 1u;
                const float* src = (is_b ? b_blk + row * (int64_t)ldb : a_blk + row * (int64_t)lda) + ((w >> 10) & 127u) * 4;
                if (cnt == 4) v = *(const f32x4*)src;
                else
                    for (int e = 0; e < 4; e++) v[e] = e < cnt ? src[e] : 0.f;   // tail columns of a block no multiple of 4 wide
            }
            stage[p] = v;
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int p = 0; p < MAXP; p++) {
            const int idx = p * 512 + tid;
            if (idx < n_pieces) *(f32x4*)(lds + buf * rows_chunk * ldl + idx * 4) = stage[p];
        }
    };

    int buf = 0;
    if (m_lo < m_hi) load(m_lo);
    for (int64_t m = m_lo; m < m_hi; m += rows_chunk) {
        store(buf);
        __syncthreads();
        if (m + rows_chunk < m_hi) load(m + rows_chunk);
        const float* t0 = lds + buf * rows_chunk * ldl;
        // one row pair: the dY operand once, the wave's X operands, then the MFMAs (the first form of this loop read two operands,
        // waited and multiplied, tile by tile behind a branch each: 41 % of the fp32 matrix rate)
        if (a_live) {                                         // (uniform)
            // one row pair: the dY operand once, the wave's X operands, then the MFMAs.  (The first form of this loop read two
            // operands, waited and multiplied, tile by tile behind a branch each: 41 % of the fp32 matrix rate; reading the next
            // pair's operands ahead by hand on top of this form changed nothing.)
            for (int rp = phase; rp < rows_chunk / 2; rp += n_phase) {
                const float* rowp = t0 + (2 * rp + h) * ldl;
                const float a = rowp[aoff];               // columns past kw / nw of a tile hold zeros or the other operand:
                float b[TPW];                             // masked at the write-out
#pragma unroll
                for (int i = 0; i < TPW; i++) b[i] = rowp[boff[i]];
                // (a tile slot past the block's last tile column multiplies the operand of column 0 into an accumulator that
                // is never written out: no branch in this loop - with one around each MFMA hipcc spilled the accumulators)
#pragma unroll
                for (int i = 0; i < TPW; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[i], acc[i], 0, 0, 0);
                if (sums_here) colsum += a;
            }
        }
        buf ^= 1;
    }
    // partial block of (split, phase)
    const int64_t slot = (int64_t)blockIdx.x * n_phase + phase;
    float* out = part + slot * (int64_t)K1 * N;
#pragma unroll
    for (int i = 0; i < TPW; i++) {
        if (!live[i]) continue;
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int kl = kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h, nl = tn[i] * 32 + l31;
            if (kl < kw && nl < nw) out[(int64_t)(k_base + kl) * N + n_base + nl] = acc[i][e];
        }
    }
    if (psum != nullptr && nb == 0 && sums_here) {
        const int kl = kt * 32 + l31;
        if (kl < kw) psum[(slot * 2 + h) * (int64_t)K1 + k_base + kl] = colsum;
    }
}

__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ part, const float* __restrict__ psum, float* __restrict__ C,
                                                      int ldc, float* __restrict__ colsum, int K1, int N, int slots) {
    const int64_t total = (int64_t)K1 * N;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total + K1; i += (int64_t)gridDim.x * blockDim.x) {
        float sum = 0.f;
        if (i < total) {
            for (int s = 0; s < slots; s++) sum += part[(int64_t)s * total + i];          // fixed order
            C[(i / N) * ldc + i % N] = sum;
        } else if (colsum != nullptr) {
            const int64_t k = i - total;
            for (int s = 0; s < 2 * slots; s++) sum += psum[(int64_t)s * K1 + k];
            colsum[k] = sum;
        }
    }
}

// small outputs (a 32 x 8 gradient has 288 sums of 2,048 partials each: one thread per sum is a chain of dependent adds behind
// strided loads): one WAVE per output element, lane l adds the slots l, l + 64, ... in order, a fixed butterfly adds the lanes
__global__ __launch_bounds__(256) void k_wgrad_reduce_wave(const float* __restrict__ part, const float* __restrict__ psum,
                                                           float* __restrict__ C, int ldc, float* __restrict__ colsum, int K1, int N,
                                                           int slots) {
    const int64_t total = (int64_t)K1 * N;
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= total + K1) return;
    float sum = 0.f;
    if (i < total) {
        for (int s = lane; s < slots; s += 64) sum += part[(int64_t)s * total + i];
    } else {
        if (colsum == nullptr) return;
        for (int s = lane; s < 2 * slots; s += 64) sum += psum[(int64_t)s * K1 + (i - total)];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if (lane == 0) {
        if (i < total) C[(i / N) * ldc + i % N] = sum;
        else colsum[i - total] = sum;
    }
}

struct WgradPlan {
    int blocks_k, blocks_n, tpw, n_phase, ktp, ntp, splits, rows_chunk;
    int64_t rows_per_split;
    size_t lds_bytes;
};

WgradPlan wgrad_plan(int64_t M, int K1, int N) {
    WgradPlan p;
    p.blocks_k = (K1 + 255) / 256;
    p.blocks_n = (N + 255) / 256;
    const int kw = K1 < 256 ? K1 : 256, nw = N < 256 ? N : 256;
    const int kt_n = (kw + 31) / 32, nt_n = (nw + 31) / 32;          // tile rows / columns of the largest block
    p.ktp = kt_n <= 1 ? 1 : (kt_n <= 2 ? 2 : (kt_n <= 4 ? 4 : 8));
    const int G = 8 / p.ktp;                                          // waves per tile row
    p.n_phase = 1;
    p.ntp = 1;
    if (nt_n >= G) {
        p.tpw = (nt_n + G - 1) / G;                                   // tiles per wave: 1 .. 8
    } else {
        p.tpw = 1;
        p.ntp = nt_n <= 1 ? 1 : (nt_n <= 2 ? 2 : 4);
        p.n_phase = G / p.ntp;
    }
    // rows per staged chunk: as many as 8 float4 pieces per thread carry (narrow operands: a barrier per 32 rows would be all
    // the kernel does), at most 512
    const int ldl = ((kw + 3) & ~3) + ((nw + 3) & ~3);
    p.rows_chunk = (16384 / ldl) / 32 * 32;
    if (p.rows_chunk > 512) p.rows_chunk = 512;
    if (p.rows_chunk < 32) p.rows_chunk = 32;
    const int blocks = p.blocks_k * p.blocks_n;
    int64_t want = (num_cus() + blocks - 1) / blocks;                 // one workgroup per CU in total
    const int64_t max_by_rows = (M + 2 * p.rows_chunk - 1) / (2 * p.rows_chunk);
    if (want > max_by_rows) want = max_by_rows;
    if (want < 1) want = 1;
    p.rows_per_split = ((M + want - 1) / want + 31) / 32 * 32;
    if (p.rows_per_split < 32) p.rows_per_split = 32;
    p.splits = (int)((M + p.rows_per_split - 1) / p.rows_per_split);
    if (p.splits < 1) p.splits = 1;
    p.lds_bytes = (size_t)2 * p.rows_chunk * ldl * sizeof(float) + 128;   // (+ the reach of a last partial tile)
    return p;
}

}  // namespace

size_t linear_wgrad_workspace_bytes(int64_t M, int K1, int N) {
    const WgradPlan p = wgrad_plan(M, K1, N);
    const size_t slots = (size_t)p.splits * p.n_phase;
    return slots * ((size_t)K1 * N + 2 * (size_t)K1) * sizeof(float) + 256;
}

// dW[K1][N] (ldc) = dY[M][K1]^T X[M][N]; colsum[K1] = column sums of dY (may be null).  lda, ldb multiples of 4, 16-byte aligned
int launch_linear_wgrad_f32(const float* dY, int lda, const float* X, int ldb, float* dW, int ldc, float* colsum, int64_t M, int K1, int N,
                            void* ws, size_t ws_bytes, hipStream_t st) {
    T2P_CHECK_ARG(dW && M >= 0 && K1 >= 1 && N >= 1 && ldc >= N, "linear_wgrad: bad arguments");
    T2P_CHECK_ARG(M == 0 || (dY && X && lda >= K1 && ldb >= N && lda % 4 == 0 && ldb % 4 == 0 && ((((uintptr_t)dY) | ((uintptr_t)X)) & 15) == 0),
                  "linear_wgrad: lda=%d / ldb=%d must be multiples of 4 and the operands 16-byte aligned", lda, ldb);
    T2P_CHECK_ARG(K1 <= 256 || K1 % 4 == 0, "linear_wgrad: K1=%d beyond 256 must be a multiple of 4", K1);
    T2P_CHECK_ARG(N <= 256 || N % 4 == 0, "linear_wgrad: N=%d beyond 256 must be a multiple of 4", N);
    const WgradPlan p = wgrad_plan(M, K1, N);
    if (ws == nullptr || ws_bytes < linear_wgrad_workspace_bytes(M, K1, N)) {
        set_error("linear_wgrad: workspace %zu B < required %zu B", ws_bytes, linear_wgrad_workspace_bytes(M, K1, N));
        return T2P_E_WORKSPACE;
    }
    const int slots = p.splits * p.n_phase;
    float* part = (float*)ws;
    float* psum = part + (size_t)slots * K1 * N;
    ProfScope ps_("wgrad_f32", st);
    if (M > 0) {
        dim3 grid((unsigned)p.splits, (unsigned)(p.blocks_k * p.blocks_n));
#define WGRAD_CASE(T)                                                                                                              \
    {                                                                                                                              \
        T2P_TRY(reserve_lds((const void*)k_wgrad_f32<T>, p.lds_bytes, "wgrad_f32"));                                               \
        hipLaunchKernelGGL(k_wgrad_f32<T>, grid, dim3(512), p.lds_bytes, st, dY, lda, X, ldb, part, psum, M, K1, N, p.blocks_n,    \
                           p.rows_per_split, p.n_phase, p.ktp, p.ntp, p.rows_chunk);                                               \
    }
        switch (p.tpw) {
            case 1: WGRAD_CASE(1) break;
            case 2: WGRAD_CASE(2) break;
            case 3: WGRAD_CASE(3) break;
            case 4: WGRAD_CASE(4) break;
            case 5: WGRAD_CASE(5) break;
            case 6: WGRAD_CASE(6) break;
            case 7: WGRAD_CASE(7) break;
            default: WGRAD_CASE(8) break;
        }
#undef WGRAD_CASE
        T2P_CHECK_LAUNCH("linear_wgrad");
    }
    const int64_t total = (int64_t)K1 * N + K1;
    const int n_slots = M > 0 ? slots : 0;
    if (total <= 16384 && n_slots >= 128)
        hipLaunchKernelGGL(k_wgrad_reduce_wave, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, st, (const float*)part, (const float*)psum,
                           dW, ldc, colsum, K1, N, n_slots);
    else {
        const unsigned grid_r = (unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
        hipLaunchKernelGGL(k_wgrad_reduce, dim3(grid_r), dim3(256), 0, st, (const float*)part, (const float*)psum, dW, ldc, colsum, K1, N,
                           n_slots);
    }
    T2P_CHECK_LAUNCH("linear_wgrad_reduce");
    return 0;
}

}  // namespace t2p

// GlobalAbstraction layer 2 + global max pool (512 -> 1024, max over each object's 32 points), f16x3 path.
// (reference: GlobalAbstractionLayer.forward, models/pointcloud/pointnet2.py:45-49)
//
// The generic weight-stationary kernel (ws_gemm.hip) needs 256 registers per lane for the hi/lo weight image of a
// [512 x 32] column block, so it runs one wave per SIMD on 447 registers, keeps half of the weights in AGPRs (hipcc
// copies them back operand by operand: 4 v_accvgpr_mov + s_nop per MFMA) and has nothing to cover a stall with: it
// sat at 50 % of the MFMA rate.  Here the K = 512 reduction of a column block is split over a PAIR of waves (k halves
// of 256: 128 weight registers each), so the workgroup has 8 waves = 2 per SIMD at <= 256 registers, no AGPR traffic,
// and the partner's MFMAs cover each wave's LDS / barrier / epilogue time.  One wave of a pair hands its partial
// 32x32 block over through an LDS exchange area; the other adds it, applies bias + ReLU and reduces the 32 rows
// (= one object) to the pooled row.  The activation tiles are double-buffered in LDS and filled by LDS-DMA loads.
//
// Input: the activations of GA layer 1 as two fp16 planes (hi, lo) [M][512] (written by ws_gemm's split output).
// Column slices of 128 of one row stream run on the same XCD (block b -> XCD b % 8) and share the rows through its L2.
#include "t2p_common.h"
// The products run on v_mfma_f32_16x16x32_f16 (four 16 x 16 blocks per wave and tile; round 5: 14.87-14.92 -> 14.78-14.82 ms per step
// against v_mfma_f32_32x32x16_f16 in three interleaved A/B pairs - see sa3.hip for the operand layout).

namespace t2p {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr int K = 512, NW = 128, NT = 512;
constexpr int KH = K / 2;            // k range of one wave
constexpr int S16_FULL = K / 16;     // steps of the packed weight image
constexpr int LDHH = K + 8;          // plane row stride in halves (16-byte pad: conflict-free ds_read_b128)
constexpr int PLANE = 32 * LDHH;     // halves
constexpr int TILE_HALVES = 2 * PLANE;
constexpr int XCH_FLOATS = 4 * 16 * 64;  // exchange area: 4 column blocks x 16 accumulator registers x 64 lanes
constexpr size_t kLds = (size_t)2 * TILE_HALVES * 2 + (size_t)XCH_FLOATS * 4;
constexpr int CHUNKS = (2 * 32 * (K / 8)) / NT;  // 16-byte chunks staged per thread and tile (= 8)

__global__ __launch_bounds__(NT, 2) void k_ga2(WsParams p, int n_slices) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    _Float16* tile = (_Float16*)lds;                  // [2][hi plane | lo plane]
    float* xch = (float*)(tile + 2 * TILE_HALVES);    // [4][16][64]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave & 3, kh = wave >> 2;

    // XCD-aware stream / slice mapping (same as ws_gemm.hip)
    const int lin = blockIdx.x, nblk = gridDim.x;
    int slice, stream, n_streams;
    if ((nblk % (8 * n_slices)) == 0) {
        const int xcd = lin & 7, j = lin >> 3, per_xcd = nblk >> 3;
        slice = j % n_slices;
        stream = xcd * (per_xcd / n_slices) + j / n_slices;
        n_streams = nblk / n_slices;
    } else {
        slice = lin % n_slices;
        stream = lin / n_slices;
        n_streams = nblk / n_slices;
    }
    const int ncol0 = slice * NW + wn * 32;

    // 16x16x32 operands: lane = 16 q + i holds k = kh*256 + 64 q + 8 s + e of step s (s < 8), column 16 cb + i of the wave's 32;
    // packed image entry (step = 8 q + s, half = kh)
    constexpr int S32 = KH / 32;
    const int q16 = lane >> 4, i16 = lane & 15;
    half8 w_hi[2][S32], w_lo[2][S32];
    {
        const uint4* wp = (const uint4*)p.W_x3;
        const int plane_u4 = (p.ldw / 32) * S16_FULL * 64;
#pragma unroll
        for (int cb = 0; cb < 2; cb++)
#pragma unroll
            for (int s = 0; s < S32; s++) {
                const int idx = (((ncol0 / 32) * S16_FULL + (8 * q16 + s)) * 2 + kh) * 32 + 16 * cb + i16;
                w_hi[cb][s] = __builtin_bit_cast(half8, wp[idx]);
                w_lo[cb][s] = __builtin_bit_cast(half8, wp[plane_u4 + idx]);
            }
    }
    float bias2[2];
#pragma unroll
    for (int cb = 0; cb < 2; cb++) bias2[cb] = p.bias ? p.bias[ncol0 + 16 * cb + i16] : 0.f;

    // The next tile travels HBM/L2 -> LDS directly (global_load_lds_dwordx4: one plane row = 1 KB = one wave instruction,
    // 8 per wave and tile; rows are padded, the 64 lanes of an instruction are not): no staging registers, no ds_write
    // traffic beside the operand reads; the __syncthreads() that ends the tile drains it (vmcnt(0)).  Measured against
    // register staging (8 global_load_dwordx4 + 8 ds_write_b128 per thread, written in the second half of the k loop):
    // 15.3 vs 15.8 ms per 12k cells (written in the first half: 16.1).
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void gl_void;
    auto dma_piece = [&](int64_t g, int buf, int it) {
        const int q = it * (NT / 64) + wave;  // 64 plane rows over 8 waves
        const int pl = q >> 5, row = q & 31;
        const _Float16* base = (const _Float16*)(pl ? p.A_lo : p.A_hi);
        const _Float16* src = base + (g * 32 + row) * (int64_t)p.lda + lane * 8;
        _Float16* dst = tile + buf * TILE_HALVES + pl * PLANE + row * LDHH;  // wave-uniform; the lane's 16 bytes follow
        __builtin_amdgcn_global_load_lds((gl_void*)src, (lds_void*)dst, 16, 0, 0);
    };
    auto dma_tile = [&](int64_t g, int buf) {
#pragma unroll
        for (int it = 0; it < CHUNKS; it++) dma_piece(g, buf, it);
    };
    int64_t g = stream;
    if (g < p.n_groups) dma_tile(g, 0);
    __syncthreads();
    for (int i = 0; g < p.n_groups; g += n_streams, i++) {
        const int64_t gn = g + n_streams;
        const bool more = gn < p.n_groups;

        typedef float f32x4v __attribute__((ext_vector_type(4)));
        const f32x4v kZero4 = {0.f, 0.f, 0.f, 0.f};
        f32x4v acc[4], accx[4];    // [2 rb + cb]: rows 16 rb + 4 q16 + v, column 16 cb + i16
        const _Float16* hrow = tile + (i & 1) * TILE_HALVES + i16 * LDHH + kh * KH + q16 * 64;
        half8 a_hi[2], a_lo[2];
#pragma unroll
        for (int rb = 0; rb < 2; rb++) {
            a_hi[rb] = *(const half8*)(hrow + rb * 16 * LDHH);
            a_lo[rb] = *(const half8*)(hrow + PLANE + rb * 16 * LDHH);
        }
#pragma unroll
        for (int sl = 0; sl < 4 * S32; sl++) {
            const int s = sl >> 2, b = sl & 3, rb = b >> 1, cb = b & 1;
            __builtin_amdgcn_sched_barrier(0);
            // operands of the next step: row block 0 is idle while block 1 multiplies, and the other way round
            if (s + 1 < S32 && b == 2) {
                a_hi[0] = *(const half8*)(hrow + (s + 1) * 8);
                a_lo[0] = *(const half8*)(hrow + PLANE + (s + 1) * 8);
            }
            if (s > 0 && b == 0) {
                a_hi[1] = *(const half8*)(hrow + 16 * LDHH + s * 8);
                a_lo[1] = *(const half8*)(hrow + PLANE + 16 * LDHH + s * 8);
            }
            acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[rb], w_hi[cb][s], s == 0 ? kZero4 : acc[b], 0, 0, 0);
            accx[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[rb], w_lo[cb][s], s == 0 ? kZero4 : accx[b], 0, 0, 0);
            accx[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo[rb], w_hi[cb][s], accx[b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // one LDS-DMA piece of the next tile per two slots of the first half, behind a block's three MFMAs (round 5: 14.77-14.81 ms
            // per step against 14.89-14.98 with the piece between the second and third MFMA)
            if (more && (sl & 1) == 0 && (sl >> 1) < CHUNKS) dma_piece(gn, (i + 1) & 1, sl >> 1);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int v = 0; v < 4; v++) acc[b][v] = fmaf(accx[b][v], 1.f / 2048.f, acc[b][v]);
        float* x = xch + wn * (16 * 64) + lane;
        const bool sender = ((i & 1) != 0) == (kh == 0);
        if (sender) {
#pragma unroll
            for (int b = 0; b < 4; b++)
#pragma unroll
                for (int v = 0; v < 4; v++) x[(4 * b + v) * 64] = acc[b][v];
        }
        __syncthreads();
        if (!sender) {
            // partner's partial + bias, ReLU (= starting the max at 0), max over the 32 rows of the object: per column block
            // over its two row blocks x four rows in this lane, then over the four lane quarters
#pragma unroll
            for (int cb = 0; cb < 2; cb++) {
                float m = 0.f;
#pragma unroll
                for (int rb = 0; rb < 2; rb++)
#pragma unroll
                    for (int v = 0; v < 4; v++) m = fmaxf(m, (acc[2 * rb + cb][v] + x[(4 * (2 * rb + cb) + v) * 64]) + bias2[cb]);
                m = fmaxf(m, __shfl_xor(m, 16, 64));
                m = fmaxf(m, __shfl_xor(m, 32, 64));
                if (q16 == 0) p.out[g * (int64_t)p.ldo + ncol0 + 16 * cb + i16] = m;
            }
        }
    }
}

}  // namespace

// out[M/32][ldo] (columns [0, 1024)) = max over each 32-row group of relu(A W + b); A as fp16 hi / lo planes [M][lda]
int launch_ga2(const WsParams& p_in, hipStream_t st) {
    T2P_TRY(reserve_lds((const void*)k_ga2, kLds, "ga2"));
    WsParams p = p_in;
    p.n_groups = p.M / 32;
    if (p.n_groups <= 0) return 0;
    const int n_slices = 1024 / NW;
    int64_t streams = matrix_wgs() / n_slices;
    if (streams < 1) streams = 1;
    if (streams > p.n_groups) streams = p.n_groups;
    if (streams >= 8) streams -= streams % 8;
    ProfScope ps_("ws_groupmax_k512_n1024", st);
    T2P_REPEAT(ps_) hipLaunchKernelGGL(k_ga2, dim3((unsigned)(streams * n_slices)), dim3(NT), kLds, st, p, n_slices);
    T2P_CHECK_LAUNCH("ga2");
    return 0;
}

}  // namespace t2p

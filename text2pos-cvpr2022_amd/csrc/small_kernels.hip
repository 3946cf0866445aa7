// Small HBM/latency-bound helper kernels of the cell branch: row L2-normalisation, colour / position MLPs,
// per-cell pooling and the per-cell kNN graph (the K <= 6 layer-1 tables live in sample_group.hip).
//
// Reference call sites: F.normalize models/object_encoder.py:110-135, models/cell_retrieval.py:73,94,105;
// gnn.global_max_pool / global_mean_pool models/cell_retrieval.py:98,102; knn inside gnn.DynamicEdgeConv
// models/cell_retrieval.py:46-48,97.
#include "t2p_common.h"

namespace t2p {
namespace {

// F.normalize(x, dim=-1): x / max(||x||_2, 1e-12); one wavefront per row.
__global__ __launch_bounds__(256) void k_rownorm(const float* __restrict__ in, int ld_in, int64_t n_rows, int dim,
                                                 float* __restrict__ out, int ld_out, int col0) {
    const int lane = threadIdx.x & 63;
    int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const float* x = in + row * ld_in;
    float ss = 0.f;
    for (int i = lane; i < dim; i += 64) ss = fmaf(x[i], x[i], ss);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 64);
    float nrm = sqrtf(ss);
    float den = nrm > 1e-12f ? nrm : 1e-12f;
    float* y = out + row * ld_out + col0;
    for (int i = lane; i < dim; i += 64) y[i] = x[i] / den;
}

// F.normalize(table[idx[row]]) -> out[row*ld_out + col0 ...]: the --class_embed / --color_embed ablations
// (models/object_encoder.py:103-120).  One wavefront per row.
__global__ __launch_bounds__(256) void k_gather_rownorm(const float* __restrict__ table, const int32_t* __restrict__ idx,
                                                        int64_t n_rows, int dim, float* __restrict__ out, int ld_out,
                                                        int col0) {
    const int lane = threadIdx.x & 63;
    int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const float* x = table + (int64_t)idx[row] * dim;
    float ss = 0.f;
    for (int i = lane; i < dim; i += 64) ss = fmaf(x[i], x[i], ss);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 64);
    const float nrm = sqrtf(ss);
    const float den = nrm > 1e-12f ? nrm : 1e-12f;
    float* y = out + row * ld_out + col0;
    for (int i = lane; i < dim; i += 64) y[i] = x[i] / den;
}

__global__ void k_segpool(const float* __restrict__ in, int dim, const int32_t* __restrict__ seg_ptr, int n_seg,
                          float* __restrict__ out, int mean) {
    int s = blockIdx.x;
    int lo = seg_ptr[s], hi = seg_ptr[s + 1];
    for (int c = threadIdx.x; c < dim; c += blockDim.x) {
        if (mean) {
            float acc = 0.f;
            for (int r = lo; r < hi; r++) acc += in[(int64_t)r * dim + c];
            out[(int64_t)s * dim + c] = hi > lo ? acc / (float)(hi - lo) : 0.f;
        } else {
            float m = hi > lo ? -INFINITY : 0.f;
            for (int r = lo; r < hi; r++) m = fmaxf(m, in[(int64_t)r * dim + c]);
            out[(int64_t)s * dim + c] = m;
        }
    }
}

// kNN graph inside one segment (cell): all-pairs squared distances in LDS, then a k-pass ordered selection.
// Distance pinned to the sequential fp32 form acc = acc + (a-b)*(a-b), no FMA (oracle/primitives.c); it is symmetric bit
// for bit ((a-b)^2 == (b-a)^2), so only the n (n + 1) / 2 pairs j >= i are formed - enumerated densely over the threads -
// and, for cells of up to kKnnStage rows, from a copy of the rows in LDS (16-byte reads, row stride padded by 16 B) instead
// of two dependent 4-byte global loads per term.
constexpr int kKnnStage = 64;
__global__ __launch_bounds__(256) void k_knn(const float* __restrict__ x, int dim, const int32_t* __restrict__ seg_ptr,
                                             int k, int32_t* __restrict__ out_idx, int stage_rows) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) float dmat[];   // [n][n] distances | [n][dim + 4] staged rows
    int s = blockIdx.x;
    int lo = seg_ptr[s], hi = seg_ptr[s + 1];
    int n = hi - lo;
    const bool staged = n <= stage_rows && (dim & 3) == 0;
    const int ld = dim + 4;
    float* xs = dmat + ((n * n + 3) & ~3);
    if (staged) {
        const int q4 = dim >> 2;
        for (int e = threadIdx.x; e < n * q4; e += blockDim.x) {
            const int r = e / q4, c = e - r * q4;
            *(f32x4*)(xs + r * ld + c * 4) = *(const f32x4*)(x + (int64_t)(lo + r) * dim + c * 4);
        }
        __syncthreads();
    }
    const int n_pairs = n * (n + 1) / 2;
    for (int e = threadIdx.x; e < n_pairs; e += blockDim.x) {
        // pair e of the upper triangle, row-major: row i starts at i n - i (i - 1) / 2
        int a_lo = 0, a_hi = n - 1;
        while (a_lo < a_hi) {
            const int m = (a_lo + a_hi + 1) >> 1;
            if (m * n - m * (m - 1) / 2 <= e) a_lo = m; else a_hi = m - 1;
        }
        const int i = a_lo, j = i + (e - (i * n - i * (i - 1) / 2));
        float acc = 0.f;
        if (staged) {
            const float* a = xs + j * ld;
            const float* b = xs + i * ld;
            for (int t = 0; t < dim; t += 4) {
                const f32x4 av = *(const f32x4*)(a + t), bv = *(const f32x4*)(b + t);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    float df = av[u] - bv[u];
                    float sq = df * df;
                    acc = acc + sq;
                }
            }
        } else {
            const float* a = x + (int64_t)(lo + j) * dim;
            const float* b = x + (int64_t)(lo + i) * dim;
            for (int t = 0; t < dim; t++) {
                float df = a[t] - b[t];
                float sq = df * df;
                acc = acc + sq;
            }
        }
        dmat[i * n + j] = acc;
        dmat[j * n + i] = acc;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        float last_d = -1.f;
        int last_j = -1;
        for (int q = 0; q < k; q++) {
            float bd = INFINITY;
            int bj = -1;
            for (int j = 0; j < n; j++) {
                float d = dmat[i * n + j];
                bool after = (d > last_d) || (d == last_d && j > last_j);
                if (after && (bj < 0 || d < bd)) { bd = d; bj = j; }
            }
            out_idx[(int64_t)(lo + i) * k + q] = bj >= 0 ? lo + bj : -1;
            if (bj >= 0) { last_d = bd; last_j = bj; }
            else { last_d = INFINITY; last_j = 0x7fffffff; }
        }
    }
}

// colour / position encoders of ObjectEncoder (models/object_encoder.py:40-41,124-135): 3 -> 64 -> D, each layer
// Linear+BN+ReLU, then F.normalize.  A block owns 32 objects: the hidden layer goes to LDS, every thread then owns one
// (or two) output columns for all 32 objects, so a W2 element is fetched once per 32 objects (one wave per object
// re-read the whole 64 x D matrix per object: 12 GB of L2 traffic per call at 192 k objects).
// ROWS = 32 for large batches; 8 for small ones (round 6): a thread's 64-step fma chain per row is serial, so a block of 32 rows takes
// ~32 us whatever the batch, and a 1,000-object call (the reference's 64-cell batches) filled 31 of the 256 CUs with it.  The
// summation order of an output does not depend on ROWS: same bits.
template <int kMlp3Rows>
__global__ __launch_bounds__(256) void k_mlp3_norm(const float* __restrict__ in3, int64_t n_rows,
                                                   const float* __restrict__ w1, const float* __restrict__ b1,
                                                   const float* __restrict__ w2, const float* __restrict__ b2, int D,
                                                   float* __restrict__ out, int ld_out, int col0) {
    __shared__ float hid[kMlp3Rows][64];
    __shared__ float ss[4][kMlp3Rows];   // per-wave partial sums of squares, combined in a fixed order (deterministic)
    const int tid = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * kMlp3Rows;
    for (int i = tid; i < kMlp3Rows * 64; i += 256) {
        const int r = i >> 6, u = i & 63;
        float h = 0.f;
        if (row0 + r < n_rows) {
            const float* x = in3 + (row0 + r) * 3;
            float a = b1[u];
            a = fmaf(x[0], w1[u], a);
            a = fmaf(x[1], w1[64 + u], a);
            a = fmaf(x[2], w1[128 + u], a);
            h = fmaxf(a, 0.f);
        }
        hid[r][u] = h;
    }
    __syncthreads();
    float o[2][kMlp3Rows];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int c = tid + 256 * j;
        if (c >= D) break;
        const float bv = b2[c];
#pragma unroll
        for (int r = 0; r < kMlp3Rows; r++) o[j][r] = bv;
        for (int k = 0; k < 64; k++) {
            const float w = w2[k * D + c];
#pragma unroll
            for (int r = 0; r < kMlp3Rows; r++) o[j][r] = fmaf(hid[r][k], w, o[j][r]);
        }
#pragma unroll
        for (int r = 0; r < kMlp3Rows; r++) o[j][r] = fmaxf(o[j][r], 0.f);
    }
    // row norms: per-wave butterfly, one partial per wave and row
#pragma unroll
    for (int r = 0; r < kMlp3Rows; r++) {
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < 2; j++)
            if (tid + 256 * j < D) v = fmaf(o[j][r], o[j][r], v);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if ((tid & 63) == 0) ss[tid >> 6][r] = v;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int c = tid + 256 * j;
        if (c >= D) break;
#pragma unroll
        for (int r = 0; r < kMlp3Rows; r++) {
            if (row0 + r >= n_rows) break;
            const float nrm = sqrtf((ss[0][r] + ss[1][r]) + (ss[2][r] + ss[3][r]));
            out[(row0 + r) * ld_out + col0 + c] = o[j][r] / (nrm > 1e-12f ? nrm : 1e-12f);
        }
    }
}

// ---- on-device dataloader ---------------------------------------------------------------------------
// The per-object transform chain of dataloading/kitti360pose/utils.py:89-110 (Data -> T.FixedPoints(P) [-> T.RandomRotate]
// -> T.NormalizeScale -> Batch) with one wavefront per object.  The P sampled points are staged in LDS ([P][3] floats per
// wave); NormalizeScale then reproduces the host chain BIT FOR BIT:
//   * `pos.mean(dim=-2)` is summed in the order ATen's CPU kernel uses for a [P, 3] fp32 tensor (SumKernel.cpp: row_sum =
//     multi_row_sum over the row viewed as [P/4, 4]: four interleaved partial sums per column, each a cascade that folds
//     its running sum into the next level every 16 items (level_power = max(4, CeilLog2(P/4) / 4) = 4 for every P <= 2^18),
//     the P % 4 tail rows added to partial sum 0, then ((p0 + p1) + p2) + p3), followed by a true division by P;
//   * `pos - mean`, `abs().max()`, `(1 / max) * 0.999999` and `pos * scale` are single correctly rounded fp32 operations
//     in both places (the library is built with -ffp-contract=off).
// tests/test_gpu_parity.py holds the kernel to torch's result with array_equal.
__device__ __forceinline__ uint64_t pack_mix64(uint64_t x) {   // splitmix64 finaliser (= synthetic._mix on the host)
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// Column sums of the wave's staged points in ATen's order; every lane returns (sx, sy, sz).
__device__ __forceinline__ void aten_column_sums(const float* __restrict__ pts, int n_pts, int lane, float& sx, float& sy, float& sz) {
    float part = 0.f;
    if (lane < 12) {                      // lane = 4 * column + k: partial sum k of that column
        const int c = lane >> 2, k = lane & 3;
        const int size_ilp = n_pts >> 2;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        int i = 0;
        while (i + 16 <= size_ilp) {
            for (int j = 0; j < 16; j++, i++) acc[0] += pts[(4 * i + k) * 3 + c];
            for (int j = 1; j < 4; j++) {
                acc[j] += acc[j - 1];
                acc[j - 1] = 0.f;
                if ((i & (0xF << (4 * j))) != 0) break;
            }
        }
        for (; i < size_ilp; i++) acc[0] += pts[(4 * i + k) * 3 + c];
        for (int j = 1; j < 4; j++) acc[0] += acc[j];
        if (k == 0)
            for (int r = size_ilp * 4; r < n_pts; r++) acc[0] += pts[r * 3 + c];
        part = acc[0];
    }
    // ((p0 + p1) + p2) + p3 of each column
    const float p1 = __shfl_down(part, 1, 64), p2 = __shfl_down(part, 2, 64), p3 = __shfl_down(part, 3, 64);
    const float tot = ((part + p1) + p2) + p3;
    sx = __shfl(tot, 0, 64);
    sy = __shfl(tot, 4, 64);
    sz = __shfl(tot, 8, 64);
}

// T.NormalizeScale of the wave's staged points, written to q [n_pts][3].
__device__ __forceinline__ void normalize_scale_store(const float* __restrict__ pts, int n_pts, int lane, float* __restrict__ q) {
    // the callers staged `pts` lane-strided; the column sums read it in another lane layout: order the wave's LDS stores before
    // its LDS loads explicitly (in-order LDS issue makes it work without, but nothing would stop a future compiler from moving them)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float sx, sy, sz;
    aten_column_sums(pts, n_pts, lane, sx, sy, sz);
    const float mx = sx / (float)n_pts, my = sy / (float)n_pts, mz = sz / (float)n_pts;
    float amax = 0.f;
    for (int i = lane; i < n_pts; i += 64)
        amax = fmaxf(amax, fmaxf(fabsf(pts[i * 3] - mx), fmaxf(fabsf(pts[i * 3 + 1] - my), fabsf(pts[i * 3 + 2] - mz))));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
    const float scale = (1.f / amax) * 0.999999f;
    for (int i = lane; i < n_pts; i += 64) {
        q[i * 3] = (pts[i * 3] - mx) * scale;
        q[i * 3 + 1] = (pts[i * 3 + 1] - my) * scale;
        q[i * 3 + 2] = (pts[i * 3 + 2] - mz) * scale;
    }
}

// t2p_pack_objects: ragged raw objects back to back + the host's T.FixedPoints draw (sample_idx) -> packed encoder inputs and
// the per-object means of models/object_encoder.py:121-131 (Object3d.get_center() / get_color_rgb(): means over ALL raw
// points, accumulated in float64 like NumPy - from the fp32 copies of the points that were uploaded).
// rot (optional, [n_obj][2] = cos, sin of the object's angle): T.RandomRotate(deg, axis=2) of the training transform
// (training/coarse.py:192-198), applied between the resampling and NormalizeScale: pos <- pos @ [[c, s, 0], [-s, c, 0],
// [0, 0, 1]].  The angle is drawn on the host like the FixedPoints indices.
__global__ __launch_bounds__(256) void k_pack_objects(const float* __restrict__ raw_xyz, const float* __restrict__ raw_rgb,
                                                      const int32_t* __restrict__ obj_ptr,
                                                      const int32_t* __restrict__ sample_idx,
                                                      const float* __restrict__ rot, int64_t n_obj, int n_pts,
                                                      float* __restrict__ xyz, float* __restrict__ rgb,
                                                      float* __restrict__ center, float* __restrict__ mean_rgb) {
    extern __shared__ float pack_lds[];
    const int lane = threadIdx.x & 63;
    float* pts = pack_lds + (threadIdx.x >> 6) * n_pts * 3;
    const int64_t o = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= n_obj) return;
    const int lo = obj_ptr[o], hi = obj_ptr[o + 1], m = hi - lo;
    // means over all raw points
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int i = lane; i < m; i += 64) {
        const float* p = raw_xyz + (int64_t)(lo + i) * 3;
        const float* c = raw_rgb + (int64_t)(lo + i) * 3;
#pragma unroll
        for (int d = 0; d < 3; d++) { acc[d] += (double)p[d]; acc[3 + d] += (double)c[d]; }
    }
#pragma unroll
    for (int d = 0; d < 6; d++) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) acc[d] += __shfl_xor(acc[d], off, 64);
    }
    if (lane < 3) center[o * 3 + lane] = (float)(acc[lane] / (double)m);
    else if (lane < 6) mean_rgb[o * 3 + lane - 3] = (float)(acc[lane] / (double)m);
    // resample (+ rotate) into LDS, colours straight to the output
    const bool rotate = rot != nullptr;
    const float rc = rotate ? rot[o * 2] : 1.f, rs = rotate ? rot[o * 2 + 1] : 0.f;
    for (int i = lane; i < n_pts; i += 64) {
        const int64_t j = lo + sample_idx[o * n_pts + i];
        float x = raw_xyz[j * 3], y = raw_xyz[j * 3 + 1];
        if (rotate) {  // row vector times the matrix, terms added in column order like the fp32 matmul
            const float xr = x * rc + y * -rs, yr = x * rs + y * rc;
            x = xr;
            y = yr;
        }
        pts[i * 3] = x;
        pts[i * 3 + 1] = y;
        pts[i * 3 + 2] = raw_xyz[j * 3 + 2];
        float* c = rgb + (o * n_pts + i) * 3;
        c[0] = raw_rgb[j * 3];
        c[1] = raw_rgb[j * 3 + 1];
        c[2] = raw_rgb[j * 3 + 2];
    }
    normalize_scale_store(pts, n_pts, lane, xyz + o * n_pts * 3);
}

// t2p_pack_scene_objects: the same chain for a scene whose raw points live in HBM (uploaded once): output slot s takes scene
// object obj_id[s]; its T.FixedPoints draw is counter-based - point p of the slot is raw point
//     ((mix64(key[s] ^ p * 0xD6E8FEB86659FD93) >> 32) * m) >> 32        (m = points of the object)
// so a slot's sample depends on its key only (the host derives it from (seed, global cell index, object slot):
// pipeline.PerCellTransform), whichever rank, batch or stream packs it; centre / mean colour are gathered from the scene's
// per-object tables (the reference's float64 means, computed once on the host: exact).
__global__ __launch_bounds__(256) void k_pack_scene(const float* __restrict__ raw_xyz, const float* __restrict__ raw_rgb,
                                                    const int32_t* __restrict__ obj_ptr, const int32_t* __restrict__ obj_id,
                                                    const uint64_t* __restrict__ key, const float* __restrict__ scene_center,
                                                    const float* __restrict__ scene_color, int64_t n_out, int n_pts,
                                                    float* __restrict__ xyz, float* __restrict__ rgb, float* __restrict__ center,
                                                    float* __restrict__ mean_rgb, int32_t* __restrict__ idx_out) {
    extern __shared__ float pack_lds[];
    const int lane = threadIdx.x & 63;
    float* pts = pack_lds + (threadIdx.x >> 6) * n_pts * 3;
    const int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= n_out) return;
    const int64_t o = obj_id[s];
    const int64_t lo = obj_ptr[o];
    const uint64_t m = (uint64_t)(obj_ptr[o + 1] - obj_ptr[o]);
    const uint64_t k = key[s];
    if (lane < 3) {
        if (center != nullptr) center[s * 3 + lane] = scene_center[o * 3 + lane];
        if (mean_rgb != nullptr) mean_rgb[s * 3 + lane] = scene_color[o * 3 + lane];
    }
    for (int i = lane; i < n_pts; i += 64) {
        const uint32_t d = (uint32_t)(((pack_mix64(k ^ ((uint64_t)i * 0xD6E8FEB86659FD93ull)) >> 32) * m) >> 32);
        const int64_t j = lo + d;
        pts[i * 3] = raw_xyz[j * 3];
        pts[i * 3 + 1] = raw_xyz[j * 3 + 1];
        pts[i * 3 + 2] = raw_xyz[j * 3 + 2];
        if (rgb != nullptr) {
            float* c = rgb + (s * n_pts + i) * 3;
            c[0] = raw_rgb[j * 3];
            c[1] = raw_rgb[j * 3 + 1];
            c[2] = raw_rgb[j * 3 + 2];
        }
        if (idx_out != nullptr) idx_out[s * n_pts + i] = (int32_t)d;
    }
    normalize_scale_store(pts, n_pts, lane, xyz + s * n_pts * 3);
}

// PairwiseRankingLoss (training/losses.py:126-164) on the score matrix S = im_n s_n^T [B][B]:
//   cost_s[i][j] = max(0, margin - S[j][j] + S[i][j]),  cost_im[i][j] = max(0, margin - S[i][i] + S[i][j]),  diagonals 0,
//   loss = (sum cost_s + sum cost_im) / B.
// One block per row: the row's loss, dLoss/dS off the diagonal, and the row's count of active cost_im terms (which
// the diagonal entry of the gradient needs); fixed-order reductions, so the result is run-to-run identical.
__global__ __launch_bounds__(256) void k_rank_rows(const float* __restrict__ S, int B, float margin,
                                                   float* __restrict__ row_loss, float* __restrict__ dS,
                                                   float* __restrict__ row_cnt) {
    __shared__ float red[2][256];
    const int i = blockIdx.x, tid = threadIdx.x;
    const float di = S[(int64_t)i * B + i], inv = 1.f / (float)B;
    float loss = 0.f, cnt = 0.f;
    for (int j = tid; j < B; j += 256) {
        if (j == i) continue;
        const float sij = S[(int64_t)i * B + j];
        const float cs = margin - S[(int64_t)j * B + j] + sij, ci = margin - di + sij;
        loss += fmaxf(cs, 0.f) + fmaxf(ci, 0.f);
        cnt += ci > 0.f ? 1.f : 0.f;
        dS[(int64_t)i * B + j] = ((cs > 0.f ? 1.f : 0.f) + (ci > 0.f ? 1.f : 0.f)) * inv;
    }
    red[0][tid] = loss;
    red[1][tid] = cnt;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            red[0][tid] += red[0][tid + s];
            red[1][tid] += red[1][tid + s];
        }
        __syncthreads();
    }
    if (tid == 0) {
        row_loss[i] = red[0][0];
        row_cnt[i] = red[1][0];
    }
}
// diagonal of the gradient: -(active cost_s terms of column j + active cost_im terms of row j) / B
__global__ __launch_bounds__(256) void k_rank_diag(const float* __restrict__ S, int B, float margin,
                                                   const float* __restrict__ row_cnt, float* __restrict__ dS) {
    __shared__ float red[256];
    const int j = blockIdx.x, tid = threadIdx.x;
    const float dj = S[(int64_t)j * B + j];
    float cnt = 0.f;
    for (int i = tid; i < B; i += 256)
        if (i != j) cnt += (margin - dj + S[(int64_t)i * B + j]) > 0.f ? 1.f : 0.f;
    red[tid] = cnt;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    if (tid == 0) dS[(int64_t)j * B + j] = -(red[0] + row_cnt[j]) / (float)B;
}

// HardestRankingLoss (training/losses.py:167-201): per row i the largest hinge margin + S[i][j] - S[i][i] over j != i, per
// column j the largest margin + S[i][j] - S[j][j] over i != j; loss = mean of the row maxima + mean of the column maxima.
// Block b < B: row b; block b >= B: column b - B.  Writes the maximum (>= 0) and its position (-1: nothing above 0; first
// position on ties, torch.max's choice on the CPU path the reference was written against).
__global__ __launch_bounds__(256) void k_hardest_scan(const float* __restrict__ S, int B, float margin, float* __restrict__ best,
                                                      int32_t* __restrict__ where) {
    __shared__ float bv[256];
    __shared__ int bi[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const bool is_row = b < B;
    const int a = is_row ? b : b - B;
    const float d = S[(int64_t)a * B + a];
    float v = 0.f;
    int w = -1;
    for (int k = tid; k < B; k += 256) {
        if (k == a) continue;
        const float h = margin + (is_row ? S[(int64_t)a * B + k] : S[(int64_t)k * B + a]) - d;
        if (h > v) { v = h; w = k; }
    }
    bv[tid] = v;
    bi[tid] = w;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            const float ov = bv[tid + s];
            const int ow = bi[tid + s];
            if (ow >= 0 && (ov > bv[tid] || (ov == bv[tid] && (bi[tid] < 0 || ow < bi[tid])))) {
                bv[tid] = ov;
                bi[tid] = ow;
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        best[b] = bv[0];
        where[b] = bi[0];
    }
}
// dLoss/dS from the winners: +1/B at the winning entry, -1/B on the diagonal of its row (row maxima) / column (column maxima)
__global__ void k_hardest_grad(const int32_t* __restrict__ where, int B, float* __restrict__ dS) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= 2 * B) return;
    const int w = where[b];
    if (w < 0) return;
    const bool is_row = b < B;
    const int a = is_row ? b : b - B;
    const float g = 1.f / (float)B;
    atomicAdd(dS + (is_row ? (int64_t)a * B + w : (int64_t)w * B + a), g);
    atomicAdd(dS + (int64_t)a * B + a, -g);
}

__global__ void k_cell_index(const int32_t* __restrict__ cell_ptr, int n_cells, int32_t o_lo,
                             int32_t* __restrict__ seg_ptr_local, int32_t* __restrict__ first, uint32_t* guard) {
    if (guard != nullptr && blockIdx.x == 0 && threadIdx.x < G_SLOTS) guard[threadIdx.x] = 0u;  // new chunk: guard words start at 0
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c <= n_cells; c += gridDim.x * blockDim.x) {
        const int32_t lo = cell_ptr[c] - o_lo;
        seg_ptr_local[c] = lo;
        if (c < n_cells) {
            const int32_t hi = cell_ptr[c + 1] - o_lo;
            for (int32_t o = lo; o < hi; o++) first[o] = lo;
        }
    }
}

// fp16-range verdict of one chunk (t2p_common.h GuardSlot): bit 0..2 = SA level l may have staged relu(A_j - B_i) past
// fp16's largest finite value, bit 3 = an SA output row (split by the next dense kernel) did, bit 4 = the GA hidden planes
// may have (bound ||W1||_1 max(F_3, 1) + max|b1|), bit 5 = a row of the LDS-tiled GEMMs did, bit 6 = NaN in the input points / colours,
// bit 7 = an SA level's hidden activations or outputs are all below kGuardTiny (too small for the fp16 pieces)
__global__ void k_guard_check(const uint32_t* __restrict__ guard, int32_t* flag, GuardBounds gb) {
    if (threadIdx.x != 0) return;
    const float lim = 65504.f;
    int code = 0;
    const uint32_t in_bits = guard[G_INPUT];
    if (in_bits > 0x7f800000u) code |= 64;                                  // a NaN among the input points / colours
    const float in_max = in_bits > 0x7f800000u ? __builtin_inff() : fmaxf(__uint_as_float(in_bits), 1.f);   // unpublished: within [-1, 1]
    for (int l = 0; l < 3; l++) {
        // point table: level 0 bounded from the inputs, levels 1 and 2 reported by their dense kernels (unpublished = below
        // the floor); centroid table: ||W1p||_1 max|xyz|
        const float a = l == 0 ? gb.a1_l1 * in_max + gb.a1_bmax : fmaxf(__uint_as_float(guard[l == 1 ? G_A2 : G_A3]), kGuardFloor);
        const float b = gb.wp_l1[l] * in_max;
        if (!(a + b < lim)) code |= 1 << l;   // also catches inf patterns
        if (!(__uint_as_float(guard[G_F1 + l]) < lim)) code |= 8;
        // low side (bit 7): every hidden activation relu(A_j - B_i) of the level is below kGuardTiny (a_lo: exact maximum of
        // the point table at levels 1 and 2, the input bound at level 0), or the level's whole OUTPUT is (exact maximum)
        const float a_lo = l == 0 ? a : __uint_as_float(guard[l == 1 ? G_A2 : G_A3]);
        const float f_lo = __uint_as_float(guard[G_F1 + l]);
        if ((a_lo + b > 0.f && a_lo + b < kGuardTiny) || (f_lo > 0.f && f_lo < kGuardTiny)) code |= 128;
    }
    if (!(fmaxf(__uint_as_float(guard[G_F3]), 1.f) * gb.ga1_l1 + gb.ga1_bmax < lim)) code |= 16;
    if (!(__uint_as_float(guard[G_GEMM_IN]) < lim)) code |= 32;
    if (code != 0) atomicOr(flag, code);
}

}  // namespace

int launch_guard_check(const uint32_t* guard, int32_t* overflow_flag, const GuardBounds& b, hipStream_t st) {
    if (guard == nullptr || overflow_flag == nullptr) return 0;
    hipLaunchKernelGGL(k_guard_check, dim3(1), dim3(64), 0, st, guard, overflow_flag, b);
    T2P_CHECK_LAUNCH("guard_check");
    return 0;
}

int launch_rownorm(const float* in, int ld_in, int64_t n_rows, int dim, float* out, int ld_out, int col0,
                   hipStream_t st) {
    if (n_rows == 0) return 0;
    ProfScope ps_("rownorm", st);
    hipLaunchKernelGGL(k_rownorm, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, st, in, ld_in, n_rows, dim, out,
                       ld_out, col0);
    T2P_CHECK_LAUNCH("rownorm");
    return 0;
}

int launch_gather_rownorm(const float* table, const int32_t* idx, int64_t n_rows, int dim, float* out, int ld_out,
                          int col0, hipStream_t st) {
    if (n_rows == 0) return 0;
    ProfScope ps_("gather_rownorm", st);
    hipLaunchKernelGGL(k_gather_rownorm, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, st, table, idx, n_rows, dim,
                       out, ld_out, col0);
    T2P_CHECK_LAUNCH("gather_rownorm");
    return 0;
}

int launch_segmax(const float* in, int dim, const int32_t* seg_ptr, int n_seg, float* out, int mean, hipStream_t st) {
    if (n_seg == 0) return 0;
    ProfScope ps_("segpool", st);
    hipLaunchKernelGGL(k_segpool, dim3(n_seg), dim3(256), 0, st, in, dim, seg_ptr, n_seg, out, mean);
    T2P_CHECK_LAUNCH("segpool");
    return 0;
}

int launch_mlp3_norm(const float* in3, int64_t n_rows, const float* w1, const float* b1, const float* w2,
                     const float* b2, int D, float* out, int ld_out, int col0, hipStream_t st) {
    T2P_CHECK_ARG(D % 64 == 0 && D <= 512, "mlp3_norm: D=%d must be a multiple of 64, <= 512", D);
    if (n_rows == 0) return 0;
    ProfScope ps_("mlp3_norm", st);
    if (n_rows >= 16384)
        hipLaunchKernelGGL(k_mlp3_norm<32>, dim3((unsigned)((n_rows + 31) / 32)), dim3(256), 0, st, in3, n_rows, w1, b1, w2, b2, D, out, ld_out,
                           col0);
    else
        hipLaunchKernelGGL(k_mlp3_norm<8>, dim3((unsigned)((n_rows + 7) / 8)), dim3(256), 0, st, in3, n_rows, w1, b1, w2, b2, D, out, ld_out,
                           col0);
    T2P_CHECK_LAUNCH("mlp3_norm");
    return 0;
}

int launch_pack_objects(const float* raw_xyz, const float* raw_rgb, const int32_t* obj_ptr, const int32_t* sample_idx,
                        const float* rot, int64_t n_obj, int n_pts, float* xyz, float* rgb, float* center, float* mean_rgb,
                        hipStream_t st) {
    T2P_CHECK_ARG(n_pts <= 1024, "pack_objects: n_pts=%d > 1024 (the sampled points of an object are staged in LDS)", n_pts);
    if (n_obj == 0) return 0;
    ProfScope ps_("pack_objects", st);
    hipLaunchKernelGGL(k_pack_objects, dim3((unsigned)((n_obj + 3) / 4)), dim3(256), (size_t)4 * n_pts * 3 * sizeof(float), st, raw_xyz,
                       raw_rgb, obj_ptr, sample_idx, rot, n_obj, n_pts, xyz, rgb, center, mean_rgb);
    T2P_CHECK_LAUNCH("pack_objects");
    return 0;
}

int launch_pack_scene(const float* raw_xyz, const float* raw_rgb, const int32_t* obj_ptr, const int32_t* obj_id, const uint64_t* key,
                      const float* scene_center, const float* scene_color, int64_t n_out, int n_pts, float* xyz, float* rgb,
                      float* center, float* mean_rgb, int32_t* idx_out, hipStream_t st) {
    T2P_CHECK_ARG(n_pts <= 1024, "pack_scene_objects: n_pts=%d > 1024 (the sampled points of an object are staged in LDS)", n_pts);
    if (n_out == 0) return 0;
    ProfScope ps_("pack_scene", st);
    hipLaunchKernelGGL(k_pack_scene, dim3((unsigned)((n_out + 3) / 4)), dim3(256), (size_t)4 * n_pts * 3 * sizeof(float), st, raw_xyz,
                       raw_rgb, obj_ptr, obj_id, key, scene_center, scene_color, n_out, n_pts, xyz, rgb, center, mean_rgb, idx_out);
    T2P_CHECK_LAUNCH("pack_scene");
    return 0;
}

int launch_pairwise_ranking(const float* scores, int batch, float margin, float* row_loss, float* d_scores, float* row_cnt,
                            hipStream_t st) {
    if (batch == 0) return 0;
    hipLaunchKernelGGL(k_rank_rows, dim3((unsigned)batch), dim3(256), 0, st, scores, batch, margin, row_loss, d_scores, row_cnt);
    T2P_CHECK_LAUNCH("rank_rows");
    hipLaunchKernelGGL(k_rank_diag, dim3((unsigned)batch), dim3(256), 0, st, scores, batch, margin, row_cnt, d_scores);
    T2P_CHECK_LAUNCH("rank_diag");
    return 0;
}

int launch_hardest_ranking(const float* scores, int batch, float margin, float* best, int32_t* where, float* d_scores,
                           hipStream_t st) {
    if (batch == 0) return 0;
    hipLaunchKernelGGL(k_hardest_scan, dim3((unsigned)(2 * batch)), dim3(256), 0, st, scores, batch, margin, best, where);
    T2P_CHECK_LAUNCH("hardest_scan");
    hipError_t e = hipMemsetAsync(d_scores, 0, sizeof(float) * (size_t)batch * batch, st);
    if (e != hipSuccess) {
        set_error("hardest_ranking: memset failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    hipLaunchKernelGGL(k_hardest_grad, dim3((unsigned)((2 * batch + 255) / 256)), dim3(256), 0, st, where, batch, d_scores);
    T2P_CHECK_LAUNCH("hardest_grad");
    return 0;
}

int launch_cell_index(const int32_t* cell_ptr, int n_cells, int32_t o_lo, int32_t* seg_ptr_local, int32_t* first,
                      hipStream_t st, uint32_t* guard_to_clear) {
    ProfScope ps_("cell_index", st);
    hipLaunchKernelGGL(k_cell_index, dim3((unsigned)((n_cells + 1 + 255) / 256)), dim3(256), 0, st, cell_ptr, n_cells,
                       o_lo, seg_ptr_local, first, guard_to_clear);
    T2P_CHECK_LAUNCH("cell_index");
    return 0;
}

int launch_knn(const float* x, int dim, const int32_t* seg_ptr, int n_seg, int max_seg_rows, int k, int32_t* out_idx,
               hipStream_t st) {
    // all-pairs distance matrix of the largest segment lives in dynamic LDS (<= 192 rows -> 144 KiB)
    if (n_seg == 0) return 0;
    const int kMaxRows = 192;
    T2P_CHECK_ARG(max_seg_rows >= 0 && max_seg_rows <= kMaxRows, "knn: a cell with %d objects exceeds the %d-row limit",
                  max_seg_rows, kMaxRows);
    size_t lds = (size_t)(((max_seg_rows > 0 ? max_seg_rows : 1) * max_seg_rows + 3) & ~3) * sizeof(float);
    // cells of up to kKnnStage rows keep a copy of their rows in LDS behind the distance matrix, when both fit
    int stage_arg = 0;
    if (dim <= 512 && (dim & 3) == 0) {
        const int stage_rows = max_seg_rows < kKnnStage ? max_seg_rows : kKnnStage;
        const size_t extra = (size_t)stage_rows * (dim + 4) * sizeof(float);
        if (lds + extra <= 160 * 1024) {
            lds += extra;
            stage_arg = stage_rows;
        }
    }
    T2P_TRY(reserve_lds((const void*)k_knn, 160 * 1024, "knn"));
    ProfScope ps_("knn", st);
    hipLaunchKernelGGL(k_knn, dim3(n_seg), dim3(256), lds, st, x, dim, seg_ptr, k, out_idx, stage_arg);
    T2P_CHECK_LAUNCH("knn");
    return 0;
}

}  // namespace t2p

// Training-mode building blocks of the cell branch (second part; the text branch is in lstm.hip).
//
// The inference kernels fold every BatchNorm into its Linear layer.  In training mode (model.train(),
// training/coarse.py:32) BatchNorm1d normalises with the statistics of the CURRENT rows and the reference calls its PointNet++
// once per cell (models/object_encoder.py:92-95), so the statistics of those layers are per cell: "segments" below.  Every
// Linear + BatchNorm1d + ReLU block of models/modules.py:21-29 becomes  t2p_gemm -> t2p_bn_relu_train_forward;  the
// max-aggregations (PointConv aggr="max", global_max_pool, DynamicEdgeConv aggr="max") are segment maxima over rows that are
// already sorted by destination, with the winning row remembered for the backward pass.
// First correct path: plain kernels, one block per (segment, 64 columns); no fusion with the GEMMs yet.
#include "t2p_common.h"

namespace t2p {
namespace {

constexpr int kCols = 64;   // columns per block
constexpr int kRowsPar = 4; // row lanes per block (256 threads)

// fixed-order combination of the kRowsPar partial results of a column
__device__ __forceinline__ double combine4(double (*red)[kCols], int rl, int cl, double v) {
    red[rl][cl] = v;
    __syncthreads();
    const double r = ((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]));
    __syncthreads();
    return r;
}

// BatchNorm output of one element, the ONE place that spells the formula: the backward kernels recompute the ReLU mask
// (y > 0) from x with it instead of reading y back (two of the seven tensor passes of the backward), so it must give the
// forward's bits (-ffp-contract=off: four separate roundings, in this order)
__device__ __forceinline__ float bn_out(float x, float mean, float invstd, float gamma, float beta) {
    return (x - mean) * invstd * gamma + beta;
}

// dz = dy where the ReLU let the activation through (or everywhere without a ReLU), else 0 - as a bit mask.  Written as the
// select `(!relu || bn_out(...) > 0.f) ? dy : 0.f`, hipcc (ROCm 7.2, -O3) compiled k_bn_bwd_apply4 to `v_mov dz, 0` for ALL lanes
// followed by an EMPTY `s_and_saveexec` region where the move back belonged: dx came out as if no row had passed the ReLU
// (the statistics kernel next to it, same expression, was compiled correctly).  test_bn_relu_train_matches_torch_per_segment
// catches it; the mask form below leaves no branch to get wrong.
__device__ __forceinline__ float relu_gate(float dy, int relu, float x, float mean, float invstd, float gamma, float beta) {
    const int keep = (relu == 0) | (int)(bn_out(x, mean, invstd, gamma, beta) > 0.f);
    return __int_as_float(__float_as_int(dy) & -keep);
}

// Row chunk z of gridDim.z of segment s: [lo, hi)
__device__ __forceinline__ void chunk_rows(const int32_t* seg_ptr, int s, int& lo, int& hi, int& n) {
    const int r0 = seg_ptr[s], r1 = seg_ptr[s + 1];
    n = r1 - r0;
    const int per = (n + (int)gridDim.z - 1) / (int)gridDim.z;
    lo = r0 + (int)blockIdx.z * per;
    hi = lo + per < r1 ? lo + per : r1;
}

// stage 1 of the statistics: per (segment, row chunk) partial sums of x and x^2 in float64; part [n_seg][R][2][C]
__global__ __launch_bounds__(256) void k_bn_partial(const float* __restrict__ x, const int32_t* __restrict__ seg_ptr, int C,
                                                    double* __restrict__ part) {
    __shared__ double red[kRowsPar][kCols];
    const int s = blockIdx.x, cl = threadIdx.x % kCols, rl = threadIdx.x / kCols;
    const int c = blockIdx.y * kCols + cl;
    int lo, hi, n;
    chunk_rows(seg_ptr, s, lo, hi, n);
    const bool ok = c < C;
    double a0 = 0.0, a1 = 0.0;
    for (int r = lo + rl; r < hi; r += kRowsPar) {
        const double v = ok ? (double)x[(int64_t)r * C + c] : 0.0;
        a0 += v;
        a1 += v * v;
    }
    const double s0 = combine4(red, rl, cl, a0);
    const double s1 = combine4(red, rl, cl, a1);
    if (ok && rl == 0) {
        double* p = part + (((int64_t)s * gridDim.z + blockIdx.z) * 2) * C;
        p[c] = s0;
        p[C + c] = s1;
    }
}
// stage 2: chunks combined in a fixed order -> mean, 1/sqrt(var + eps) (biased variance, as BatchNorm1d normalises), and
// the unbiased variance for the running estimate
__global__ void k_bn_finish(const double* __restrict__ part, const int32_t* __restrict__ seg_ptr, int n_seg, int C, int R,
                            float eps, float* __restrict__ mean, float* __restrict__ invstd,
                            float* __restrict__ var_unbiased) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= (int64_t)n_seg * C) return;
    const int s = (int)(i / C), c = (int)(i % C);
    const int n = seg_ptr[s + 1] - seg_ptr[s];
    double s0 = 0.0, s1 = 0.0;
    for (int z = 0; z < R; z++) {
        const double* p = part + (((int64_t)s * R + z) * 2) * C;
        s0 += p[c];
        s1 += p[C + c];
    }
    const double m = n > 0 ? s0 / n : 0.0;
    double ss = s1 - s0 * m;  // sum (x - m)^2
    if (ss < 0.0) ss = 0.0;
    const double var = n > 0 ? ss / n : 0.0;
    mean[i] = (float)m;
    invstd[i] = (float)(1.0 / sqrt(var + (double)eps));
    var_unbiased[i] = (float)(n > 1 ? ss / (n - 1) : var);
}

// y = act(gamma (x - mean) invstd + beta) for the rows of chunk blockIdx.z of segment blockIdx.x
__global__ __launch_bounds__(256) void k_bn_apply(const float* __restrict__ x, const int32_t* __restrict__ seg_ptr, int C,
                                                  const float* __restrict__ mean, const float* __restrict__ invstd,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta, int relu,
                                                  float* __restrict__ y) {
    const int s = blockIdx.x, cl = threadIdx.x % kCols, rl = threadIdx.x / kCols;
    const int c = blockIdx.y * kCols + cl;
    if (c >= C) return;
    int lo, hi, n;
    chunk_rows(seg_ptr, s, lo, hi, n);
    const float m = mean[(int64_t)s * C + c], is = invstd[(int64_t)s * C + c], g = gamma[c], b = beta[c];
    for (int r = lo + rl; r < hi; r += kRowsPar) {
        const float v = bn_out(x[(int64_t)r * C + c], m, is, g, b);
        y[(int64_t)r * C + c] = relu ? fmaxf(v, 0.f) : v;
    }
}

// backward, stage 1: per chunk partial sums of dz = dy (bn_out(x) > 0) and dz xhat; part [n_seg][R][2][C]
__global__ __launch_bounds__(256) void k_bn_bwd_partial(const float* __restrict__ dy, const float* __restrict__ x,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const int32_t* __restrict__ seg_ptr, int C, const float* __restrict__ mean,
                                                        const float* __restrict__ invstd, int relu,
                                                        double* __restrict__ part) {
    __shared__ double red[kRowsPar][kCols];
    const int s = blockIdx.x, cl = threadIdx.x % kCols, rl = threadIdx.x / kCols;
    const int c = blockIdx.y * kCols + cl;
    int lo, hi, n;
    chunk_rows(seg_ptr, s, lo, hi, n);
    const bool ok = c < C;
    const float m = ok ? mean[(int64_t)s * C + c] : 0.f, is = ok ? invstd[(int64_t)s * C + c] : 0.f;
    const float g = ok ? gamma[c] : 0.f, b = ok ? beta[c] : 0.f;
    double a0 = 0.0, a1 = 0.0;
    if (ok)
        for (int r = lo + rl; r < hi; r += kRowsPar) {
            const int64_t i = (int64_t)r * C + c;
            const float dz = relu_gate(dy[i], relu, x[i], m, is, g, b);
            a0 += (double)dz;
            a1 += (double)dz * (double)((x[i] - m) * is);
        }
    const double s0 = combine4(red, rl, cl, a0);
    const double s1 = combine4(red, rl, cl, a1);
    if (ok && rl == 0) {
        double* p = part + (((int64_t)s * gridDim.z + blockIdx.z) * 2) * C;
        p[c] = s0;
        p[C + c] = s1;
    }
}
// stage 2: per-segment dbeta = sum dz, dgamma = sum dz xhat (chunks in a fixed order)
__global__ void k_bn_bwd_finish(const double* __restrict__ part, int n_seg, int C, int R, float* __restrict__ dgamma_seg,
                                float* __restrict__ dbeta_seg) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= (int64_t)n_seg * C) return;
    const int s = (int)(i / C), c = (int)(i % C);
    double s0 = 0.0, s1 = 0.0;
    for (int z = 0; z < R; z++) {
        const double* p = part + (((int64_t)s * R + z) * 2) * C;
        s0 += p[c];
        s1 += p[C + c];
    }
    dbeta_seg[i] = (float)s0;
    dgamma_seg[i] = (float)s1;
}
// stage 2b: row n_seg of both tables = the sum over the segments (the layer's weight / bias gradient), float64, segment order
__global__ void k_bn_bwd_total(int n_seg, int C, float* __restrict__ dgamma_seg, float* __restrict__ dbeta_seg) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double t0 = 0.0, t1 = 0.0;
    for (int s = 0; s < n_seg; s++) {
        t0 += (double)dbeta_seg[(int64_t)s * C + c];
        t1 += (double)dgamma_seg[(int64_t)s * C + c];
    }
    dbeta_seg[(int64_t)n_seg * C + c] = (float)t0;
    dgamma_seg[(int64_t)n_seg * C + c] = (float)t1;
}
// stage 3: dx = gamma invstd (dz - sum dz / n - xhat sum(dz xhat) / n)
__global__ __launch_bounds__(256) void k_bn_bwd_apply(const float* __restrict__ dy, const float* __restrict__ x,
                                                      const float* __restrict__ beta, const int32_t* __restrict__ seg_ptr, int C,
                                                      const float* __restrict__ mean, const float* __restrict__ invstd,
                                                      const float* __restrict__ gamma, int relu,
                                                      const float* __restrict__ dgamma_seg,
                                                      const float* __restrict__ dbeta_seg, float* __restrict__ dx) {
    const int s = blockIdx.x, cl = threadIdx.x % kCols, rl = threadIdx.x / kCols;
    const int c = blockIdx.y * kCols + cl;
    if (c >= C) return;
    int lo, hi, n;
    chunk_rows(seg_ptr, s, lo, hi, n);
    const float m = mean[(int64_t)s * C + c], is = invstd[(int64_t)s * C + c], g = gamma[c], b = beta[c];
    const float inv_n = n > 0 ? 1.f / (float)n : 0.f;
    const float sdz = dbeta_seg[(int64_t)s * C + c] * inv_n, sdx = dgamma_seg[(int64_t)s * C + c] * inv_n;
    for (int r = lo + rl; r < hi; r += kRowsPar) {
        const int64_t i = (int64_t)r * C + c;
        const float dz = relu_gate(dy[i], relu, x[i], m, is, g, b);
        dx[i] = g * is * (dz - sdz - (x[i] - m) * is * sdx);
    }
}

// ---- the same four kernels with 16-byte accesses (C % 4 == 0: every BatchNorm width of the model) -------------------------
// The scalar forms above move 4 bytes per lane and keep ONE load in flight per thread: 0.7-0.9 TB/s on the edge tensors of a
// 64-cell training step (2.6 GB of activations per pass), 34 of the step's 69 ms.  Here a thread owns a column QUAD (float4), a
// block is TQ quads x (256 / TQ) row lanes (TQ = 8 for C = 32, else 16 = 64 columns), and the row loop is unrolled by four:
// four 16-byte loads per input in flight per thread.  Reductions keep a fixed order (per thread its rows in ascending order,
// then the row lanes 0 .. RL-1, then the chunks): deterministic, float64.
template <int RL, int W>
__device__ __forceinline__ void reduce_lanes(double (*red)[W + 1], int rl, int c0, const double (&v)[4], double (&out)[4]) {
#pragma unroll
    for (int e = 0; e < 4; e++) red[rl][c0 + e] = v[e];
    __syncthreads();
    if (rl == 0) {
#pragma unroll
        for (int e = 0; e < 4; e++) {
            double a = 0.0;
            for (int k = 0; k < RL; k++) a += red[k][c0 + e];
            out[e] = a;
        }
    }
    __syncthreads();
}

template <int TQ>
__global__ __launch_bounds__(256) void k_bn_partial4(const float* __restrict__ x, const int32_t* __restrict__ seg_ptr, int C,
                                                     double* __restrict__ part) {
    constexpr int RL = 256 / TQ, W = TQ * 4;
    __shared__ double red[RL][W + 1];
    const int s = blockIdx.x, cq = threadIdx.x % TQ, rl = threadIdx.x / TQ;
    const int c = (blockIdx.y * TQ + cq) * 4;
    int lo, hi, n;
    chunk_rows(seg_ptr, s, lo, hi, n);
    const bool ok = c < C;
    double a0[4] = {0.0, 0.0, 0.0, 0.0}, a1[4] = {0.0, 0.0, 0.0, 0.0};
    if (ok) {
        const float* px = x + c;
        int r = lo + rl;
        for (; r + 3 * RL < hi; r += 4 * RL) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = *(const f32x4*)(px + (int64_t)(r + u * RL) * C);
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const double d = (double)v[u][e];
                    a0[e] += d;
                    a1[e] += d * d;
                }
        }
        for (; r < hi; r += RL) {
            const f32x4 v = *(const f32x4*)(px + (int64_t)r * C);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const double d = (double)v[e];
                a0[e] += d;
                a1[e] += d * d;
            }
        }
    }
    double s0[4], s1[4];
    reduce_lanes<RL, W>(red, rl, cq * 4, a0, s0);
    reduce_lanes<RL, W>(red, rl, cq * 4, a1, s1);
    if (ok && rl == 0) {
        double* p = part + (((int64_t)s * gridDim.z + blockIdx.z) * 2) * C;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            p[c + e] = s0[e];
            p[C + c + e] = s1[e];
        }
    }
}

template <int TQ>
__global__ __launch_bounds__(256) void k_bn_apply4(const float* __restrict__ x, const int32_t* __restrict__ seg_ptr, int C,
                                                   const float* __restrict__ mean, const float* __restrict__ invstd,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta, int relu,
                                                   float* __restrict__ y) {
    constexpr int RL = 256 / TQ;
    const int s = blockIdx.x, cq = threadIdx.x % TQ, rl = threadIdx.x / TQ;
    const int c = (blockIdx.y * TQ + cq) * 4;
    if (c >= C) return;
    int lo, hi, n;
    chunk_rows(seg_ptr, s, lo, hi, n);
    const f32x4 m = *(const f32x4*)(mean + (int64_t)s * C + c), is = *(const f32x4*)(invstd + (int64_t)s * C + c);
    const f32x4 g = *(const f32x4*)(gamma + c), b = *(const f32x4*)(beta + c);
    auto one = [&](const f32x4& v) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const float t = bn_out(v[e], m[e], is[e], g[e], b[e]);
            o[e] = relu ? fmaxf(t, 0.f) : t;
        }
        return o;
    };
    int r = lo + rl;
    for (; r + 3 * RL < hi; r += 4 * RL) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = *(const f32x4*)(x + (int64_t)(r + u * RL) * C + c);
#pragma unroll
        for (int u = 0; u < 4; u++) *(f32x4*)(y + (int64_t)(r + u * RL) * C + c) = one(v[u]);
    }
    for (; r < hi; r += RL) *(f32x4*)(y + (int64_t)r * C + c) = one(*(const f32x4*)(x + (int64_t)r * C + c));
}

template <int TQ>
__global__ __launch_bounds__(256) void k_bn_bwd_partial4(const float* __restrict__ dy, const float* __restrict__ x,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const int32_t* __restrict__ seg_ptr, int C, const float* __restrict__ mean,
                                                         const float* __restrict__ invstd, int relu,
                                                         double* __restrict__ part) {
    constexpr int RL = 256 / TQ, W = TQ * 4;
    __shared__ double red[RL][W + 1];
    const int s = blockIdx.x, cq = threadIdx.x % TQ, rl = threadIdx.x / TQ;
    const int c = (blockIdx.y * TQ + cq) * 4;
    int lo, hi, n;
    chunk_rows(seg_ptr, s, lo, hi, n);
    const bool ok = c < C;
    double a0[4] = {0.0, 0.0, 0.0, 0.0}, a1[4] = {0.0, 0.0, 0.0, 0.0};
    if (ok) {
        const f32x4 m = *(const f32x4*)(mean + (int64_t)s * C + c), is = *(const f32x4*)(invstd + (int64_t)s * C + c);
        const f32x4 g = *(const f32x4*)(gamma + c), b = *(const f32x4*)(beta + c);
        auto add = [&](const f32x4& vdy, const f32x4& vx) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float dz = relu_gate(vdy[e], relu, vx[e], m[e], is[e], g[e], b[e]);
                a0[e] += (double)dz;
                a1[e] += (double)dz * (double)((vx[e] - m[e]) * is[e]);
            }
        };
        int r = lo + rl;
        for (; r + RL < hi; r += 2 * RL) {
            const int64_t i0 = (int64_t)r * C + c, i1 = (int64_t)(r + RL) * C + c;
            const f32x4 d0 = *(const f32x4*)(dy + i0), x0 = *(const f32x4*)(x + i0);
            const f32x4 d1 = *(const f32x4*)(dy + i1), x1 = *(const f32x4*)(x + i1);
            add(d0, x0);
            add(d1, x1);
        }
        for (; r < hi; r += RL) {
            const int64_t i0 = (int64_t)r * C + c;
            add(*(const f32x4*)(dy + i0), *(const f32x4*)(x + i0));
        }
    }
    double s0[4], s1[4];
    reduce_lanes<RL, W>(red, rl, cq * 4, a0, s0);
    reduce_lanes<RL, W>(red, rl, cq * 4, a1, s1);
    if (ok && rl == 0) {
        double* p = part + (((int64_t)s * gridDim.z + blockIdx.z) * 2) * C;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            p[c + e] = s0[e];
            p[C + c + e] = s1[e];
        }
    }
}

template <int TQ>
__global__ __launch_bounds__(256) void k_bn_bwd_apply4(const float* __restrict__ dy, const float* __restrict__ x,
                                                       const float* __restrict__ beta, const int32_t* __restrict__ seg_ptr, int C,
                                                       const float* __restrict__ mean, const float* __restrict__ invstd,
                                                       const float* __restrict__ gamma, int relu,
                                                       const float* __restrict__ dgamma_seg,
                                                       const float* __restrict__ dbeta_seg, float* __restrict__ dx) {
    constexpr int RL = 256 / TQ;
    const int s = blockIdx.x, cq = threadIdx.x % TQ, rl = threadIdx.x / TQ;
    const int c = (blockIdx.y * TQ + cq) * 4;
    if (c >= C) return;
    int lo, hi, n;
    chunk_rows(seg_ptr, s, lo, hi, n);
    const int64_t sc = (int64_t)s * C + c;
    const f32x4 m = *(const f32x4*)(mean + sc), is = *(const f32x4*)(invstd + sc), g = *(const f32x4*)(gamma + c);
    const f32x4 b = *(const f32x4*)(beta + c);
    const float inv_n = n > 0 ? 1.f / (float)n : 0.f;
    const f32x4 sdz = *(const f32x4*)(dbeta_seg + sc) * inv_n, sdx = *(const f32x4*)(dgamma_seg + sc) * inv_n;
    auto one = [&](const f32x4& vdy, const f32x4& vx) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const float dz = relu_gate(vdy[e], relu, vx[e], m[e], is[e], g[e], b[e]);
            o[e] = g[e] * is[e] * (dz - sdz[e] - (vx[e] - m[e]) * is[e] * sdx[e]);
        }
        return o;
    };
    int r = lo + rl;
    for (; r + RL < hi; r += 2 * RL) {
        const int64_t i0 = (int64_t)r * C + c, i1 = (int64_t)(r + RL) * C + c;
        const f32x4 d0 = *(const f32x4*)(dy + i0), x0 = *(const f32x4*)(x + i0);
        const f32x4 d1 = *(const f32x4*)(dy + i1), x1 = *(const f32x4*)(x + i1);
        *(f32x4*)(dx + i0) = one(d0, x0);
        *(f32x4*)(dx + i1) = one(d1, x1);
    }
    for (; r < hi; r += RL) {
        const int64_t i0 = (int64_t)r * C + c;
        *(f32x4*)(dx + i0) = one(*(const f32x4*)(dy + i0), *(const f32x4*)(x + i0));
    }
}

// out[s][c] = max over the rows of segment s (first row wins ties), arg[s][c] = that row (-1: empty segment, out = 0)
__global__ __launch_bounds__(256) void k_segment_max(const float* __restrict__ x, const int32_t* __restrict__ seg_ptr, int C,
                                                     float* __restrict__ out, int32_t* __restrict__ arg) {
    __shared__ float bv[kRowsPar][kCols];
    __shared__ int bi[kRowsPar][kCols];
    const int s = blockIdx.x, cl = threadIdx.x % kCols, rl = threadIdx.x / kCols;
    const int c = blockIdx.y * kCols + cl;
    const bool ok = c < C;
    float best = -INFINITY;
    int who = -1;
    for (int r = seg_ptr[s] + rl; r < seg_ptr[s + 1]; r += kRowsPar) {
        const float v = ok ? x[(int64_t)r * C + c] : 0.f;
        if (v > best) { best = v; who = r; }
    }
    bv[rl][cl] = best;
    bi[rl][cl] = who;
    __syncthreads();
    if (ok && rl == 0) {
        for (int k = 1; k < kRowsPar; k++)
            if (bi[k][cl] >= 0 && (who < 0 || bv[k][cl] > best || (bv[k][cl] == best && bi[k][cl] < who))) {
                best = bv[k][cl];
                who = bi[k][cl];
            }
        out[(int64_t)s * C + c] = who >= 0 ? best : 0.f;
        arg[(int64_t)s * C + c] = who;
    }
}

// dx = 0 except dx[arg[s][c]][c] = dout[s][c]
__global__ __launch_bounds__(256) void k_segment_max_backward(const float* __restrict__ dout, const int32_t* __restrict__ arg,
                                                              const int32_t* __restrict__ seg_ptr, int C,
                                                              float* __restrict__ dx) {
    const int s = blockIdx.x, cl = threadIdx.x % kCols, rl = threadIdx.x / kCols;
    const int c = blockIdx.y * kCols + cl;
    if (c >= C) return;
    const int who = arg[(int64_t)s * C + c];
    const float g = dout[(int64_t)s * C + c];
    for (int r = seg_ptr[s] + rl; r < seg_ptr[s + 1]; r += kRowsPar) dx[(int64_t)r * C + c] = r == who ? g : 0.f;
}

// out[s][c] = mean over the rows of segment s (0 for an empty one); fixed order (4 row lanes, then combined)
__global__ __launch_bounds__(256) void k_segment_mean(const float* __restrict__ x, const int32_t* __restrict__ seg_ptr, int C,
                                                      float* __restrict__ out) {
    __shared__ double red[kRowsPar][kCols];
    const int s = blockIdx.x, cl = threadIdx.x % kCols, rl = threadIdx.x / kCols;
    const int c = blockIdx.y * kCols + cl;
    const int r0 = seg_ptr[s], r1 = seg_ptr[s + 1];
    const bool ok = c < C;
    double acc = 0.0;
    for (int r = r0 + rl; r < r1; r += kRowsPar) acc += ok ? (double)x[(int64_t)r * C + c] : 0.0;
    const double t = combine4(red, rl, cl, acc);
    if (ok && rl == 0) out[(int64_t)s * C + c] = r1 > r0 ? (float)(t / (double)(r1 - r0)) : 0.f;
}
// dx[r][c] = dout[s][c] / n_s for the rows of segment s
__global__ __launch_bounds__(256) void k_segment_mean_backward(const float* __restrict__ dout,
                                                               const int32_t* __restrict__ seg_ptr, int C,
                                                               float* __restrict__ dx) {
    const int s = blockIdx.x, cl = threadIdx.x % kCols, rl = threadIdx.x / kCols;
    const int c = blockIdx.y * kCols + cl;
    if (c >= C) return;
    const int r0 = seg_ptr[s], r1 = seg_ptr[s + 1];
    const float g = r1 > r0 ? dout[(int64_t)s * C + c] / (float)(r1 - r0) : 0.f;
    for (int r = r0 + rl; r < r1; r += kRowsPar) dx[(int64_t)r * C + c] = g;
}

// PointConv message input of every edge: out[e] = [x[src[e]] | pos[src[e]] - pos_c[dst[e]]]  (models/pointcloud/pointnet2.py:31-35:
// cat([x_j, pos_j - pos_i])); one thread per output element
__global__ void k_edge_feat_fwd(const float* __restrict__ x, const float* __restrict__ pos, const float* __restrict__ pos_c,
                                const int32_t* __restrict__ src, const int32_t* __restrict__ dst, int64_t E, int C, int W,
                                float* __restrict__ out) {
    // W >= C + 3: row pitch of out; the columns behind the message are written as zeros (a pitch that is a multiple of 8 hands the
    // Linear layer behind it an operand it need not pad: the copy of [E, 67] into [E, 72] was 0.1-0.15 ms per level and direction)
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= E * W) return;
    const int64_t e = i / W;
    const int c = (int)(i % W);
    const int s = src[e];
    float v = 0.f;
    if (c < C) v = x[(int64_t)s * C + c];
    else if (c < C + 3) v = pos[(int64_t)s * 3 + (c - C)] - pos_c[(int64_t)dst[e] * 3 + (c - C)];
    out[i] = v;
}
// its backward with respect to x (positions are inputs): dx[src[e]] += dout[e][:C]; dx zeroed by the caller
__global__ void k_edge_feat_bwd(const float* __restrict__ dout, const int32_t* __restrict__ src, int64_t E, int C, int W,
                                float* __restrict__ dx) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= E * C) return;
    const int64_t e = i / C;
    const int c = (int)(i % C);
    atomicAdd(dx + (int64_t)src[e] * C + c, dout[e * W + c]);
}
// DynamicEdgeConv message input: out[e] = [x[tgt[e]] | x[src[e]] - x[tgt[e]]]  (models/cell_retrieval.py:46-48)
__global__ void k_pair_feat_fwd(const float* __restrict__ x, const int32_t* __restrict__ tgt, const int32_t* __restrict__ src,
                                int64_t E, int D, float* __restrict__ out) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= E * 2 * D) return;
    const int64_t e = i / (2 * D);
    const int c = (int)(i % (2 * D));
    const float xt = x[(int64_t)tgt[e] * D + (c % D)];
    out[i] = c < D ? xt : x[(int64_t)src[e] * D + (c - D)] - xt;
}
// dx[tgt] += dA - dB, dx[src] += dB
__global__ void k_pair_feat_bwd(const float* __restrict__ dout, const int32_t* __restrict__ tgt,
                                const int32_t* __restrict__ src, int64_t E, int D, float* __restrict__ dx) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= E * D) return;
    const int64_t e = i / D;
    const int c = (int)(i % D);
    const float da = dout[e * 2 * D + c], db = dout[e * 2 * D + D + c];
    atomicAdd(dx + (int64_t)tgt[e] * D + c, da - db);
    atomicAdd(dx + (int64_t)src[e] * D + c, db);
}
// backward of F.normalize(x, dim=-1) (eps 1e-12): dx = (dy - x_n <x_n, dy>) / max(|x|, eps); one wavefront per row
__global__ __launch_bounds__(256) void k_rownorm_bwd(const float* __restrict__ x, const float* __restrict__ dy, int64_t n_rows,
                                                     int dim, float* __restrict__ dx) {
    const int lane = threadIdx.x & 63;
    const int64_t r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    const float* xr = x + r * dim;
    const float* gr = dy + r * dim;
    float ss = 0.f, dot = 0.f;
    for (int c = lane; c < dim; c += 64) {
        ss += xr[c] * xr[c];
        dot += xr[c] * gr[c];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        ss += __shfl_xor(ss, off, 64);
        dot += __shfl_xor(dot, off, 64);
    }
    const float nrm = fmaxf(sqrtf(ss), 1e-12f), inv = 1.f / nrm;
    const float proj = dot * inv * inv;  // <x_n, dy> / |x|
    for (int c = lane; c < dim; c += 64) dx[r * dim + c] = (gr[c] - xr[c] * proj) * inv;
}

}  // namespace

int launch_edge_feat_fwd(const float* x, const float* pos, const float* pos_c, const int32_t* src, const int32_t* dst, int64_t E,
                         int C, int W, float* out, hipStream_t st) {
    T2P_CHECK_ARG(W >= C + 3, "edge_features: width %d < channels + 3 = %d", W, C + 3);
    const int64_t n = E * W;
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_edge_feat_fwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, pos, pos_c, src, dst, E, C, W, out);
    T2P_CHECK_LAUNCH("edge_feat_fwd");
    return 0;
}
int launch_edge_feat_bwd(const float* dout, const int32_t* src, int64_t E, int C, int W, float* dx, hipStream_t st) {
    T2P_CHECK_ARG(W >= C + 3, "edge_features: width %d < channels + 3 = %d", W, C + 3);
    const int64_t n = E * C;
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_edge_feat_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dout, src, E, C, W, dx);
    T2P_CHECK_LAUNCH("edge_feat_bwd");
    return 0;
}
int launch_pair_feat_fwd(const float* x, const int32_t* tgt, const int32_t* src, int64_t E, int D, float* out, hipStream_t st) {
    const int64_t n = E * 2 * D;
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_pair_feat_fwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, tgt, src, E, D, out);
    T2P_CHECK_LAUNCH("pair_feat_fwd");
    return 0;
}
int launch_pair_feat_bwd(const float* dout, const int32_t* tgt, const int32_t* src, int64_t E, int D, float* dx, hipStream_t st) {
    const int64_t n = E * D;
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_pair_feat_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dout, tgt, src, E, D, dx);
    T2P_CHECK_LAUNCH("pair_feat_bwd");
    return 0;
}
int launch_segment_mean(const float* x, const int32_t* seg_ptr, int n_seg, int C, float* out, hipStream_t st) {
    if (n_seg == 0) return 0;
    hipLaunchKernelGGL(k_segment_mean, dim3((unsigned)n_seg, (unsigned)((C + kCols - 1) / kCols)), dim3(256), 0, st, x, seg_ptr,
                       C, out);
    T2P_CHECK_LAUNCH("segment_mean");
    return 0;
}
int launch_segment_mean_backward(const float* dout, const int32_t* seg_ptr, int n_seg, int C, float* dx, hipStream_t st) {
    if (n_seg == 0) return 0;
    hipLaunchKernelGGL(k_segment_mean_backward, dim3((unsigned)n_seg, (unsigned)((C + kCols - 1) / kCols)), dim3(256), 0, st,
                       dout, seg_ptr, C, dx);
    T2P_CHECK_LAUNCH("segment_mean_backward");
    return 0;
}
int launch_rownorm_bwd(const float* x, const float* dy, int64_t n_rows, int dim, float* dx, hipStream_t st) {
    if (n_rows == 0) return 0;
    hipLaunchKernelGGL(k_rownorm_bwd, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, st, x, dy, n_rows, dim, dx);
    T2P_CHECK_LAUNCH("rownorm_bwd");
    return 0;
}

// ---- edge lists of a set-abstraction level from the compact row lists of k_sample_group ------------------------------------
// The training-mode path needs PointConv's edges as (source dense row, target centroid row) arrays sorted by target
// (models/pointcloud/pointnet2.py:26-35 with torch_geometric's self-loop rewrite: remove the edges whose two CELL-local indices
// agree, append (i, i) for every centroid row i of the cell).  k_sample_group already writes exactly that list per object -
// hits in ascending source order, then the self-loop row, centroid after centroid - for the inference kernels; under their
// max-aggregation a hit that duplicates the self loop is harmless, under batch statistics it is not, so it is dropped here.
// First version of the path built the lists with torch tensor ops (nonzero of the hit mask, stable sort, bincounts): ~30
// launches and 3 size read-backs per level; now: one count kernel, one cumsum, one expand kernel per level and ONE read-back
// of the three edge totals.
//   keep(row) = not (self_loops and row is a hit and (o - f) n_dense + src == (o - f) n_cent + centroid)    f = first object of o's cell
__device__ __forceinline__ bool edge_row_kept(uint32_t e16, int rank_in_cell, int n_dense, int n_cent, int self_loops) {
    const int cb = (int)(e16 >> 8), src = (int)(e16 & 0xFFu);
    if (!self_loops || (cb & 0x80)) return true;
    return rank_in_cell * n_dense + src != rank_in_cell * n_cent + cb;
}

// counts [n_obj * n_cent]: kept rows per centroid.  One wave per object.
__global__ __launch_bounds__(64) void k_edge_counts(const uint16_t* __restrict__ rows, const uint16_t* __restrict__ n_rows,
                                                    const int32_t* __restrict__ first_obj, int64_t n_obj, int n_dense, int n_cent,
                                                    int self_loops, int32_t* __restrict__ counts) {
    __shared__ int cnt[128];
    const int lane = threadIdx.x;
    const int pitch = n_cent * 33;
    for (int64_t o = blockIdx.x; o < n_obj; o += gridDim.x) {
        for (int i = lane; i < n_cent; i += 64) cnt[i] = 0;
        __syncthreads();
        const uint16_t* r = rows + o * pitch;
        const int n = n_rows[o], rank = (int)(o - first_obj[o]);
        for (int i = lane; i < n; i += 64) {
            const uint32_t e = r[i];
            if (edge_row_kept(e, rank, n_dense, n_cent, self_loops)) atomicAdd(&cnt[(e >> 8) & 0x7F], 1);
        }
        __syncthreads();
        for (int i = lane; i < n_cent; i += 64) counts[o * n_cent + i] = cnt[i];
        __syncthreads();
    }
}

// src / dst [E] int32: the kept rows of all objects in list order (= sorted by target row); cent_ptr [n_obj * n_cent + 1] is the
// exclusive prefix of `counts`, so an object's rows start at cent_ptr[o n_cent].  One wave per object, 64 rows per round.
__global__ __launch_bounds__(64) void k_edge_expand(const uint16_t* __restrict__ rows, const uint16_t* __restrict__ n_rows,
                                                    const int32_t* __restrict__ first_obj, const int32_t* __restrict__ cent_ptr,
                                                    int64_t n_obj, int n_dense, int n_cent, int self_loops,
                                                    int32_t* __restrict__ src, int32_t* __restrict__ dst) {
    const int lane = threadIdx.x;
    const int pitch = n_cent * 33;
    for (int64_t o = blockIdx.x; o < n_obj; o += gridDim.x) {
        const uint16_t* r = rows + o * pitch;
        const int n = n_rows[o];
        const int64_t f = first_obj[o];
        const int rank = (int)(o - f);
        int64_t base = cent_ptr[o * n_cent];
        for (int i0 = 0; i0 < n; i0 += 64) {
            const int i = i0 + lane;
            const uint32_t e = i < n ? r[i] : 0u;
            const bool keep = i < n && edge_row_kept(e, rank, n_dense, n_cent, self_loops);
            const unsigned long long m = __ballot(keep);
            if (keep) {
                const int before = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
                const int cb = (int)(e >> 8), c = cb & 0x7F, sidx = (int)(e & 0xFFu);
                // a self-loop row names the dense row of the cell's batch with the centroid row's cell-local index
                const int64_t s_row = (cb & 0x80) ? f * n_dense + ((int64_t)rank * n_cent + sidx) : o * n_dense + sidx;
                src[base + before] = (int32_t)s_row;
                dst[base + before] = (int32_t)(o * n_cent + c);
            }
            base += __popcll(m);
        }
    }
}

// the 16-byte forms need C % 4 == 0 and 16-byte aligned rows / per-channel vectors
static bool bn_vec_ok(int C, const void* a, const void* b, const void* c, const void* d, const void* e) {
    return C % 4 == 0 && (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d | (uintptr_t)e) & 15) == 0;
}

// row chunks per segment: ~512 rows per block for balanced segments (64 cells x 4 column blocks x 2 chunks of 4 k rows put two
// blocks on a CU at SA3: the four passes ran at 3.4 TB/s; 1,024 / 512 rows per block: -5 % / -12 % of their time)
#ifndef T2P_BN_CHUNK_ROWS
#define T2P_BN_CHUNK_ROWS 512
#endif
static int bn_chunks(int64_t rows, int n_seg) {
    int64_t r = (rows / (n_seg > 0 ? n_seg : 1) + T2P_BN_CHUNK_ROWS - 1) / T2P_BN_CHUNK_ROWS;
    return (int)(r < 1 ? 1 : (r > 256 ? 256 : r));
}

int launch_bn_relu_train_forward(const float* x, const int32_t* seg_ptr, int n_seg, int64_t rows, int C, const float* gamma,
                                 const float* beta, float eps, int relu, float* y, float* mean, float* invstd,
                                 float* var_unbiased, double* part, hipStream_t st) {
    if (n_seg == 0) return 0;
    const int R = bn_chunks(rows, n_seg);
    const dim3 grid((unsigned)n_seg, (unsigned)((C + kCols - 1) / kCols), (unsigned)R);
    const bool vec = bn_vec_ok(C, x, y, mean, invstd, gamma) && ((uintptr_t)beta & 15) == 0;
    if (vec && C <= 32) hipLaunchKernelGGL(k_bn_partial4<8>, dim3((unsigned)n_seg, (unsigned)((C + 31) / 32), (unsigned)R), dim3(256), 0, st, x, seg_ptr, C, part);
    else if (vec) hipLaunchKernelGGL(k_bn_partial4<16>, grid, dim3(256), 0, st, x, seg_ptr, C, part);
    else hipLaunchKernelGGL(k_bn_partial, grid, dim3(256), 0, st, x, seg_ptr, C, part);
    T2P_CHECK_LAUNCH("bn_partial");
    const int64_t sc = (int64_t)n_seg * C;
    hipLaunchKernelGGL(k_bn_finish, dim3((unsigned)((sc + 255) / 256)), dim3(256), 0, st, part, seg_ptr, n_seg, C, R, eps, mean,
                       invstd, var_unbiased);
    T2P_CHECK_LAUNCH("bn_finish");
    if (vec && C <= 32) hipLaunchKernelGGL(k_bn_apply4<8>, dim3((unsigned)n_seg, (unsigned)((C + 31) / 32), (unsigned)R), dim3(256), 0, st, x, seg_ptr, C, mean, invstd, gamma, beta, relu, y);
    else if (vec) hipLaunchKernelGGL(k_bn_apply4<16>, grid, dim3(256), 0, st, x, seg_ptr, C, mean, invstd, gamma, beta, relu, y);
    else hipLaunchKernelGGL(k_bn_apply, grid, dim3(256), 0, st, x, seg_ptr, C, mean, invstd, gamma, beta, relu, y);
    T2P_CHECK_LAUNCH("bn_apply");
    return 0;
}

int launch_bn_relu_train_backward(const float* dy, const float* x, const float* beta, const int32_t* seg_ptr, int n_seg,
                                  int64_t rows, int C, const float* mean, const float* invstd, const float* gamma, int relu,
                                  float* dx, float* dgamma_seg, float* dbeta_seg, double* part, hipStream_t st) {
    if (n_seg == 0) return 0;
    const int R = bn_chunks(rows, n_seg);
    const dim3 grid((unsigned)n_seg, (unsigned)((C + kCols - 1) / kCols), (unsigned)R);
    const bool vec = bn_vec_ok(C, x, beta, mean, invstd, gamma) && (((uintptr_t)dy | (uintptr_t)dx | (uintptr_t)dgamma_seg | (uintptr_t)dbeta_seg) & 15) == 0;
    const dim3 grid8((unsigned)n_seg, (unsigned)((C + 31) / 32), (unsigned)R);
    if (vec && C <= 32) hipLaunchKernelGGL(k_bn_bwd_partial4<8>, grid8, dim3(256), 0, st, dy, x, gamma, beta, seg_ptr, C, mean, invstd, relu, part);
    else if (vec) hipLaunchKernelGGL(k_bn_bwd_partial4<16>, grid, dim3(256), 0, st, dy, x, gamma, beta, seg_ptr, C, mean, invstd, relu, part);
    else hipLaunchKernelGGL(k_bn_bwd_partial, grid, dim3(256), 0, st, dy, x, gamma, beta, seg_ptr, C, mean, invstd, relu, part);
    T2P_CHECK_LAUNCH("bn_bwd_partial");
    const int64_t sc = (int64_t)n_seg * C;
    hipLaunchKernelGGL(k_bn_bwd_finish, dim3((unsigned)((sc + 255) / 256)), dim3(256), 0, st, part, n_seg, C, R, dgamma_seg,
                       dbeta_seg);
    T2P_CHECK_LAUNCH("bn_bwd_finish");
    hipLaunchKernelGGL(k_bn_bwd_total, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, st, n_seg, C, dgamma_seg, dbeta_seg);
    T2P_CHECK_LAUNCH("bn_bwd_total");
    if (vec && C <= 32) hipLaunchKernelGGL(k_bn_bwd_apply4<8>, grid8, dim3(256), 0, st, dy, x, beta, seg_ptr, C, mean, invstd, gamma, relu, dgamma_seg, dbeta_seg, dx);
    else if (vec) hipLaunchKernelGGL(k_bn_bwd_apply4<16>, grid, dim3(256), 0, st, dy, x, beta, seg_ptr, C, mean, invstd, gamma, relu, dgamma_seg, dbeta_seg, dx);
    else hipLaunchKernelGGL(k_bn_bwd_apply, grid, dim3(256), 0, st, dy, x, beta, seg_ptr, C, mean, invstd, gamma, relu, dgamma_seg,
                            dbeta_seg, dx);
    T2P_CHECK_LAUNCH("bn_bwd_apply");
    return 0;
}

size_t bn_train_workspace_bytes(int64_t rows, int n_seg, int C) {
    return (size_t)n_seg * bn_chunks(rows, n_seg) * 2 * C * sizeof(double);
}

int launch_segment_max(const float* x, const int32_t* seg_ptr, int n_seg, int C, float* out, int32_t* arg, hipStream_t st) {
    if (n_seg == 0) return 0;
    hipLaunchKernelGGL(k_segment_max, dim3((unsigned)n_seg, (unsigned)((C + kCols - 1) / kCols)), dim3(256), 0, st, x, seg_ptr, C,
                       out, arg);
    T2P_CHECK_LAUNCH("segment_max");
    return 0;
}

int launch_segment_max_backward(const float* dout, const int32_t* arg, const int32_t* seg_ptr, int n_seg, int C, float* dx,
                                hipStream_t st) {
    if (n_seg == 0) return 0;
    hipLaunchKernelGGL(k_segment_max_backward, dim3((unsigned)n_seg, (unsigned)((C + kCols - 1) / kCols)), dim3(256), 0, st,
                       dout, arg, seg_ptr, C, dx);
    T2P_CHECK_LAUNCH("segment_max_backward");
    return 0;
}

int launch_edge_counts(const uint16_t* rows, const uint16_t* n_rows, const int32_t* first_obj, int64_t n_obj, int n_dense, int n_cent,
                       int self_loops, int32_t* counts, hipStream_t st) {
    T2P_CHECK_ARG(n_cent >= 1 && n_cent <= 128 && n_dense >= n_cent && n_dense <= 256, "edge_counts: n_dense=%d n_cent=%d", n_dense, n_cent);
    if (n_obj == 0) return 0;
    const int64_t grid = n_obj < 65535 * 16 ? n_obj : 65535 * 16;
    hipLaunchKernelGGL(k_edge_counts, dim3((unsigned)grid), dim3(64), 0, st, rows, n_rows, first_obj, n_obj, n_dense, n_cent, self_loops,
                       counts);
    T2P_CHECK_LAUNCH("edge_counts");
    return 0;
}

int launch_edge_expand(const uint16_t* rows, const uint16_t* n_rows, const int32_t* first_obj, const int32_t* cent_ptr, int64_t n_obj,
                       int n_dense, int n_cent, int self_loops, int32_t* src, int32_t* dst, hipStream_t st) {
    T2P_CHECK_ARG(n_cent >= 1 && n_cent <= 128 && n_dense >= n_cent && n_dense <= 256, "edge_expand: n_dense=%d n_cent=%d", n_dense, n_cent);
    T2P_CHECK_ARG(n_obj * (int64_t)n_dense < (1LL << 31), "edge_expand: row indices beyond int32");
    if (n_obj == 0) return 0;
    const int64_t grid = n_obj < 65535 * 16 ? n_obj : 65535 * 16;
    hipLaunchKernelGGL(k_edge_expand, dim3((unsigned)grid), dim3(64), 0, st, rows, n_rows, first_obj, cent_ptr, n_obj, n_dense, n_cent,
                       self_loops, src, dst);
    T2P_CHECK_LAUNCH("edge_expand");
    return 0;
}

}  // namespace t2p

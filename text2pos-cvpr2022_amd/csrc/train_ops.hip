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

// mean / 1/sqrt(var + eps) (biased variance, as BatchNorm1d normalises) of every column over the rows of a segment
__global__ __launch_bounds__(256) void k_bn_stats(const float* __restrict__ x, const int32_t* __restrict__ seg_ptr, int C,
                                                  float eps, float* __restrict__ mean, float* __restrict__ invstd,
                                                  float* __restrict__ var_unbiased) {
    __shared__ double red[kRowsPar][kCols];
    const int s = blockIdx.x, cl = threadIdx.x % kCols, rl = threadIdx.x / kCols;
    const int c = blockIdx.y * kCols + cl;
    const int r0 = seg_ptr[s], r1 = seg_ptr[s + 1], n = r1 - r0;
    const bool ok = c < C;
    double acc = 0.0;
    for (int r = r0 + rl; r < r1; r += kRowsPar) acc += ok ? (double)x[(int64_t)r * C + c] : 0.0;
    const double m = n > 0 ? combine4(red, rl, cl, acc) / (double)n : 0.0;
    acc = 0.0;
    for (int r = r0 + rl; r < r1; r += kRowsPar) {
        const double d = ok ? (double)x[(int64_t)r * C + c] - m : 0.0;
        acc += d * d;
    }
    const double ss = combine4(red, rl, cl, acc);
    if (ok && rl == 0) {
        const double var = n > 0 ? ss / (double)n : 0.0;
        mean[(int64_t)s * C + c] = (float)m;
        invstd[(int64_t)s * C + c] = (float)(1.0 / sqrt(var + (double)eps));
        var_unbiased[(int64_t)s * C + c] = (float)(n > 1 ? ss / (double)(n - 1) : var);
    }
}

// y = act(gamma (x - mean) invstd + beta) for the rows of segment blockIdx.x
__global__ __launch_bounds__(256) void k_bn_apply(const float* __restrict__ x, const int32_t* __restrict__ seg_ptr, int C,
                                                  const float* __restrict__ mean, const float* __restrict__ invstd,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta, int relu,
                                                  float* __restrict__ y) {
    const int s = blockIdx.x, cl = threadIdx.x % kCols, rl = threadIdx.x / kCols;
    const int c = blockIdx.y * kCols + cl;
    if (c >= C) return;
    const float m = mean[(int64_t)s * C + c], is = invstd[(int64_t)s * C + c], g = gamma[c], b = beta[c];
    for (int r = seg_ptr[s] + rl; r < seg_ptr[s + 1]; r += kRowsPar) {
        const float v = (x[(int64_t)r * C + c] - m) * is * g + b;
        y[(int64_t)r * C + c] = relu ? fmaxf(v, 0.f) : v;
    }
}

// backward of the block above: dz = dy (y > 0);  dx = gamma invstd / n (n dz - sum dz - xhat sum(dz xhat));
// per-segment dgamma = sum dz xhat, dbeta = sum dz (summed over the segments by the caller)
__global__ __launch_bounds__(256) void k_bn_backward(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ y, const int32_t* __restrict__ seg_ptr, int C,
                                                     const float* __restrict__ mean, const float* __restrict__ invstd,
                                                     const float* __restrict__ gamma, int relu, float* __restrict__ dx,
                                                     float* __restrict__ dgamma_seg, float* __restrict__ dbeta_seg) {
    __shared__ double red[kRowsPar][kCols];
    const int s = blockIdx.x, cl = threadIdx.x % kCols, rl = threadIdx.x / kCols;
    const int c = blockIdx.y * kCols + cl;
    const int r0 = seg_ptr[s], r1 = seg_ptr[s + 1], n = r1 - r0;
    const bool ok = c < C;
    const float m = ok ? mean[(int64_t)s * C + c] : 0.f, is = ok ? invstd[(int64_t)s * C + c] : 0.f;
    double a0 = 0.0, a1 = 0.0;
    for (int r = r0 + rl; r < r1; r += kRowsPar) {
        if (!ok) break;
        const int64_t i = (int64_t)r * C + c;
        const float dz = (!relu || y[i] > 0.f) ? dy[i] : 0.f;
        a0 += (double)dz;
        a1 += (double)dz * (double)((x[i] - m) * is);
    }
    const double sdz = combine4(red, rl, cl, a0);
    const double sdx = combine4(red, rl, cl, a1);
    if (!ok) return;
    if (rl == 0) {
        dgamma_seg[(int64_t)s * C + c] = (float)sdx;
        dbeta_seg[(int64_t)s * C + c] = (float)sdz;
    }
    const float g = gamma[c], inv_n = n > 0 ? 1.f / (float)n : 0.f;
    for (int r = r0 + rl; r < r1; r += kRowsPar) {
        const int64_t i = (int64_t)r * C + c;
        const float dz = (!relu || y[i] > 0.f) ? dy[i] : 0.f;
        const float xh = (x[i] - m) * is;
        dx[i] = g * is * (dz - (float)sdz * inv_n - xh * (float)sdx * inv_n);
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

}  // namespace

int launch_bn_relu_train_forward(const float* x, const int32_t* seg_ptr, int n_seg, int C, const float* gamma,
                                 const float* beta, float eps, int relu, float* y, float* mean, float* invstd,
                                 float* var_unbiased, hipStream_t st) {
    if (n_seg == 0) return 0;
    const dim3 grid((unsigned)n_seg, (unsigned)((C + kCols - 1) / kCols));
    hipLaunchKernelGGL(k_bn_stats, grid, dim3(256), 0, st, x, seg_ptr, C, eps, mean, invstd, var_unbiased);
    T2P_CHECK_LAUNCH("bn_stats");
    hipLaunchKernelGGL(k_bn_apply, grid, dim3(256), 0, st, x, seg_ptr, C, mean, invstd, gamma, beta, relu, y);
    T2P_CHECK_LAUNCH("bn_apply");
    return 0;
}

int launch_bn_relu_train_backward(const float* dy, const float* x, const float* y, const int32_t* seg_ptr, int n_seg, int C,
                                  const float* mean, const float* invstd, const float* gamma, int relu, float* dx,
                                  float* dgamma_seg, float* dbeta_seg, hipStream_t st) {
    if (n_seg == 0) return 0;
    const dim3 grid((unsigned)n_seg, (unsigned)((C + kCols - 1) / kCols));
    hipLaunchKernelGGL(k_bn_backward, grid, dim3(256), 0, st, dy, x, y, seg_ptr, C, mean, invstd, gamma, relu, dx, dgamma_seg,
                       dbeta_seg);
    T2P_CHECK_LAUNCH("bn_backward");
    return 0;
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

}  // namespace t2p

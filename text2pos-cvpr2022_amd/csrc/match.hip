// Fine stage: SuperGlue-style hint <-> object matching + offset regression.
//
// Replaces (reference): SuperGlue.forward models/superglue.py:239-330 (AttentionalGNN :132-146, AttentionalPropagation
// :119-129, MultiHeadedAttention :97-116, log_optimal_transport :149-177, mutual-NN + threshold :297-322) and
// mlp_offsets of SuperGlueMatch.forward models/superglue_matcher.py:116.
//
// MI355X design: a sample is tiny (16 objects + 6 hints, D = 128), a batch is not (query x top-k candidates), and the
// same layer processes both token sets with the same weights.  So all tokens of all samples form ONE row matrix
// X [B (M+N)][D] and every Conv1d(k=1) of the GNN is a row GEMM on the fp32 MFMA kernel (tg_gemm.hip): per layer
//   QKV = X Wqkv (q | k | v projected for every token once: "self" / "cross" only select which rows a token attends to)
//   MSG = per-sample multi-head attention (k_attn: one workgroup per sample, its QKV rows in LDS)
//   CAT[:, D:] = MSG Wm;   H = relu(CAT W1) (BatchNorm folded);   X += H W2 (residual in the GEMM epilogue)
// X lives in the left half of CAT [T][2D], so the concat of AttentionalPropagation is free.  The optimal-transport
// part (score matrix, 50 log-Sinkhorn iterations on (M+1) x (N+1), exp, mutual nearest neighbours, threshold) and the
// offset MLP run in one wavefront per sample with the matrix in LDS (k_match_final).  fp32 throughout.
#include "t2p_common.h"

#include "../../include/t2p.h"

namespace t2p {
namespace {

constexpr int kHeads = 4;

__global__ void k_concat(const float* __restrict__ d0, const float* __restrict__ d1, int64_t B, int M, int N, int D,
                         float* __restrict__ cat) {
    const int T = M + N;
    const int64_t total = B * T * (int64_t)D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % D);
        const int64_t row = i / D;
        const int t = (int)(row % T);
        const int64_t b = row / T;
        cat[row * (2 * D) + c] = t < M ? d0[(b * M + t) * D + c] : d1[(b * N + (t - M)) * D + c];
    }
}

// One workgroup per sample.  qkv rows [T][3D] (q | k | v), channel c of a projection = d * heads + h.
// Work item = (token t, head h): scores against the source token set, softmax, weighted sum of the values.
template <int DH>
__global__ __launch_bounds__(256) void k_attn(const float* __restrict__ qkv, int M, int N, int cross,
                                              float* __restrict__ msg) {
    constexpr int D = DH * kHeads;
    extern __shared__ float sm[];
    const int T = M + N;
    const int LD = 3 * D + 4;  // +4: rows of different tokens start in different banks
    float* rows = sm;               // [T][LD]
    float* sc = sm + T * LD;        // [T * heads][max(M, N)]
    const int S = M > N ? M : N;
    const int64_t b = blockIdx.x;
    const float* src = qkv + b * T * (int64_t)(3 * D);
    for (int i = threadIdx.x; i < T * 3 * D; i += blockDim.x) rows[(i / (3 * D)) * LD + (i % (3 * D))] = src[i];
    __syncthreads();
    const float scale = 1.0f / sqrtf((float)DH);
    for (int item = threadIdx.x; item < T * kHeads; item += blockDim.x) {
        const int t = item / kHeads, h = item % kHeads;
        const bool side0 = t < M;                       // token of set 0 (objects)
        const bool from0 = cross ? !side0 : side0;      // source set
        const int s0 = from0 ? 0 : M, ns = from0 ? M : N;
        const float* q = rows + t * LD + h;
        float* s = sc + item * S;
        float mx = -INFINITY;
        for (int m = 0; m < ns; m++) {
            const float* k = rows + (s0 + m) * LD + D + h;
            float a = 0.f;
#pragma unroll
            for (int d = 0; d < DH; d++) a = fmaf(q[d * kHeads], k[d * kHeads], a);
            a *= scale;
            s[m] = a;
            mx = fmaxf(mx, a);
        }
        float den = 0.f;
        for (int m = 0; m < ns; m++) {
            const float e = expf(s[m] - mx);
            s[m] = e;
            den += e;
        }
        float acc[DH];
#pragma unroll
        for (int d = 0; d < DH; d++) acc[d] = 0.f;
        for (int m = 0; m < ns; m++) {
            const float p = s[m] / den;
            const float* v = rows + (s0 + m) * LD + 2 * D + h;
#pragma unroll
            for (int d = 0; d < DH; d++) acc[d] = fmaf(p, v[d * kHeads], acc[d]);
        }
        float* o = msg + (b * T + t) * (int64_t)D + h;
#pragma unroll
        for (int d = 0; d < DH; d++) o[d * kHeads] = acc[d];
    }
}

// One wavefront per sample: score matrix, log-space Sinkhorn with dustbins, exp, mutual NN + threshold, offsets.
__global__ __launch_bounds__(64) void k_match_final(const float* __restrict__ md /*[B T][D] final_proj outputs*/,
                                                    const float* __restrict__ d1 /*[B N][D] hint encodings*/,
                                                    int M, int N, int D, float alpha, int iters, float thresh,
                                                    const float* __restrict__ wo1, const float* __restrict__ bo1,
                                                    const float* __restrict__ wo2, const float* __restrict__ bo2,
                                                    float* __restrict__ P, int64_t* __restrict__ matches0,
                                                    int64_t* __restrict__ matches1, float* __restrict__ ms0,
                                                    float* __restrict__ ms1, float* __restrict__ offsets) {
    extern __shared__ float sm[];
    const int lane = threadIdx.x;
    const int T = M + N, M1 = M + 1, N1 = N + 1;
    float* Z = sm;                  // [M1][N1]
    float* u = Z + M1 * N1;         // [M1]
    float* v = u + M1;              // [N1]
    float* rmax = v + N1;           // [M] row max (inner), [N] col max
    int* ridx = (int*)(rmax + M + N);  // [M] + [N]
    float* hid = (float*)(ridx + M + N);  // [D/2]
    const int64_t b = blockIdx.x;
    const float* m0 = md + b * T * (int64_t)D;
    const float* m1 = m0 + M * (int64_t)D;
    const float inv = 1.0f / sqrtf((float)D);
    for (int e = lane; e < M1 * N1; e += 64) {
        const int i = e / N1, j = e % N1;
        float a = alpha;
        if (i < M && j < N) {
            a = 0.f;
            for (int k = 0; k < D; k++) a = fmaf(m0[i * (int64_t)D + k], m1[j * (int64_t)D + k], a);
            a *= inv;
        }
        Z[e] = a;
    }
    const float norm = -logf((float)(M + N));
    const float lmu_bin = logf((float)N) + norm, lnu_bin = logf((float)M) + norm;
    if (lane < M1) u[lane] = 0.f;
    if (lane < N1) v[lane] = 0.f;
    __syncthreads();
    for (int it = 0; it < iters; it++) {
        if (lane < M1) {  // u = log_mu - logsumexp_j(Z + v)
            float mx = -INFINITY;
            for (int j = 0; j < N1; j++) mx = fmaxf(mx, Z[lane * N1 + j] + v[j]);
            float s = 0.f;
            for (int j = 0; j < N1; j++) s += expf(Z[lane * N1 + j] + v[j] - mx);
            u[lane] = (lane < M ? norm : lmu_bin) - (mx + logf(s));
        }
        __syncthreads();
        if (lane < N1) {  // v = log_nu - logsumexp_i(Z + u)
            float mx = -INFINITY;
            for (int i = 0; i < M1; i++) mx = fmaxf(mx, Z[i * N1 + lane] + u[i]);
            float s = 0.f;
            for (int i = 0; i < M1; i++) s += expf(Z[i * N1 + lane] + u[i] - mx);
            v[lane] = (lane < N ? norm : lnu_bin) - (mx + logf(s));
        }
        __syncthreads();
    }
    for (int e = lane; e < M1 * N1; e += 64) {
        const int i = e / N1, j = e % N1;
        const float z = Z[e] + u[i] + v[j] - norm;
        Z[e] = z;
        P[b * M1 * N1 + e] = expf(z);
    }
    __syncthreads();
    // maxima over the inner (non-dustbin) block; ties -> first index
    if (lane < M) {
        float mx = -INFINITY;
        int bi = 0;
        for (int j = 0; j < N; j++)
            if (Z[lane * N1 + j] > mx) { mx = Z[lane * N1 + j]; bi = j; }
        rmax[lane] = mx;
        ridx[lane] = bi;
    }
    if (lane < N) {
        float mx = -INFINITY;
        int bi = 0;
        for (int i = 0; i < M; i++)
            if (Z[i * N1 + lane] > mx) { mx = Z[i * N1 + lane]; bi = i; }
        rmax[M + lane] = mx;
        ridx[M + lane] = bi;
    }
    __syncthreads();
    if (lane < M) {
        const int j = ridx[lane];
        const bool mutual = ridx[M + j] == lane;
        const float s = mutual ? expf(rmax[lane]) : 0.f;
        ms0[b * M + lane] = s;
        matches0[b * M + lane] = (mutual && s > thresh) ? j : -1;
    }
    if (lane < N) {
        const int i = ridx[M + lane];
        const bool mutual1 = ridx[i] == lane;
        const bool mutual0 = ridx[M + ridx[i]] == i;                 // mutual0[i]
        const float s0 = mutual0 ? expf(rmax[i]) : 0.f;              // mscores0[i]
        const float s = mutual1 ? s0 : 0.f;
        ms1[b * N + lane] = s;
        const bool valid0 = mutual0 && s0 > thresh;
        matches1[b * N + lane] = (mutual1 && valid0) ? i : -1;
    }
    // offsets = Linear(D/2 -> 2)(relu(Linear(D -> D/2)(hint)))
    const int H = D / 2;
    for (int j = 0; j < N; j++) {
        const float* x = d1 + (b * N + j) * (int64_t)D;
        __syncthreads();
        for (int o = lane; o < H; o += 64) {
            float a = bo1[o];
            for (int k = 0; k < D; k++) a = fmaf(x[k], wo1[k * H + o], a);
            hid[o] = fmaxf(a, 0.f);
        }
        __syncthreads();
        if (lane < 2) {
            float a = bo2[lane];
            for (int k = 0; k < H; k++) a = fmaf(hid[k], wo2[k * 2 + lane], a);
            offsets[(b * N + j) * 2 + lane] = a;
        }
    }
}

struct MatchWs {
    float *cat, *qkv, *msg, *hid, *md;
};

size_t carve_match(char* base, int64_t B, int M, int N, int D, MatchWs* ws) {
    const size_t T = (size_t)B * (M + N);
    size_t off = 0;
    auto take = [&](size_t n) {
        off = (off + 255) & ~(size_t)255;
        float* p = base ? (float*)(base + off) : nullptr;
        off += n * sizeof(float);
        return p;
    };
    MatchWs w;
    w.cat = take(T * 2 * D);
    w.qkv = take(T * 3 * D);
    w.msg = take(T * D);
    w.hid = take(T * 2 * D);
    w.md = take(T * D);
    if (ws) *ws = w;
    return ((off + 255) & ~(size_t)255) + 256;
}

}  // namespace
}  // namespace t2p

using namespace t2p;

extern "C" {

size_t t2p_match_workspace_bytes(int64_t batch, int32_t n_obj, int32_t n_hints, int32_t embed_dim) {
    if (batch <= 0) return 256;
    return carve_match(nullptr, batch, n_obj, n_hints, embed_dim, nullptr);
}

int t2p_match(const float* desc0, const float* desc1, int64_t batch, int32_t n_obj, int32_t n_hints, int32_t embed_dim,
              const t2p_match_weights* w, int32_t sinkhorn_iters, float match_threshold, float* P, int64_t* matches0,
              int64_t* matches1, float* mscores0, float* mscores1, float* offsets, void* workspace,
              size_t workspace_bytes, t2p_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    const int M = n_obj, N = n_hints, D = embed_dim;
    T2P_CHECK_ARG(desc0 && desc1 && w && P && matches0 && matches1 && mscores0 && mscores1 && offsets, "match: NULL argument");
    T2P_CHECK_ARG(batch >= 0 && M >= 1 && N >= 1 && M < 64 && N < 64, "match: need 1 <= n_obj, n_hints <= 63 (got %d, %d)", M, N);
    if (D != 64 && D != 128 && D != 256) {
        set_error("match: embed_dim=%d not built (64, 128, 256)", D);
        return T2P_E_UNSUPPORTED;
    }
    T2P_CHECK_ARG(w->n_layers >= 0 && w->n_layers <= 64 && (w->n_layers == 0 || w->cross != nullptr), "match: bad layer list");
    T2P_CHECK_ARG(sinkhorn_iters >= 0, "match: sinkhorn_iters < 0");
    if (batch == 0) return 0;
    const int T = M + N;
    const size_t lds_attn = ((size_t)T * (3 * D + 4) + (size_t)T * kHeads * (M > N ? M : N)) * sizeof(float);
    if (lds_attn > 160 * 1024) {
        set_error("match: %d tokens x D=%d need %zu B of LDS per sample (> 160 KiB)", T, D, lds_attn);
        return T2P_E_UNSUPPORTED;
    }
    MatchWs ws;
    const size_t need = carve_match((char*)workspace, batch, M, N, D, &ws);
    if (workspace == nullptr || need > workspace_bytes) {
        set_error("match: workspace %zu B < %zu B", workspace_bytes, need);
        return T2P_E_WORKSPACE;
    }
    const int64_t rows = batch * T;
    {
        ProfScope ps_("match_concat", st);
        const int64_t total = rows * D;
        const unsigned grid = (unsigned)((total + 255) / 256 < 65535 ? (total + 255) / 256 : 65535);
        hipLaunchKernelGGL(k_concat, dim3(grid), dim3(256), 0, st, desc0, desc1, batch, M, N, D, ws.cat);
        T2P_CHECK_LAUNCH("match_concat");
    }
    auto attn = D == 64 ? k_attn<16> : (D == 128 ? k_attn<32> : k_attn<64>);
    T2P_TRY(reserve_lds((const void*)attn, 160 * 1024, "match"));
    for (int l = 0; l < w->n_layers; l++) {
        const float* wqkv = w->wqkv + (size_t)l * D * 3 * D;
        const float* bqkv = w->bqkv + (size_t)l * 3 * D;
        const float* wm = w->wm + (size_t)l * D * D;
        const float* bm = w->bm + (size_t)l * D;
        const float* w1 = w->w1 + (size_t)l * 2 * D * 2 * D;
        const float* b1 = w->b1 + (size_t)l * 2 * D;
        const float* w2 = w->w2 + (size_t)l * 2 * D * D;
        const float* b2 = w->b2 + (size_t)l * D;
        const bool x3 = w->wqkv_x3 != nullptr;
        // image sizes in halves: [2][N][Kp]
        auto img = [&](const void* base, int k, int n) {
            return (const void*)((const uint16_t*)base + (size_t)l * 2 * n * ((k + 31) / 32 * 32));
        };
        if (x3)
            T2P_TRY(launch_gemm_x3(ws.cat, 2 * D, img(w->wqkv_x3, D, 3 * D), w->scale_qkv, bqkv, ws.qkv, 3 * D, 0, rows, D, 3 * D, 0, st));
        else
            T2P_TRY(launch_gemm(ws.cat, 2 * D, wqkv, bqkv, ws.qkv, 3 * D, 0, rows, D, 3 * D, 0, st));
        {
            ProfScope ps_("match_attn", st);
            hipLaunchKernelGGL(attn, dim3((unsigned)batch), dim3(256), lds_attn, st, ws.qkv, M, N, w->cross[l], ws.msg);
            T2P_CHECK_LAUNCH("match_attn");
        }
        if (x3) {
            T2P_TRY(launch_gemm_x3(ws.msg, D, img(w->wm_x3, D, D), w->scale_m, bm, ws.cat, 2 * D, D, rows, D, D, 0, st));
            T2P_TRY(launch_gemm_x3(ws.cat, 2 * D, img(w->w1_x3, 2 * D, 2 * D), w->scale_1, b1, ws.hid, 2 * D, 0, rows, 2 * D, 2 * D, 1, st));
            T2P_TRY(launch_gemm_x3(ws.hid, 2 * D, img(w->w2_x3, 2 * D, D), w->scale_2, b2, ws.cat, 2 * D, 0, rows, 2 * D, D, 0, st, ws.cat, 2 * D));
        } else {
            T2P_TRY(launch_gemm(ws.msg, D, wm, bm, ws.cat, 2 * D, D, rows, D, D, 0, st));                 // message -> CAT[:, D:]
            T2P_TRY(launch_gemm(ws.cat, 2 * D, w1, b1, ws.hid, 2 * D, 0, rows, 2 * D, 2 * D, 1, st));      // Conv + BN + ReLU
            T2P_TRY(launch_gemm(ws.hid, 2 * D, w2, b2, ws.cat, 2 * D, 0, rows, 2 * D, D, 0, st, ws.cat, 2 * D));  // X += delta
        }
    }
    if (w->wf_x3 != nullptr)
        T2P_TRY(launch_gemm_x3(ws.cat, 2 * D, w->wf_x3, w->scale_f, w->bf, ws.md, D, 0, rows, D, D, 0, st));
    else
        T2P_TRY(launch_gemm(ws.cat, 2 * D, w->wf, w->bf, ws.md, D, 0, rows, D, D, 0, st));
    {
        const size_t lds = ((size_t)(M + 1) * (N + 1) + (M + 1) + (N + 1) + 2 * (M + N) + D / 2 + 8) * sizeof(float);
        ProfScope ps_("match_final", st);
        hipLaunchKernelGGL(k_match_final, dim3((unsigned)batch), dim3(64), lds, st, ws.md, desc1, M, N, D, w->bin_score,
                           sinkhorn_iters, match_threshold, w->wo1, w->bo1, w->wo2, w->bo2, P, matches0, matches1,
                           mscores0, mscores1, offsets);
        T2P_CHECK_LAUNCH("match_final");
    }
    return 0;
}

}  // extern "C"

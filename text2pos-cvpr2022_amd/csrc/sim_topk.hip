// All-pairs cosine scores in float64 + ordered top-k, scores never materialised in HBM.
//
// Replaces the host loop of training/coarse.py:134-140 (per query: `scores = cell_encodings[:] @ text_encodings[q]`
// on float64 arrays, `np.argsort(-scores)[0:max(top_k)]`).  The reference ranks in float64 (np.zeros arrays,
// training/coarse.py:100,103), so the GEMM runs on v_mfma_f64_16x16x4_f64: the fp32 embeddings are widened on the
// fly and every score is an fp64 fma chain -- ranking ties are then only exact duplicates, which are ordered by
// ascending cell index (the pinned stable order, oracle/model.py::retrieve_topk_f64).
//
// Pass 1  (grid = query blocks x cell splits): a workgroup keeps 128 queries as MFMA B operands in registers
//         (fp32, widened at use), streams its cell range through LDS in 32-row blocks (register-prefetched) and
//         every lane maintains a private sorted top-KCAP list per query in registers.
// Pass 2  (one wave per query): k-round ordered selection over the 4 x splits partial lists.
#include "t2p_common.h"

namespace t2p {
namespace {

constexpr int KCAP = 16;     // per-lane list capacity == largest supported k
constexpr int QB = 128;      // queries per workgroup (4 waves x 2 tiles x 16)
constexpr int CB = 32;       // cells staged per iteration

struct Cand {
    double s;
    int i;
};
__device__ __forceinline__ bool better(double s, int i, double s2, int i2) { return s > s2 || (s == s2 && i < i2); }

template <int DIM>
__global__ __launch_bounds__(256, 1) void k_sim_partial(const float* __restrict__ Q, const float* __restrict__ Cm,
                                                         int64_t nq, int64_t nc, int cells_per_split,
                                                         double* __restrict__ ps, int* __restrict__ pi, int n_split) {
    constexpr int LDC = DIM + 4;
    constexpr int KQ = DIM / 4;   // k per lane group (lane>>4 owns k in [g*KQ, (g+1)*KQ))
    constexpr int F4 = CB * DIM / 4 / 256;
    __shared__ __attribute__((aligned(16))) float c_lds[CB * LDC];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g4 = lane >> 4, l15 = lane & 15;
    const int64_t q0 = (int64_t)blockIdx.x * QB + wave * 32;
    const int split = blockIdx.y;
    const int64_t c_begin = (int64_t)split * cells_per_split;
    const int64_t c_end = (c_begin + cells_per_split) < nc ? (c_begin + cells_per_split) : nc;

    // B operands: this wave's 2 query tiles, fp32 in registers
    float qreg[2][KQ];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int64_t q = q0 + t * 16 + l15;
#pragma unroll
        for (int s4 = 0; s4 < KQ / 4; s4++) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (q < nq) v = *(const f32x4*)(Q + q * DIM + g4 * KQ + s4 * 4);
#pragma unroll
            for (int e = 0; e < 4; e++) qreg[t][s4 * 4 + e] = v[e];
        }
    }
    double ls[2][KCAP];
    int li[2][KCAP];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int j = 0; j < KCAP; j++) { ls[t][j] = -INFINITY; li[t][j] = 0x7fffffff; }

    f32x4 stage[F4];
    auto load_block = [&](int64_t cb) {
#pragma unroll
        for (int it = 0; it < F4; it++) {
            const int qd = it * 256 + tid;
            const int r = qd / (DIM / 4), c4 = qd % (DIM / 4);
            const int64_t cell = cb + r;
            stage[it] = cell < c_end ? *(const f32x4*)(Cm + cell * DIM + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_block = [&]() {
#pragma unroll
        for (int it = 0; it < F4; it++) {
            const int qd = it * 256 + tid;
            const int r = qd / (DIM / 4), c4 = qd % (DIM / 4);
            *(f32x4*)(c_lds + r * LDC + c4 * 4) = stage[it];
        }
    };

    if (c_begin < c_end) load_block(c_begin);
    for (int64_t cb = c_begin; cb < c_end; cb += CB) {
        store_block();
        __syncthreads();
        if (cb + CB < c_end) load_block(cb + CB);
#pragma unroll
        for (int ct = 0; ct < CB / 16; ct++) {
            f64x4 acc[2];
#pragma unroll
            for (int t = 0; t < 2; t++) acc[t] = f64x4{0.0, 0.0, 0.0, 0.0};
            const float* arow = c_lds + (ct * 16 + l15) * LDC + g4 * KQ;
#pragma unroll
            for (int s4 = 0; s4 < KQ / 4; s4++) {
                const f32x4 a = *(const f32x4*)(arow + s4 * 4);
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const double ad = (double)a[e];
#pragma unroll
                    for (int t = 0; t < 2; t++)
                        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ad, (double)qreg[t][s4 * 4 + e], acc[t], 0, 0, 0);
                }
            }
            // D layout (f64 16x16x4): column (query) = lane&15, row (cell) = (lane>>4) + 4*reg
#pragma unroll
            for (int t = 0; t < 2; t++) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int64_t cell = cb + ct * 16 + g4 + 4 * r;
                    const double s = acc[t][r];
                    const int ci = (int)cell;
                    if (cell < c_end && better(s, ci, ls[t][KCAP - 1], li[t][KCAP - 1])) {
                        ls[t][KCAP - 1] = s;
                        li[t][KCAP - 1] = ci;
#pragma unroll
                        for (int j = KCAP - 1; j > 0; j--) {
                            const bool sw = better(ls[t][j], li[t][j], ls[t][j - 1], li[t][j - 1]);
                            const double ts = ls[t][j];
                            const int ti = li[t][j];
                            ls[t][j] = sw ? ls[t][j - 1] : ts;
                            li[t][j] = sw ? li[t][j - 1] : ti;
                            ls[t][j - 1] = sw ? ts : ls[t][j - 1];
                            li[t][j - 1] = sw ? ti : li[t][j - 1];
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    // partial lists: [q][split][g4][KCAP]
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int64_t q = q0 + t * 16 + l15;
        if (q < nq) {
            const int64_t base = ((q * n_split + split) * 4 + g4) * KCAP;
#pragma unroll
            for (int j = 0; j < KCAP; j++) { ps[base + j] = ls[t][j]; pi[base + j] = li[t][j]; }
        }
    }
}

// one wave per query: k ordered selection rounds over n_cand candidates
__global__ __launch_bounds__(256) void k_topk_merge(const double* __restrict__ ps, const int* __restrict__ pi,
                                                    int64_t nq, int n_cand, int k, int64_t index_offset,
                                                    int64_t* __restrict__ out_idx, double* __restrict__ out_score) {
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    const double* s = ps + q * n_cand;
    const int* id = pi + q * n_cand;
    double last_s = INFINITY;
    int last_i = -1;
    for (int r = 0; r < k; r++) {
        double bs = -INFINITY;
        int bi = 0x7fffffff;
        for (int c = lane; c < n_cand; c += 64) {
            const double cs = s[c];
            const int ci = id[c];
            if (ci == 0x7fffffff) continue;
            const bool after = better(last_s, last_i, cs, ci);  // strictly after the previous pick
            if (after && better(cs, ci, bs, bi)) { bs = cs; bi = ci; }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const double os = __shfl_xor(bs, off, 64);
            const int oi = __shfl_xor(bi, off, 64);
            if (better(os, oi, bs, bi)) { bs = os; bi = oi; }
        }
        if (lane == 0) {
            const bool ok = bi != 0x7fffffff;
            out_idx[q * k + r] = ok ? (int64_t)bi + index_offset : -1;
            out_score[q * k + r] = ok ? bs : -INFINITY;
        }
        if (bi == 0x7fffffff) { last_s = -INFINITY; last_i = 0x7fffffff; }
        else { last_s = bs; last_i = bi; }
    }
}

int pick_splits(int64_t nq, int64_t nc) {
    const int64_t qblocks = (nq + QB - 1) / QB;
    int64_t want = (2LL * num_cus() + qblocks - 1) / qblocks;  // ~2 workgroups per CU in flight
    int64_t max_split = (nc + 4 * CB - 1) / (4 * CB);          // keep >= 128 cells per split
    if (want > max_split) want = max_split;
    if (want < 1) want = 1;
    if (want > 1024) want = 1024;
    return (int)want;
}

}  // namespace

size_t sim_topk_workspace_bytes(int64_t nq, int64_t nc, int k) {
    (void)k;
    const int sp = pick_splits(nq, nc);
    return (size_t)nq * sp * 4 * KCAP * (sizeof(double) + sizeof(int)) + 256;
}

int launch_sim_topk(const float* Q, const float* Cm, int64_t nq, int64_t nc, int dim, int k, int64_t c_index_offset,
                    int64_t* out_idx, double* out_score, void* ws, size_t ws_bytes, hipStream_t st) {
    T2P_CHECK_ARG(k >= 1 && k <= KCAP, "sim_topk: k=%d outside [1,%d]", k, KCAP);
    T2P_CHECK_ARG(dim == 256 || dim == 128 || dim == 384, "sim_topk: dim=%d not instantiated (128, 256, 384)", dim);
    T2P_CHECK_ARG(nc < 0x7fffffff, "sim_topk: nc too large");
    T2P_CHECK_ARG((((uintptr_t)Q) & 15) == 0 && (((uintptr_t)Cm) & 15) == 0, "sim_topk: Q and C must be 16-byte aligned");
    if (nq == 0) return 0;
    const int sp = pick_splits(nq, nc);
    const size_t need = sim_topk_workspace_bytes(nq, nc, k);
    if (ws_bytes < need || ws == nullptr) {
        set_error("sim_topk: workspace %zu B < required %zu B", ws_bytes, need);
        return T2P_E_WORKSPACE;
    }
    const int64_t n_lists = nq * sp * 4 * KCAP;
    double* ps = (double*)ws;
    int* pi = (int*)(ps + n_lists);
    int cps = (int)((nc + sp - 1) / sp);
    cps = ((cps + CB - 1) / CB) * CB;
    dim3 grid((unsigned)((nq + QB - 1) / QB), (unsigned)sp);
    {
    ProfScope ps_("sim_partial", st);
    if (dim == 256)
        hipLaunchKernelGGL(k_sim_partial<256>, grid, dim3(256), 0, st, Q, Cm, nq, nc, cps, ps, pi, sp);
    else if (dim == 384)
        hipLaunchKernelGGL(k_sim_partial<384>, grid, dim3(256), 0, st, Q, Cm, nq, nc, cps, ps, pi, sp);
    else
        hipLaunchKernelGGL(k_sim_partial<128>, grid, dim3(256), 0, st, Q, Cm, nq, nc, cps, ps, pi, sp);
    }
    T2P_CHECK_LAUNCH("sim_partial");
    ProfScope ps2_("topk_merge", st);
    hipLaunchKernelGGL(k_topk_merge, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, st, ps, pi, nq, sp * 4 * KCAP, k,
                       c_index_offset, out_idx, out_score);
    T2P_CHECK_LAUNCH("topk_merge");
    return 0;
}

}  // namespace t2p

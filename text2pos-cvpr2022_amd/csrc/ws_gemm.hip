// Weight-stationary streaming fp32-MFMA kernels: the dense hot loop of the PointNet++ / DGCNN cell branch.
//
// Replaces (reference call sites):
//   gnn.PointConv(local_nn)(x, (pos, pos[idx]), edge_index)   models/pointcloud/pointnet2.py:31-35  (sa1/sa2/sa3)
//   GlobalAbstractionLayer mlp + global_max_pool               models/pointcloud/pointnet2.py:45-49
//   gnn.DynamicEdgeConv(mlp, k=8, aggr="max")                   models/cell_retrieval.py:46-48,97
//
// Design (MI355X-first, not a translation of PyG's gather -> addmm -> scatter chain):
//   * Every per-edge MLP  relu(BN(W2 relu(BN(W1 [x_j | pos_j - pos_i]))))  is split algebraically.  Layer 1 is
//     linear before its ReLU, so  W1 [x_j | pos_j - pos_i] + b = A_j - B_i  with one table row per dense POINT
//     (A_j = W1 [x_j | pos_j] + b) and per CENTROID (B_i = W1p pos_i); the per-EDGE work is only
//     h_e = relu(A_j - B_i) (VALU, while staging) and the layer-2 GEMM.  DynamicEdgeConv is the same with
//     h_e = relu(P_i + Q_j).
//   * The layer-2 weight slice of a workgroup lives in VGPRs for the whole launch (up to 256 registers per lane:
//     SA3 keeps the full 256x256 W2 across the 4 waves of a CU), so the persistent loop streams only edge rows:
//     no weight traffic, one barrier per row tile, MFMA operands B from registers and A from LDS (ds_read_b128).
//   * v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD).  Lane half h = lane>>5 owns the contiguous K range
//     [h*K/2, (h+1)*K/2), which turns the A-operand fetch into 16-byte LDS reads; the k summation order is
//     therefore (0, K/2, 1, K/2+1, ...) -- fixed, deterministic.
//   * Staging is software-pipelined over a double-buffered LDS tile: the global gathers of batch t+1 are issued
//     before the MFMA block of batch t and land in registers behind it; the VALU part (+/- destination term, ReLU)
//     and the ds_write follow the MFMA block; one barrier per batch.
//   * Max-aggregation: all layer outputs are post-ReLU (>= 0), so the segmented max is an LDS integer atomic max
//     on the float bit pattern, initialised to +0.  Rows are sorted by destination, so each lane first folds the
//     runs of equal destination among its 16 accumulator rows in registers (typically 16 -> 2-3 atomics).
#include "t2p_common.h"

namespace t2p {
namespace {

constexpr int kMaxRows = 128 * 33 + 256;  // largest SA group (128 centroids x (32 + self loop)) + tile padding
constexpr int kAccFloats = 8192;          // 128x64 = 64x128 = 32x256

template <int K, int NW, int WN, int RT, int MODE>
struct WsCfg {
    static constexpr int WM = 4 / WN;
    static constexpr int NTW = NW / (32 * WN);
    static constexpr int KS = K / 2;
    static constexpr int TR = WM * RT * 32;  // rows staged per barrier interval
    static constexpr int LDH = K + 4;        // padded hidden-row stride (floats)
    static constexpr bool EDGE = (MODE == WS_EDGE_SA || MODE == WS_EDGE_KNN);
    static constexpr int HID_FLOATS = TR * LDH;  // one of the two staging buffers
    static constexpr int ACC_FLOATS = EDGE ? kAccFloats : 0;
    static constexpr int F4_PER_ROW = K / 4;
    static constexpr int TOTAL_F4 = TR * F4_PER_ROW;
    static constexpr int ITERS = (TOTAL_F4 + 255) / 256;
    static constexpr size_t lds_bytes() {
        size_t b = (size_t)(2 * HID_FLOATS + ACC_FLOATS) * 4;
        if (EDGE) b += (size_t)kMaxRows * 4 + kMaxRows + 132 * 4;
        return b;
    }
    static_assert(K % 8 == 0, "K must be a multiple of 8");
    static_assert(NW % (32 * WN) == 0, "NW must split into 32-column tiles per wave");
};

template <int K, int NW, int WN, int RT, int MODE>
__global__ __launch_bounds__(256, 1) void k_ws(WsParams p, int n_slices) {
    using C = WsCfg<K, NW, WN, RT, MODE>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* hid = lds;  // two buffers of HID_FLOATS
    int* acc_lds = (int*)(lds + 2 * C::HID_FLOATS);
    int* rows_src = (int*)(lds + 2 * C::HID_FLOATS + C::ACC_FLOATS);
    uint8_t* rows_dst = (uint8_t*)(rows_src + kMaxRows);
    int* scan = (int*)(rows_dst + kMaxRows);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wn = wave % WN;
    const int wm = wave / WN;
    const int h = lane >> 5;
    const int l31 = lane & 31;

    // XCD-aware stream/slice mapping: the n_slices column slices of one row stream sit on the same XCD
    // (block b runs on XCD b % 8), so they share the stream's A rows through that XCD's L2.
    const int lin = blockIdx.x;
    const int nblk = gridDim.x;
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
    const int ncol0 = slice * NW + wn * C::NTW * 32;  // first output column of this wave

    // ---- stationary weights -> registers ----------------------------------------------------------------
    float w[C::NTW][C::KS];
#pragma unroll
    for (int nt = 0; nt < C::NTW; nt++)
#pragma unroll
        for (int s = 0; s < C::KS; s++)
            w[nt][s] = p.W[(int64_t)(h * C::KS + s) * p.ldw + ncol0 + nt * 32 + l31];
    float bias[C::NTW];
#pragma unroll
    for (int nt = 0; nt < C::NTW; nt++) bias[nt] = p.bias ? p.bias[ncol0 + nt * 32 + l31] : 0.f;

    f32x4 sa[C::ITERS];                  // staged source rows (in flight behind the MFMA block)
    f32x4 sb[C::EDGE ? C::ITERS : 1];    // staged destination terms (edge modes)

    // Issue the global loads of one batch: rows [r0, r0+TR) of group g (n_rows valid rows in the group).
    auto stage_load = [&](int64_t g, int r0, int n_rows) {
#pragma unroll
        for (int it = 0; it < C::ITERS; it++) {
            const int q = it * 256 + tid;
            const int lr = q / C::F4_PER_ROW, c4 = q % C::F4_PER_ROW;
            const int r = r0 + lr;
            sa[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (C::EDGE) sb[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (((C::TOTAL_F4 % 256) == 0 || q < C::TOTAL_F4) && r < n_rows) {
                if constexpr (C::EDGE) {
                    const int src = rows_src[r];
                    const int dl = rows_dst[r];
                    const int64_t dst = (MODE == WS_EDGE_SA) ? (g * p.n_cent + dl) : (g * 32 + dl);
                    sa[it] = *(const f32x4*)(p.A + (int64_t)src * p.lda + c4 * 4);
                    sb[it] = *(const f32x4*)(p.Bc + dst * K + c4 * 4);
                } else if constexpr (MODE == WS_DENSE_GROUPMAX) {
                    sa[it] = *(const f32x4*)(p.A + (g * 32 + r) * (int64_t)p.lda + c4 * 4);
                } else {
                    sa[it] = *(const f32x4*)(p.A + (g * C::TR + r) * (int64_t)p.lda + c4 * 4);
                }
            }
        }
    };
    // VALU part + LDS write of the staged batch.
    auto stage_write = [&](int buf) {
        float* dst = hid + buf * C::HID_FLOATS;
#pragma unroll
        for (int it = 0; it < C::ITERS; it++) {
            const int q = it * 256 + tid;
            if ((C::TOTAL_F4 % 256) != 0 && q >= C::TOTAL_F4) break;
            const int lr = q / C::F4_PER_ROW, c4 = q % C::F4_PER_ROW;
            f32x4 v = sa[it];
            if constexpr (C::EDGE) {
                const f32x4 t = (MODE == WS_EDGE_SA) ? (sa[it] - sb[it]) : (sa[it] + sb[it]);
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = fmaxf(t[e], 0.f);
            }
            *(f32x4*)(dst + lr * C::LDH + c4 * 4) = v;
        }
    };
    // [RT x 32 rows] x [K] x [NTW x 32 cols] per wave.
    auto mfma_block = [&](int buf, f32x16 (&acc)[RT][C::NTW]) {
#pragma unroll
        for (int rt = 0; rt < RT; rt++)
#pragma unroll
            for (int nt = 0; nt < C::NTW; nt++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[rt][nt][e] = bias[nt];  // bias rides in the accumulator
        const float* hrow = hid + buf * C::HID_FLOATS + ((wm * RT) * 32 + l31) * C::LDH + h * C::KS;
        // A operands are fetched one chunk (QC k-quads) ahead of the MFMAs that consume them
        constexpr int NQ = C::KS / 4;
        constexpr int QC = (NQ % 4 == 0) ? 4 : ((NQ % 3 == 0) ? 3 : 1);
        constexpr int NCH = NQ / QC;
        f32x4 a_cur[RT][QC], a_nxt[RT][QC];
#pragma unroll
        for (int rt = 0; rt < RT; rt++)
#pragma unroll
            for (int qi = 0; qi < QC; qi++) a_cur[rt][qi] = *(const f32x4*)(hrow + rt * 32 * C::LDH + qi * 4);
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) {
            if (ch + 1 < NCH) {
#pragma unroll
                for (int rt = 0; rt < RT; rt++)
#pragma unroll
                    for (int qi = 0; qi < QC; qi++)
                        a_nxt[rt][qi] = *(const f32x4*)(hrow + rt * 32 * C::LDH + ((ch + 1) * QC + qi) * 4);
            }
#pragma unroll
            for (int qi = 0; qi < QC; qi++)
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int rt = 0; rt < RT; rt++)
#pragma unroll
                        for (int nt = 0; nt < C::NTW; nt++)
                            acc[rt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                a_cur[rt][qi][j], w[nt][(ch * QC + qi) * 4 + j], acc[rt][nt], 0, 0, 0);
#pragma unroll
            for (int rt = 0; rt < RT; rt++)
#pragma unroll
                for (int qi = 0; qi < QC; qi++) a_cur[rt][qi] = a_nxt[rt][qi];
        }
    };
    // Segmented max of one batch into the group's LDS accumulator (edge modes).
    auto epilogue_edge = [&](int r0, int n_rows, f32x16 (&acc)[RT][C::NTW]) {
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            const int trow0 = r0 + (wm * RT + rt) * 32;
            if (trow0 >= n_rows) continue;
            // destinations of this lane's 16 rows: quads of 4 consecutive rows at trow0 + 8*q + 4*h
            int dq[4][4];
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int r = trow0 + 8 * q + 4 * h + e;
                    dq[q][e] = r < n_rows ? (int)rows_dst[r] : -1;
                }
            bool same[16], is_end[16];
            int dd[16];
#pragma unroll
            for (int e = 0; e < 16; e++) dd[e] = dq[e >> 2][e & 3];
#pragma unroll
            for (int e = 0; e < 16; e++) {
                same[e] = e > 0 && dd[e] == dd[e - 1];
                is_end[e] = dd[e] >= 0 && (e == 15 || dd[e] != dd[e + 1]);
            }
#pragma unroll
            for (int nt = 0; nt < C::NTW; nt++) {
                const int lcol = wn * C::NTW * 32 + nt * 32 + l31;
                float v[16];
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    v[e] = acc[rt][nt][e];  // bias already inside; ReLU is implied by the signed-int max against +0
                    if (e > 0) v[e] = same[e] ? fmaxf(v[e], v[e - 1]) : v[e];
                }
#pragma unroll
                for (int e = 0; e < 16; e++)
                    if (is_end[e]) atomicMax(&acc_lds[dd[e] * NW + lcol], __float_as_int(v[e]));
            }
        }
    };

    if constexpr (!C::EDGE) {
        // ---- dense streams: every group is one batch; the pipeline runs across groups -----------------------------
        int64_t g = stream;
        auto rows_of = [&](int64_t gg) -> int {
            if constexpr (MODE == WS_DENSE_GROUPMAX) return 32;
            const int64_t left = p.M - gg * C::TR;
            return (int)(left < C::TR ? left : C::TR);
        };
        if (g < p.n_groups) {
            stage_load(g, 0, rows_of(g));
            stage_write(0);
        }
        __syncthreads();
        for (int i = 0; g < p.n_groups; g += n_streams, i++) {
            const int64_t gn = g + n_streams;
            const int n_rows = rows_of(g);
            if (gn < p.n_groups) stage_load(gn, 0, rows_of(gn));
            f32x16 acc[RT][C::NTW];
            mfma_block(i & 1, acc);
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
#pragma unroll
                for (int nt = 0; nt < C::NTW; nt++) {
                    const int lcol = wn * C::NTW * 32 + nt * 32 + l31;
                    if constexpr (MODE == WS_DENSE_STORE) {
                        const int trow0 = (wm * RT + rt) * 32;
#pragma unroll
                        for (int e = 0; e < 16; e++) {
                            const int r = trow0 + (e & 3) + 8 * (e >> 2) + 4 * h;
                            float v = acc[rt][nt][e];
                            if (p.relu) v = fmaxf(v, 0.f);
                            if (r < n_rows) p.out[(g * C::TR + r) * (int64_t)p.ldo + slice * NW + lcol] = v;
                        }
                    } else {  // max over the 32 rows of the object: in registers, then across the two lane halves
                        float m = 0.f;
#pragma unroll
                        for (int e = 0; e < 16; e++) m = fmaxf(m, acc[rt][nt][e]);
                        m = fmaxf(m, __shfl_xor(m, 32, 64));
                        if (h == 0) p.out[g * (int64_t)p.ldo + slice * NW + lcol] = m;
                    }
                }
            }
            if (gn < p.n_groups) stage_write((i + 1) & 1);
            __syncthreads();
        }
    } else {
        // ---- edge streams: a group = the destination rows of one object (SA) / 32 objects (kNN) -------------------
        for (int64_t g = stream; g < p.n_groups; g += n_streams) {
            int n_rows;
            if constexpr (MODE == WS_EDGE_SA) {
                const int nc = p.n_cent;
                const int extra = p.self_loops ? 1 : 0;
                int my = 0;
                if (tid < nc) my = (int)p.cnt[g * nc + tid] + extra;
                if (tid <= nc) scan[tid] = 0;
                __syncthreads();
                if (tid < nc) scan[tid + 1] = my;
                __syncthreads();
                if (wave == 0) {  // inclusive scan of <=128 counts by one wave, two entries per lane
                    int a0 = (2 * lane + 1 <= nc) ? scan[2 * lane + 1] : 0;
                    int a1 = (2 * lane + 2 <= nc) ? scan[2 * lane + 2] : 0;
                    int s = a0 + a1;
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) {
                        int t = __shfl_up(s, off, 64);
                        if (lane >= off) s += t;
                    }
                    int excl = s - (a0 + a1);
                    if (2 * lane + 1 <= nc) scan[2 * lane + 1] = excl + a0;
                    if (2 * lane + 2 <= nc) scan[2 * lane + 2] = excl + a0 + a1;
                }
                __syncthreads();
                n_rows = scan[nc];
                int64_t self_base = 0;
                if (p.self_loops) {
                    const int64_t first = p.obj_cell_first[g];
                    self_base = first * p.n_dense + (g - first) * (int64_t)nc;
                }
                if (tid < nc) {
                    int off = scan[tid];
                    const uint8_t* nb = p.nbr + (g * nc + tid) * 32;
                    const int c = my - extra;
                    for (int e = 0; e < c; e++) {
                        rows_src[off + e] = (int)(g * p.n_dense + nb[e]);
                        rows_dst[off + e] = (uint8_t)tid;
                    }
                    if (extra) {
                        rows_src[off + c] = (int)(self_base + tid);
                        rows_dst[off + c] = (uint8_t)tid;
                    }
                }
            } else {
                const int64_t d0 = g * 32;
                const int nd = (int)((p.n_dst - d0) < 32 ? (p.n_dst - d0) : 32);
                int my = 0;
                if (tid < nd)
                    for (int e = 0; e < p.knn_k; e++) my += p.knn_idx[(d0 + tid) * p.knn_k + e] >= 0 ? 1 : 0;
                if (tid <= 32) scan[tid] = 0;
                __syncthreads();
                if (tid < 32) scan[tid + 1] = my;
                __syncthreads();
                if (tid == 0) {
                    int s = 0;
                    for (int i = 1; i <= 32; i++) { s += scan[i]; scan[i] = s; }
                }
                __syncthreads();
                n_rows = scan[32];
                if (tid < nd) {
                    int off = scan[tid];
                    for (int e = 0; e < p.knn_k; e++) {
                        int j = p.knn_idx[(d0 + tid) * p.knn_k + e];
                        if (j >= 0) { rows_src[off] = j; rows_dst[off] = (uint8_t)tid; off++; }
                    }
                }
            }
            for (int i = tid; i < kAccFloats; i += 256) acc_lds[i] = 0;
            __syncthreads();

            const int n_batches = (n_rows + C::TR - 1) / C::TR;
            if (n_batches > 0) {
                stage_load(g, 0, n_rows);
                stage_write(0);
            }
            __syncthreads();
            for (int bt = 0; bt < n_batches; bt++) {
                const bool more = bt + 1 < n_batches;
                if (more) stage_load(g, (bt + 1) * C::TR, n_rows);
                f32x16 acc[RT][C::NTW];
                mfma_block(bt & 1, acc);
                epilogue_edge(bt * C::TR, n_rows, acc);
                if (more) stage_write((bt + 1) & 1);
                __syncthreads();
            }

            // ---- write the group's result -----------------------------------------------------------------------
            if constexpr (MODE == WS_EDGE_SA) {
                const int nc = p.n_cent;
                for (int i = tid; i < nc * NW; i += 256) {
                    const int c = i / NW, col = i % NW;
                    p.out[(g * nc + c) * (int64_t)p.ldo + col] = __int_as_float(acc_lds[i]);
                }
                // append [pos_centroid, 0 x 5] so that the next layer's A rows are [features | pos | pad]
                for (int i = tid; i < nc * 8; i += 256) {
                    const int c = i >> 3, d = i & 7;
                    float v = 0.f;
                    if (d < 3) {
                        const int loc = p.fps_idx[g * nc + c];
                        v = p.pos_src[(g * p.n_dense + loc) * (int64_t)p.ld_pos + p.pos_col0 + d];
                    }
                    p.out[(g * nc + c) * (int64_t)p.ldo + NW + d] = v;
                }
            } else {
                const int64_t d0 = g * 32;
                const int nd = (int)((p.n_dst - d0) < 32 ? (p.n_dst - d0) : 32);
                for (int i = tid; i < nd * NW; i += 256) {
                    const int c = i / NW, col = i % NW;
                    p.out[(d0 + c) * (int64_t)p.ldo + col] = __int_as_float(acc_lds[i]);
                }
            }
            __syncthreads();
        }
    }
}

template <int K, int NW, int WN, int RT, int MODE>
int launch_cfg(const WsParams& p_in, int n_slices, hipStream_t st) {
    using C = WsCfg<K, NW, WN, RT, MODE>;
    auto kern = k_ws<K, NW, WN, RT, MODE>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)C::lds_bytes());
        if (e != hipSuccess) {
            set_error("ws_gemm: cannot reserve %zu B of LDS: %s", C::lds_bytes(), hipGetErrorString(e));
            return (int)e;
        }
        attr_set = true;
    }
    WsParams p = p_in;
    if (MODE == WS_DENSE_STORE) p.n_groups = (p.M + C::TR - 1) / C::TR;
    if (p.n_groups <= 0) return 0;
    const int cus = num_cus();
    int64_t streams = cus / n_slices;
    if (streams < 1) streams = 1;
    if (streams > p.n_groups) streams = p.n_groups;
    if (streams >= 8) streams -= streams % 8;  // keep the XCD-aware mapping valid
    const unsigned grid = (unsigned)(streams * n_slices);
    static const char* kMode[] = {"dense", "groupmax", "edge_sa", "edge_knn"};
    static char name[64];
    if (name[0] == 0) snprintf(name, sizeof(name), "ws_%s_k%d_n%d", kMode[MODE], K, NW * n_slices);
    ProfScope ps_(name, st);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), C::lds_bytes(), st, p, n_slices);
    T2P_CHECK_LAUNCH("ws_gemm");
    return 0;
}

}  // namespace

int launch_ws(int mode, int K, int N, const WsParams& p, hipStream_t st) {
    T2P_CHECK_ARG(((uintptr_t)p.A & 15) == 0 && (p.lda % 4) == 0, "ws_gemm: A must be 16-byte aligned, lda %% 4 == 0");
#define WS_CASE(MODE_, K_, N_, NW_, WN_, RT_)                                     \
    if (mode == MODE_ && K == K_ && N == N_)                                       \
        return launch_cfg<K_, NW_, WN_, RT_, MODE_>(p, N_ / NW_, st);
    // SA layer-2 edge GEMMs (H -> Cout)
    WS_CASE(WS_EDGE_SA, 32, 64, 64, 2, 4)
    WS_CASE(WS_EDGE_SA, 128, 128, 128, 4, 2)
    WS_CASE(WS_EDGE_SA, 256, 256, 256, 4, 1)
    // DynamicEdgeConv layer 2
    WS_CASE(WS_EDGE_KNN, 256, 256, 256, 4, 1)
    // SA2 / SA3 layer-1 point tables  ([feat | pos | pad] -> H), GA layer 1
    WS_CASE(WS_DENSE_STORE, 72, 128, 128, 4, 2)
    WS_CASE(WS_DENSE_STORE, 136, 256, 256, 4, 1)
    WS_CASE(WS_DENSE_STORE, 264, 512, 256, 4, 1)
    // GA layer 2 + max over the 32 points of an object
    WS_CASE(WS_DENSE_GROUPMAX, 512, 1024, 128, 4, 1)
#undef WS_CASE
    set_error("ws_gemm: no instantiation for mode=%d K=%d N=%d", mode, K, N);
    return T2P_E_UNSUPPORTED;
}

}  // namespace t2p

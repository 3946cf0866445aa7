// Weight-stationary streaming MFMA kernels for the dense layers of the PointNet++ / DGCNN cell branch and for the
// DynamicEdgeConv edge layer.
//
// Replaces (reference call sites):
//   layer 1 of every SA local_nn, applied per dense point (see below)            models/pointcloud/pointnet2.py:31-35
//   GlobalAbstractionLayer mlp + global_max_pool                                  models/pointcloud/pointnet2.py:45-49
//   gnn.DynamicEdgeConv(mlp, k=8, aggr="max")                                     models/cell_retrieval.py:46-48,97
//
// Design (MI355X-first, not a translation of PyG's gather -> addmm -> scatter chain):
//   * Every per-edge MLP  relu(BN(W2 relu(BN(W1 [x_j | pos_j - pos_i]))))  is split algebraically.  Layer 1 is
//     linear before its ReLU, so  W1 [x_j | pos_j - pos_i] + b = A_j - B_i  with one table row per dense POINT
//     (A_j = W1 [x_j | pos_j] + b: the DENSE_STORE mode below) and per CENTROID (B_i = W1p pos_i); the per-EDGE work
//     is only h_e = relu(A_j - B_i) and the layer-2 GEMM (csrc/ws_sa.hip).  DynamicEdgeConv is the same with
//     h_e = relu(P_i + Q_j) (EDGE_KNN mode below).
//   * The weight slice of a workgroup lives in VGPRs for the whole launch, so the persistent loop streams only
//     activation rows: no weight traffic, one barrier per row batch, MFMA B operands from registers and A operands
//     from LDS (ds_read_b128).  Lane half h = lane>>5 owns the contiguous K range [h*K/2, (h+1)*K/2), which makes the
//     operand fetches 16-byte; the k summation order is therefore fixed and deterministic.
//   * Two arithmetic paths: X3 = 0: v_mfma_f32_32x32x2_f32 (exact fp32 fma chains); X3 = 1: the "f16x3" split
//     (x = hi + lo/2048, hi/lo fp16, three v_mfma_f32_32x32x16_f16 per 16 k, fp32 accumulation; see csrc/ws_sa.hip).
//   * Staging is software-pipelined over a double-buffered LDS tile: the global loads of batch t+1 are issued before
//     the MFMA block of batch t; their VALU part and the LDS write follow it; one barrier per batch.
//   * Column slices of one row stream (GA: N = 1024 in slices of 128) are mapped to the same XCD (block b runs on
//     XCD b % 8), so they share the stream's A rows through that XCD's L2.
#ifndef T2P_GABL   // development only (results wrong): 1 = GA layer 1 stores into a 4 MB window (no HBM write traffic)
#define T2P_GABL 0
#endif
#ifndef T2P_GA2_V1
#define T2P_GA2_V1 0
#endif
#include <type_traits>

#include "t2p_common.h"

namespace t2p {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));

constexpr int kMaxRows = 32 * 33;   // kNN group: 32 destination rows x up to 32 neighbours (+ slack)
constexpr int kAccFloats = 8192;    // 32 x 256

template <int K, int NW, int WN, int RT, int MODE, int X3, int W8 = 0>
struct WsCfg {
    static constexpr int NWAVES = (WN > 4 || W8) ? 8 : 4;  // eight waves = two per SIMD, <= 256 registers each
    static constexpr int NTH = 64 * NWAVES;
    static constexpr int WM = NWAVES / WN;
    static constexpr int NTW = NW / (32 * WN);
    static constexpr int KS = K / 2;
    static constexpr int TR = WM * RT * 32;  // rows staged per barrier interval
    static constexpr int LDH = K + 4;        // fp32 tile row stride (floats)
    static constexpr int LDHH = K + 8;       // f16x3 plane row stride (halves)
    static constexpr int PLANE = TR * LDHH;  // halves per plane (hi | lo)
    static constexpr int S16 = K / 16;
    static constexpr bool EDGE = (MODE == WS_EDGE_KNN);
    static constexpr int TILE_FLOATS = X3 ? PLANE : TR * LDH;  // one staging buffer, in floats
    static constexpr int ACC_FLOATS = EDGE ? kAccFloats : 0;
    static constexpr int F4_PER_ROW = K / 4;
    static constexpr int TOTAL_F4 = TR * F4_PER_ROW;
    static constexpr int ITERS = (TOTAL_F4 + NTH - 1) / NTH;
    static constexpr size_t lds_bytes() {
        size_t b = (size_t)(2 * TILE_FLOATS + ACC_FLOATS) * 4;
        if (EDGE) b += (size_t)kMaxRows * 4 + kMaxRows + 40 * 4;
        return b;
    }
    static_assert(K % 8 == 0, "K must be a multiple of 8");
    static_assert(!X3 || K % 16 == 0, "f16x3 needs K % 16 == 0 (one MFMA step = 8 k per lane half)");
    static_assert(NW % (32 * WN) == 0, "NW must split into 32-column tiles per wave");
};

template <int K, int NW, int WN, int RT, int MODE, int X3, int SPLIT_IO, int W8 = 0>
__global__ __launch_bounds__((WN > 4 || W8) ? 512 : 256, (WN > 4 || W8) ? 2 : 1) void k_ws(WsParams p, int n_slices) {
    using C = WsCfg<K, NW, WN, RT, MODE, X3, W8>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* hid = lds;                  // fp32: two buffers of TILE_FLOATS
    _Float16* hidh = (_Float16*)lds;   // f16x3: two buffers of [hi plane | lo plane]
    int* acc_lds = (int*)(lds + 2 * C::TILE_FLOATS);
    int* rows_src = (int*)(lds + 2 * C::TILE_FLOATS + C::ACC_FLOATS);
    uint8_t* rows_dst = (uint8_t*)(rows_src + kMaxRows);
    int* scan = (int*)(rows_dst + kMaxRows);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave % WN, wm = wave / WN, h = lane >> 5, l31 = lane & 31;

    // XCD-aware stream/slice mapping
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
    const int ncol0 = slice * NW + wn * C::NTW * 32;  // first output column of this wave

    // ---- stationary weights -> registers ----------------------------------------------------------------
    float w[X3 ? 1 : C::NTW][X3 ? 1 : C::KS];
    half8 w_hi[X3 ? C::NTW : 1][X3 ? C::S16 : 1], w_lo[X3 ? C::NTW : 1][X3 ? C::S16 : 1];
    if constexpr (X3) {
        // host-packed register image of the whole [K][ldw] matrix (packing.py::pack_f16x3)
        const uint4* wp = (const uint4*)p.W_x3;
        const int plane_u4 = (p.ldw / 32) * C::S16 * 64;
#pragma unroll
        for (int nt = 0; nt < C::NTW; nt++)
#pragma unroll
            for (int s = 0; s < C::S16; s++) {
                const int idx = (((ncol0 / 32 + nt) * C::S16 + s) * 2 + h) * 32 + l31;
                const uint4 a = wp[idx], b = wp[plane_u4 + idx];
                w_hi[nt][s] = __builtin_bit_cast(half8, a);
                w_lo[nt][s] = __builtin_bit_cast(half8, b);
            }
    } else {
#pragma unroll
        for (int nt = 0; nt < C::NTW; nt++)
#pragma unroll
            for (int s = 0; s < C::KS; s++)
                w[nt][s] = p.W[(int64_t)(h * C::KS + s) * p.ldw + ncol0 + nt * 32 + l31];
    }
    float bias[C::NTW];
#pragma unroll
    for (int nt = 0; nt < C::NTW; nt++) bias[nt] = p.bias ? p.bias[ncol0 + nt * 32 + l31] : 0.f;

    // fp16-range guard (f16x3 only, t2p_common.h): the rows these kernels split while staging are SA outputs, whose
    // magnitude the SA kernels report; here only the magnitude of the layer-1 point tables (A_2, A_3) is published, per
    // batch, by the epilogue (guard_publish: one ballot, no atomic below the floor)
    const int k_live = p.k_live > 0 ? p.k_live : K;   // pad columns of SA output rows are never written: do not read them
    f32x4 sa[C::ITERS];                  // staged source rows (in flight behind the MFMA block)
    f32x4 sb[C::EDGE ? C::ITERS : 1];    // staged destination terms (kNN edge mode)

    // kNN edge mode: destinations per group (32; 8 when the call is too small to occupy the CUs with 32-destination groups - a
    // row's MFMA sums do not depend on its tile or slot, so the results are the same bits either way)
    const int knn_group = (C::EDGE && p.knn_group > 0) ? p.knn_group : 32;
    // Issue the global loads of one batch: rows [r0, r0+TR) of group g (n_rows valid rows in the group).
    auto stage_load = [&](int64_t g, int r0, int n_rows) {
#pragma unroll
        for (int it = 0; it < C::ITERS; it++) {
            const int q = it * C::NTH + tid;
            const int lr = q / C::F4_PER_ROW, c4 = q % C::F4_PER_ROW;
            const int r = r0 + lr;
            sa[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (C::EDGE) sb[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (((C::TOTAL_F4 % C::NTH) == 0 || q < C::TOTAL_F4) && (r < n_rows || SPLIT_IO == 1)) {
                if constexpr (C::EDGE) {
                    const int src = rows_src[r];
                    const int dl = rows_dst[r];
                    if (dl != 0xFF) {  // 0xFF = empty neighbour slot (mean aggregation keeps k slots per destination)
                        sa[it] = *(const f32x4*)(p.A + (int64_t)src * p.lda + c4 * 4);
                        sb[it] = *(const f32x4*)(p.Bc + (g * knn_group + dl) * K + c4 * 4);
                    }
                } else if constexpr (SPLIT_IO == 1) {
                    // A arrives as fp16 hi / lo planes: thread chunk q = 16 bytes = 8 halves of one plane row
                    constexpr int CH_PER_PLANE = C::TR * (K / 8);
                    const int pl = q / CH_PER_PLANE, idx = q % CH_PER_PLANE;
                    const int prow = idx / (K / 8), c8 = idx % (K / 8);
                    const _Float16* base = (const _Float16*)(pl ? p.A_lo : p.A_hi);
                    sa[it] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (prow < n_rows) sa[it] = *(const f32x4*)(base + (g * C::TR + prow) * (int64_t)p.lda + c8 * 8);
                } else {
                    if (SPLIT_IO == 2 || c4 * 4 < k_live)   // (GA layer 1, split output: at its register limit, reads whole rows)
                        sa[it] = *(const f32x4*)(p.A + (g * C::TR + r) * (int64_t)p.lda + c4 * 4);
                }
            }
        }
    };
    // VALU part + LDS write of the staged batch.
    float gmax_edge = 0.f;
    auto stage_write = [&](int buf) {
#pragma unroll
        for (int it = 0; it < C::ITERS; it++) {
            const int q = it * C::NTH + tid;
            if ((C::TOTAL_F4 % C::NTH) != 0 && q >= C::TOTAL_F4) break;
            const int lr = q / C::F4_PER_ROW, c4 = q % C::F4_PER_ROW;
            f32x4 v = sa[it];
            if constexpr (C::EDGE) {
                const f32x4 t = sa[it] + sb[it];
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = fmaxf(t[e], 0.f);
            }
            if constexpr (X3 && SPLIT_IO == 1) {
                constexpr int CH_PER_PLANE = C::TR * (K / 8);
                const int pl = q / CH_PER_PLANE, idx = q % CH_PER_PLANE;
                const int prow = idx / (K / 8), c8 = idx % (K / 8);
                *(f32x4*)(hidh + buf * 2 * C::PLANE + pl * C::PLANE + prow * C::LDHH + c8 * 8) = sa[it];
            } else if constexpr (X3) {
                _Float16* dsth = hidh + buf * 2 * C::PLANE;
                if constexpr (C::EDGE)   // fp16-range guard: the kNN edge rows relu(P_i + Q_j) are split here and nowhere reported
                    gmax_edge = fmaxf(fmaxf(gmax_edge, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
                const fp16x2 h01 = cvt_pk_f16(v[0], v[1]), h23 = cvt_pk_f16(v[2], v[3]);
                const fp16x2 l01 = cvt_pk_f16((v[0] - (float)h01[0]) * 2048.f, (v[1] - (float)h01[1]) * 2048.f);
                const fp16x2 l23 = cvt_pk_f16((v[2] - (float)h23[0]) * 2048.f, (v[3] - (float)h23[1]) * 2048.f);
                uint2 ph, pl;
                ph.x = __builtin_bit_cast(uint32_t, h01); ph.y = __builtin_bit_cast(uint32_t, h23);
                pl.x = __builtin_bit_cast(uint32_t, l01); pl.y = __builtin_bit_cast(uint32_t, l23);
                *(uint2*)(dsth + lr * C::LDHH + c4 * 4) = ph;
                *(uint2*)(dsth + C::PLANE + lr * C::LDHH + c4 * 4) = pl;
            } else {
                *(f32x4*)(hid + buf * C::TILE_FLOATS + lr * C::LDH + c4 * 4) = v;
            }
        }
    };
    // [RT x 32 rows] x [K] x [NTW x 32 cols] per wave; the bias rides in the accumulator.
    auto mfma_block = [&](int buf, f32x16 (&acc)[RT][C::NTW]) {
#pragma unroll
        for (int rt = 0; rt < RT; rt++)
#pragma unroll
            for (int nt = 0; nt < C::NTW; nt++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[rt][nt][e] = bias[nt];
        if constexpr (X3) {
            f32x16 accx[RT][C::NTW];  // cross terms hi.lo' + lo'.hi, scaled by 2048
#pragma unroll
            for (int rt = 0; rt < RT; rt++)
#pragma unroll
                for (int nt = 0; nt < C::NTW; nt++)
#pragma unroll
                    for (int e = 0; e < 16; e++) accx[rt][nt][e] = 0.f;
            const _Float16* hrow = hidh + buf * 2 * C::PLANE + ((wm * RT) * 32 + l31) * C::LDHH + h * (K / 2);
            // operands of step s+1 are fetched before the MFMAs of step s (register double buffer); the scheduling
            // barriers keep hipcc from sinking the reads back to their first use (one wave per SIMD here: nothing else
            // would cover the LDS latency)
            half8 a_hi[RT], a_lo[RT], n_hi[RT], n_lo[RT];
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                a_hi[rt] = *(const half8*)(hrow + rt * 32 * C::LDHH);
                a_lo[rt] = *(const half8*)(hrow + C::PLANE + rt * 32 * C::LDHH);
            }
#pragma unroll
            for (int s = 0; s < C::S16; s++) {
                if (s + 1 < C::S16) {
#pragma unroll
                    for (int rt = 0; rt < RT; rt++) {
                        n_hi[rt] = *(const half8*)(hrow + rt * 32 * C::LDHH + (s + 1) * 8);
                        n_lo[rt] = *(const half8*)(hrow + C::PLANE + rt * 32 * C::LDHH + (s + 1) * 8);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rt = 0; rt < RT; rt++)
#pragma unroll
                    for (int nt = 0; nt < C::NTW; nt++) {
                        acc[rt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi[rt], w_hi[nt][s], acc[rt][nt], 0, 0, 0);
                        accx[rt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi[rt], w_lo[nt][s], accx[rt][nt], 0, 0, 0);
                    }
#pragma unroll
                for (int rt = 0; rt < RT; rt++)
#pragma unroll
                    for (int nt = 0; nt < C::NTW; nt++)
                        accx[rt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo[rt], w_hi[nt][s], accx[rt][nt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rt = 0; rt < RT; rt++) {
                    a_hi[rt] = n_hi[rt];
                    a_lo[rt] = n_lo[rt];
                }
            }
#pragma unroll
            for (int rt = 0; rt < RT; rt++)
#pragma unroll
                for (int nt = 0; nt < C::NTW; nt++)
#pragma unroll
                    for (int e = 0; e < 16; e++) acc[rt][nt][e] = fmaf(accx[rt][nt][e], 1.f / 2048.f, acc[rt][nt][e]);
        } else {
            const float* hrow = hid + buf * C::TILE_FLOATS + ((wm * RT) * 32 + l31) * C::LDH + h * C::KS;
            constexpr int NQ = C::KS / 4;
#pragma unroll
            for (int s4 = 0; s4 < NQ; s4++) {
                f32x4 a[RT];
#pragma unroll
                for (int rt = 0; rt < RT; rt++) a[rt] = *(const f32x4*)(hrow + rt * 32 * C::LDH + s4 * 4);
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int rt = 0; rt < RT; rt++)
#pragma unroll
                        for (int nt = 0; nt < C::NTW; nt++)
                            acc[rt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rt][j], w[nt][s4 * 4 + j], acc[rt][nt],
                                                                                0, 0, 0);
            }
        }
    };

    if constexpr (!C::EDGE) {
        // ---- dense streams: every group is one batch of TR rows; the pipeline runs across groups --------------------
        int64_t g = stream;
        auto rows_of = [&](int64_t gg) -> int {
            const int64_t left = p.M - gg * C::TR;
            return (int)(left < C::TR ? left : C::TR);
        };
        if (g < p.n_groups) {
            stage_load(g, 0, rows_of(g));
            stage_write(0);
        }
        __syncthreads();
        float gmax_all = 0.f;   // largest magnitude this thread stored over the whole launch (f16x3 table kernels only)
        (void)gmax_all;
        for (int i = 0; g < p.n_groups; g += n_streams, i++) {
            const int64_t gn = g + n_streams;
            const int n_rows = rows_of(g);
            if (gn < p.n_groups) stage_load(gn, 0, rows_of(gn));
            f32x16 acc[RT][C::NTW];
            mfma_block(i & 1, acc);
            float gmax_out = 0.f;
            (void)gmax_out;
            // (a full batch - every batch but possibly the last of a launch - stores without the per-row bound: 16 compares and
            // 16 exec-mask save / branch / restore sequences less per tile)
            auto epilogue = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                const int trow0 = (wm * RT + rt) * 32;
#pragma unroll
                for (int nt = 0; nt < C::NTW; nt++) {
                    const int lcol = wn * C::NTW * 32 + nt * 32 + l31;
                    if constexpr (MODE == WS_DENSE_STORE) {
                        // rows of this lane: trow0 + 4h + (e & 3) + 8 (e >> 2); one base pointer, uniform row steps
                        const int64_t o0 = (g * C::TR + trow0 + 4 * h) * (int64_t)p.ldo + slice * NW + lcol;
#pragma unroll
                        for (int e = 0; e < 16; e++) {
                            const int rr = (e & 3) + 8 * (e >> 2);
                            const int r = trow0 + 4 * h + rr;
                            float v = acc[rt][nt][e];
                            if (p.relu) v = fmaxf(v, 0.f);
                            // (rows past M hold the bias only).  Not for the split-output form (GA layer 1): that kernel sits
                            // at the register limit with a store-bound epilogue (a running maximum cost 14 %); its output
                            // is bounded from its input's magnitude instead (k_guard_check)
                            if constexpr (X3 && SPLIT_IO != 2) gmax_out = fmaxf(gmax_out, fabsf(v));
                            if constexpr (SPLIT_IO == 2) {  // hand the activations on already split into fp16 hi / lo
                                const fp16x2 hv = cvt_pk_f16(v, 0.f);
                                const fp16x2 lv = cvt_pk_f16((v - (float)hv[0]) * 2048.f, 0.f);
#if T2P_GABL & 1
                                if (FULL || r < n_rows) {      // same instructions, 4 MB footprint: no HBM write traffic
                                    ((__fp16*)p.out_hi + (o0 & 0xFFFFF))[rr * p.ldo] = hv[0];
                                    ((__fp16*)p.out_lo + (o0 & 0xFFFFF))[rr * p.ldo] = lv[0];
                                }
#else
                                if (FULL || r < n_rows) {
                                    ((__fp16*)p.out_hi + o0)[rr * p.ldo] = hv[0];
                                    ((__fp16*)p.out_lo + o0)[rr * p.ldo] = lv[0];
                                }
#endif
                            } else {
                                if (FULL || r < n_rows) (p.out + o0)[rr * p.ldo] = v;
                            }
                        }
                    } else {  // max over each 32-row tile = one object (ReLU = starting the max at 0)
                        float m = 0.f;
#pragma unroll
                        for (int e = 0; e < 16; e++) m = fmaxf(m, acc[rt][nt][e]);
                        m = fmaxf(m, __shfl_xor(m, 32, 64));
                        if (h == 0 && (FULL || trow0 < n_rows))
                            p.out[(g * (C::TR / 32) + wm * RT + rt) * (int64_t)p.ldo + slice * NW + lcol] = m;
                    }
                }
            }
            };
            if (n_rows == C::TR) epilogue(std::true_type{});
            else epilogue(std::false_type{});
            if constexpr (X3 && MODE == WS_DENSE_STORE && SPLIT_IO != 2) gmax_all = fmaxf(gmax_all, gmax_out);
            if (gn < p.n_groups) stage_write((i + 1) & 1);
            __syncthreads();
        }
        // the table's exact largest magnitude, once per wave (high AND low side of the guard)
        if constexpr (X3 && MODE == WS_DENSE_STORE && SPLIT_IO != 2) guard_publish_exact(p.amax_out, gmax_all);
    } else {
        // ---- kNN edge stream: a group = 32 destination objects, rows = their valid neighbours ----------------------
        for (int64_t g = stream; g < p.n_groups; g += n_streams) {
            const int64_t d0 = g * knn_group;
            const int nd = (int)((p.n_dst - d0) < knn_group ? (p.n_dst - d0) : knn_group);
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
            // max: rows = the valid neighbours, compacted; mean (knn_k == 8): exactly 8 slots per destination so that a
            // destination's rows are one aligned 8-row group of a tile and its sum has a fixed, deterministic order
            const int n_rows = p.mean ? nd * 8 : scan[32];
            if (tid < nd) {
                int off = p.mean ? tid * 8 : scan[tid];
                for (int e = 0; e < p.knn_k; e++) {
                    int j = p.knn_idx[(d0 + tid) * p.knn_k + e];
                    if (j >= 0) { rows_src[off] = j; rows_dst[off] = (uint8_t)tid; off++; }
                    else if (p.mean) { rows_src[off] = 0; rows_dst[off] = 0xFF; off++; }
                }
            }
            for (int i = tid; i < nd * NW; i += C::NTH) acc_lds[i] = 0;
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
                // segmented max (rows sorted by destination; the signed-int atomic max against +0 is the ReLU)
#pragma unroll
                for (int rt = 0; rt < RT; rt++) {
                    const int trow0 = bt * C::TR + (wm * RT + rt) * 32;
                    if (trow0 >= n_rows) continue;
                    if (p.mean) {  // DynamicEdgeConv(aggr="mean"): sum of ReLU'd rows / valid neighbours, fixed order
#pragma unroll
                        for (int nt = 0; nt < C::NTW; nt++) {
                            const int lcol = wn * C::NTW * 32 + nt * 32 + l31;
#pragma unroll
                            for (int q = 0; q < 4; q++) {
                                const int d = trow0 / 8 + q;  // destination inside the group
                                float sum = 0.f;
#pragma unroll
                                for (int e = 0; e < 4; e++) {
                                    const int r = trow0 + 8 * q + 4 * h + e;
                                    const bool ok = r < n_rows && rows_dst[r] != 0xFF;
                                    sum += ok ? fmaxf(acc[rt][nt][4 * q + e], 0.f) : 0.f;
                                }
                                const float other = __shfl_xor(sum, 32, 64);
                                sum = h == 0 ? sum + other : other + sum;  // rows 0-3 first, then rows 4-7
                                if (h == 0 && d < nd) {
                                    const int cnt = scan[d + 1] - scan[d];
                                    acc_lds[d * NW + lcol] = __float_as_int(cnt > 0 ? sum / (float)cnt : 0.f);
                                }
                            }
                        }
                        continue;
                    }
                    int dd[16];
                    bool same[16], is_end[16];
#pragma unroll
                    for (int e = 0; e < 16; e++) {
                        const int r = trow0 + 8 * (e >> 2) + 4 * h + (e & 3);
                        dd[e] = r < n_rows ? (int)rows_dst[r] : -1;
                    }
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
                            v[e] = acc[rt][nt][e];
                            if (e > 0) v[e] = same[e] ? fmaxf(v[e], v[e - 1]) : v[e];
                        }
#pragma unroll
                        for (int e = 0; e < 16; e++)
                            if (is_end[e]) atomicMax(&acc_lds[dd[e] * NW + lcol], __float_as_int(v[e]));
                    }
                }
                if (more) stage_write((bt + 1) & 1);
                __syncthreads();
            }
            for (int i = tid; i < nd * NW; i += C::NTH) {
                const int c = i / NW, col = i % NW;
                p.out[(d0 + c) * (int64_t)p.ldo + slice * NW + col] = __int_as_float(acc_lds[i]);   // (this slice's columns)
            }
            __syncthreads();
        }
        if constexpr (X3) guard_publish(p.amax_out, gmax_edge);
    }
}

template <int K, int NW, int WN, int RT, int MODE, int X3, int SPLIT_IO, int W8 = 0>
int launch_cfg(const WsParams& p_in, int n_slices, hipStream_t st) {
    using C = WsCfg<K, NW, WN, RT, MODE, X3, W8>;
    auto kern = k_ws<K, NW, WN, RT, MODE, X3, SPLIT_IO, W8>;
    T2P_TRY(reserve_lds((const void*)kern, C::lds_bytes(), "ws_gemm"));
    WsParams p = p_in;
    if (MODE != WS_EDGE_KNN) p.n_groups = (p.M + C::TR - 1) / C::TR;  // dense: one group = one batch of TR rows
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
    T2P_REPEAT(ps_) hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NTH), C::lds_bytes(), st, p, n_slices);
    T2P_CHECK_LAUNCH("ws_gemm");
    return 0;
}

}  // namespace

// mode DENSE_STORE: out[M][ldo] = act(A[M][K] W + b);  DENSE_GROUPMAX: out[M/32][ldo] = max over each 32-row group
// (M = 32 * groups);  EDGE_KNN: see WsParams.
int launch_ga2(const WsParams& p, hipStream_t st);  // ga2.hip

int launch_ws(int mode, int K, int N, const WsParams& p, hipStream_t st) {
    T2P_CHECK_ARG((((uintptr_t)p.A | (uintptr_t)p.A_hi | (uintptr_t)p.A_lo) & 15) == 0 && (p.lda % 4) == 0,
                  "ws_gemm: A must be 16-byte aligned, lda %% 4 == 0");
    const int x3 = p.W_x3 != nullptr ? 1 : 0;
    if (x3) T2P_CHECK_ARG(((uintptr_t)p.W_x3 & 15) == 0, "ws_gemm: packed f16x3 weights must be 16-byte aligned");
    if (mode == WS_DENSE_GROUPMAX) T2P_CHECK_ARG(p.M % 32 == 0, "ws_gemm: groupmax needs M %% 32 == 0");
    // split_io: 0 = fp32 in / fp32 out, 1 = fp16 hi/lo planes in, 2 = fp16 hi/lo planes out (f16x3 only)
    const int split_io = p.A_hi != nullptr ? 1 : (p.out_hi != nullptr ? 2 : 0);
    T2P_CHECK_ARG(split_io == 0 || x3 == 1, "ws_gemm: split fp16 activations need the f16x3 path");
    T2P_CHECK_ARG(p.k_live == 0 || (split_io == 0 && mode == WS_DENSE_STORE && p.k_live % 4 == 0),
                  "ws_gemm: k_live is honoured by the fp32-in / fp32-out dense mode only (multiple of 4)");
    if (split_io == 1) T2P_CHECK_ARG(p.A_lo != nullptr && p.lda % 8 == 0, "ws_gemm: split input needs both planes, lda %% 8 == 0");
    if (split_io == 2) T2P_CHECK_ARG(p.out_lo != nullptr, "ws_gemm: split output needs both planes");
#define WS_CASE(MODE_, K_, N_, NW_, WN_, RT_, X3_, SIO_)                                      \
    if (mode == MODE_ && K == K_ && N == N_ && x3 == X3_ && split_io == SIO_)                  \
        return launch_cfg<K_, NW_, WN_, RT_, MODE_, X3_, SIO_>(p, N_ / NW_, st);
    // DynamicEdgeConv layer 2 (fp32 / f16x3)
    WS_CASE(WS_EDGE_KNN, 256, 256, 256, 8, 1, 0, 0)  // 8 waves, one 32-column block each (128 weight registers)
    WS_CASE(WS_EDGE_KNN, 256, 256, 256, 8, 1, 1, 0)
    WS_CASE(WS_EDGE_KNN, 128, 128, 128, 4, 1, 0, 0)  // embed_dim 128: 4 waves, one 32-column block each
    WS_CASE(WS_EDGE_KNN, 128, 128, 128, 4, 1, 1, 0)
    WS_CASE(WS_EDGE_KNN, 384, 384, 128, 4, 1, 0, 0)  // embed_dim 300 zero-padded to 384: three 128-column slices, 4 waves each
    WS_CASE(WS_EDGE_KNN, 384, 384, 128, 4, 1, 1, 0)
    // SA2 / SA3 layer-1 point tables ([feat | pos | zero pad] -> H) and GA layer 1
    WS_CASE(WS_DENSE_STORE, 96, 128, 128, 4, 2, 0, 0)
    WS_CASE(WS_DENSE_STORE, 160, 256, 256, 4, 1, 0, 0)
    WS_CASE(WS_DENSE_STORE, 288, 512, 128, 4, 1, 0, 0)
    // f16x3: K = C + 16 (the rows' last 16 pad columns have zero weights: not staged, not multiplied)
    if (mode == WS_DENSE_STORE && K == 80 && N == 128 && x3 == 1 && split_io == 0)  // 8 waves: 4 column blocks x 2 row tiles
        return launch_cfg<80, 128, 4, 1, WS_DENSE_STORE, 1, 0, 1>(p, 1, st);
    WS_CASE(WS_DENSE_STORE, 144, 256, 256, 8, 1, 1, 0)  // 8 waves (two per SIMD), one 32-column block per wave
    WS_CASE(WS_DENSE_STORE, 272, 512, 256, 8, 1, 1, 2)  // 8 waves: two per SIMD, two column slices instead of four
    // GA layer 2 + max over the 32 points of an object
    WS_CASE(WS_DENSE_GROUPMAX, 512, 1024, 128, 4, 1, 0, 0)
#if !T2P_GA2_V1
    if (mode == WS_DENSE_GROUPMAX && K == 512 && N == 1024 && x3 == 1 && split_io == 1) return launch_ga2(p, st);
#endif
    WS_CASE(WS_DENSE_GROUPMAX, 512, 1024, 128, 4, 1, 1, 1)
#undef WS_CASE
    set_error("ws_gemm: no instantiation for mode=%d K=%d N=%d x3=%d split_io=%d", mode, K, N, x3, split_io);
    return T2P_E_UNSUPPORTED;
}

}  // namespace t2p

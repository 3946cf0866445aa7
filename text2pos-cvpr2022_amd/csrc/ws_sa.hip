// Set-abstraction edge kernel, exact-fp32 path (precision = "fp32": v_mfma_f32_32x32x2_f32 fma chains): per-edge
// ReLU(A_j - B_i) -> layer-2 GEMM -> max per centroid, for sa1/sa2/sa3.  The f16x3 path runs on sa_points.hip / sa_rows.hip / sa3.hip (levels 1 / 2 / 3); this file keeps the exact fp32 MFMA kernels and the range balancing.
// (reference: gnn.PointConv(local_nn)(x, (pos, pos[idx]), edge_index), models/pointcloud/pointnet2.py:31-35).
//
// Same arithmetic and register-resident weights as ws_gemm.hip (see the design notes there); this variant removes
// every per-object bubble of the generic edge path:
//   * the per-object edge-row lists come pre-compacted from the FPS/ball-query kernel (one u16 per row), so there is
//     no in-kernel enumeration / scan;
//   * a tiny balancing kernel gives every workgroup a CONTIGUOUS object range of (nearly) equal tile count, computed
//     from the row counts -- deterministic, no atomics, no tail imbalance;
//   * the workgroup walks its range as ONE flattened stream of row batches that crosses object boundaries, software
//     pipelined three deep:  row metadata (t+2)  ->  gathers of A_j / B_i rows (t+1)  ->  MFMA + segmented max (t),
//     with double-buffered LDS staging tiles and one barrier per batch;
//   * the per-object max accumulator is double-buffered in LDS too, so the finished object's [n_cent][C] block is
//     written to HBM (and re-zeroed) underneath the MFMAs of the next object's first batch.
#include "t2p_common.h"

namespace t2p {
namespace {

constexpr int kSub = 512;
constexpr int NT = 512;   // threads per workgroup: 8 waves = 2 per SIMD, so one wave's VALU/LDS phases overlap the other's MFMAs  // objects whose row counts / self-loop bases are cached in LDS at a time

template <int K, int N, int WN, int RT>
struct SaCfg {
    static constexpr int WM = 8 / WN;
    static constexpr int NTW = N / (32 * WN);
    static constexpr int KS = K / 2;
    static constexpr int TR = WM * RT * 32;
    static constexpr int LDH = K + 4;
    static constexpr int HID_FLOATS = TR * LDH;
    static constexpr int ACC_INTS = 8192 + N;  // n_cent * N (128x64, 64x128, 32x256) + one dummy row for padding rows
    static constexpr int F4_PER_ROW = K / 4;
    static constexpr int TOTAL_F4 = TR * F4_PER_ROW;
    static constexpr int ITERS = TOTAL_F4 / NT;
    static_assert(TOTAL_F4 % NT == 0, "staging must divide evenly over the workgroup");
    static constexpr size_t lds_bytes() {
        const size_t tile = (size_t)HID_FLOATS * 4;
        return 2 * tile + (size_t)2 * ACC_INTS * 4 + 2 * TR + kSub * 2 + kSub * 4;
    }
};

// Tile-count prefix sums over the objects and balanced contiguous ranges for n_wg workgroups.  One block.
__device__ __forceinline__ void balance_body(const uint16_t* __restrict__ n_rows, int n, int tile_rows, int n_wg,
                                             int32_t* __restrict__ prefix, int32_t* __restrict__ bounds) {
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int lo = tid * per, hi = (lo + per) < n ? (lo + per) : n;
    int s = 0;
    for (int i = lo; i < hi; i++) s += ((int)n_rows[i] + tile_rows - 1) / tile_rows;
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - s;  // exclusive
    for (int i = lo; i < hi; i++) {
        prefix[i] = run;
        run += ((int)n_rows[i] + tile_rows - 1) / tile_rows;
    }
    if (tid == 1023) prefix[n] = part[1023];
    __syncthreads();
    const int total = part[1023];
    for (int b = tid; b <= n_wg; b += 1024) {
        // first object whose prefix >= b * total / n_wg
        const long long target = ((long long)b * total) / n_wg;
        int l = 0, r = n;
        while (l < r) {
            const int m = (l + r) >> 1;
            if (prefix[m] < target) l = m + 1; else r = m;
        }
        bounds[b] = b == n_wg ? n : l;
    }
}

__global__ __launch_bounds__(1024) void k_balance(const uint16_t* __restrict__ n_rows, int n, int tile_rows, int n_wg,
                                                  int32_t* __restrict__ prefix, int32_t* __restrict__ bounds) {
    balance_body(n_rows, n, tile_rows, n_wg, prefix, bounds);
}

struct BalanceJobs {
    const uint16_t* n_rows[3];
    int32_t* prefix[3];
    int32_t* bounds[3];
    int tile_rows[3], n_wg[3];
    int n;
};
__global__ __launch_bounds__(1024) void k_balance_levels(BalanceJobs j) {  // one block per level
    const int l = blockIdx.x;
    balance_body(j.n_rows[l], j.n, j.tile_rows[l], j.n_wg[l], j.prefix[l], j.bounds[l]);
}

struct BatchIt {  // position in the flattened batch stream of a sub-range
    int gi;       // object index inside the cached sub-range
    int r0;       // first row of the batch inside the object
    int n;        // rows of the object
};

template <int K, int N, int WN, int RT>
__global__ __launch_bounds__(NT, 2) void k_ws_sa(SaParams p) {
    using C = SaCfg<K, N, WN, RT>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* hid = lds;                                       // fp32: [2][HID_FLOATS]
    constexpr int TILE_FLOATS = C::HID_FLOATS;
    int* acc_lds = (int*)(lds + 2 * TILE_FLOATS);           // [2][ACC_INTS]
    uint8_t* dstl = (uint8_t*)(acc_lds + 2 * C::ACC_INTS);  // [2][TR] destination (centroid) of every staged row
    uint16_t* nr = (uint16_t*)(dstl + 2 * C::TR);           // [kSub] rows per object
    int* sbase = (int*)(nr + kSub);                         // [kSub] source row of centroid 0's self loop

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave % WN, wm = wave / WN, h = lane >> 5, l31 = lane & 31;
    const int nc = p.n_cent;
    const int maxr = nc * 33;

    float w[C::NTW][C::KS];
    #pragma unroll
        for (int nt = 0; nt < C::NTW; nt++)
#pragma unroll
            for (int s = 0; s < C::KS; s++)
                w[nt][s] = p.W[(int64_t)(h * C::KS + s) * N + wn * C::NTW * 32 + nt * 32 + l31];
    
    float bias[C::NTW];
#pragma unroll
    for (int nt = 0; nt < C::NTW; nt++) bias[nt] = p.bias[wn * C::NTW * 32 + nt * 32 + l31];

    for (int i = tid; i < 2 * C::ACC_INTS; i += NT) acc_lds[i] = 0;

    const int g_begin = p.bounds_ws[blockIdx.x], g_end = p.bounds_ws[blockIdx.x + 1];

    for (int ga = g_begin; ga < g_end; ga += kSub) {
        const int cnt = (g_end - ga) < kSub ? (g_end - ga) : kSub;
        __syncthreads();
        for (int i = tid; i < cnt; i += NT) {
            const int g = ga + i;
            nr[i] = p.n_rows[g];
            const int first = p.first[g];
            sbase[i] = first * p.n_dense + (g - first) * nc;
        }
        __syncthreads();

        auto advance = [&](BatchIt it) -> BatchIt {
            it.r0 += C::TR;
            if (it.r0 >= it.n) {
                it.gi++;
                it.r0 = 0;
                it.n = it.gi < cnt ? (int)nr[it.gi] : 0;
            }
            return it;
        };
        auto valid = [&](const BatchIt& it) { return it.gi < cnt; };

        // Each thread stages ITERS consecutive rows (lr = (tid / F4_PER_ROW) * ITERS + k) at a fixed column quad, so its
        // row metadata is ONE aligned vector load of ITERS u16.
        static_assert(C::ITERS == 2 || C::ITERS == 4, "metadata vector is 4 or 8 bytes");
        const int rgrp = (tid / C::F4_PER_ROW) * C::ITERS;   // first staged row of this thread inside the batch
        const int c4 = tid % C::F4_PER_ROW;
        typedef uint16_t metav __attribute__((ext_vector_type(C::ITERS)));
        metav meta_d, meta_m;                                 // metadata of the batch being gathered / the one after
        f32x4 sa[C::ITERS], sb[C::ITERS];

        // M: metadata of one batch -> registers (0xFFFF = padding row).  Split in two: the load is ISSUED at the top of a
        // batch, the clean-up of the rows past the object's end runs after the MFMA block -- touching the loaded value
        // any earlier puts an s_waitcnt vmcnt(0) in front of the MFMAs, which also waits for the gathers just issued
        // (measured: that exposed the whole gather latency, 28 % of the SA3 kernel).
        auto load_meta = [&](const BatchIt& it, metav& m) {
#pragma unroll
            for (int k = 0; k < C::ITERS; k++) m[k] = 0xFFFF;
            if (valid(it) && it.r0 + rgrp < it.n) {
                const uint32_t off = (uint32_t)(ga + it.gi) * (uint32_t)maxr + (uint32_t)(it.r0 + rgrp);
                m = *(const metav*)(p.rows + off);
            }
        };
        auto fix_meta = [&](const BatchIt& it, metav& m) {
#pragma unroll
            for (int k = 0; k < C::ITERS; k++)
                if (it.r0 + rgrp + k >= it.n) m[k] = 0xFFFF;
        };
        // D: gathers of the batch's A_j and B_i rows -> registers (32-bit element offsets from uniform bases)
        auto load_data = [&](const BatchIt& it, const metav& m) {
            const uint32_t g = (uint32_t)(ga + it.gi);
            const uint32_t sb0 = valid(it) ? (uint32_t)sbase[it.gi] : 0u;
#pragma unroll
            for (int k = 0; k < C::ITERS; k++) {
                sa[k] = f32x4{0.f, 0.f, 0.f, 0.f};
                sb[k] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (m[k] != 0xFFFF) {
                    const uint32_t src = m[k] & 0xFF, d = m[k] >> 8, dl = d & 127;
                    const uint32_t srow = (d & 0x80) ? (sb0 + src) : (g * (uint32_t)p.n_dense + src);
                    sa[k] = *(const f32x4*)(p.A + (srow * (uint32_t)K + (uint32_t)c4 * 4u));
                    sb[k] = *(const f32x4*)(p.Bc + ((g * (uint32_t)nc + dl) * (uint32_t)K + (uint32_t)c4 * 4u));
                }
            }
        };
        // W: h = relu(A_j - B_i) -> LDS tile, plus the destination byte of every row
        auto write_tile = [&](int buf, const metav& m) {
            float* dst = hid + buf * C::HID_FLOATS;
#pragma unroll
            for (int k = 0; k < C::ITERS; k++) {
                const int lr = rgrp + k;
                const f32x4 t = sa[k] - sb[k];
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = fmaxf(t[e], 0.f);
                                    *(f32x4*)(dst + lr * C::LDH + c4 * 4) = v;
                
                // destination of the row; padding rows go to the accumulator's dummy row n_cent
                if (c4 == 0) dstl[buf * C::TR + lr] = m[k] == 0xFFFF ? (uint8_t)nc : (uint8_t)((m[k] >> 8) & 127);
            }
        };
        // flush one finished object's accumulator (feature columns; the [xyz 0] quad of the rows is written
        // by the centroid-table kernel), re-zero
        auto flush = [&](int64_t g, int abuf) {
            int* a = acc_lds + abuf * C::ACC_INTS;
            float* o = p.out + g * nc * (int64_t)p.ldo;
            for (int i = tid; i < nc * N; i += NT) {
                const int c = i / N, col = i % N;
                o[c * (int64_t)p.ldo + col] = __int_as_float(a[i]) * p.out_scale;
                a[i] = 0;
            }
        };

        BatchIt it_c{0, 0, cnt > 0 ? (int)nr[0] : 0};
        BatchIt it_d = advance(it_c);
        BatchIt it_m = advance(it_d);
        // prologue: M(0), M(1), D(0), W(0)
        load_meta(it_c, meta_d);
        load_meta(it_d, meta_m);
        fix_meta(it_c, meta_d);
        fix_meta(it_d, meta_m);
        load_data(it_c, meta_d);
        write_tile(0, meta_d);
        meta_d = meta_m;
        __syncthreads();

        int64_t flush_g = -1;
        int flush_buf = 0;
        for (int t = 0; valid(it_c); t++) {
            // the object finished in the previous batch drains to HBM underneath this batch's MFMAs
            if (flush_g >= 0) {
                flush(flush_g, flush_buf);
                flush_g = -1;
            }
            if (valid(it_d)) load_data(it_d, meta_d);      // D(t+1): gathers go out first ...
            load_meta(it_m, meta_m);                       // M(t+2): ... the younger metadata loads stay in flight

            // C(t): MFMA block on tile t & 1
            f32x16 acc[RT][C::NTW];
#pragma unroll
            for (int rt = 0; rt < RT; rt++)
#pragma unroll
                for (int nt = 0; nt < C::NTW; nt++)
#pragma unroll
                    for (int e = 0; e < 16; e++) {
                        acc[rt][nt][e] = bias[nt];  // bias rides in the accumulator
                    }
            const int buf = t & 1;
                            const float* hrow = hid + buf * C::HID_FLOATS + ((wm * RT) * 32 + l31) * C::LDH + h * C::KS;
                constexpr int QC = 4;                 // k-quads (16 k-steps) fetched per LDS round
                constexpr int NCH = C::KS / 4 / QC;   // chunks
                static_assert((C::KS / 4) % QC == 0, "K/8 must be a multiple of the LDS prefetch chunk");
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
            
            // max-aggregation: every accumulator row goes straight to its destination's LDS slot with an integer atomic
            // max (non-returning; the signed-int max against +0 is also the ReLU).  No run detection, no branches:
            // padding rows carry destination n_cent = the accumulator's dummy row.
            const int abuf = it_c.gi & 1;
            int* accb = acc_lds + abuf * C::ACC_INTS;
            const uint8_t* dl = dstl + buf * C::TR;
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                const int trow0 = (wm * RT + rt) * 32;
                if (it_c.r0 + trow0 >= it_c.n) continue;
                int doff[16];  // destination row offsets (ints) of this lane's 16 rows: 4 quads of 4 consecutive rows
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t four = *(const uint32_t*)(dl + trow0 + 8 * q + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; e++) doff[4 * q + e] = (int)((four >> (8 * e)) & 0xFF) * N;
                }
#pragma unroll
                for (int nt = 0; nt < C::NTW; nt++) {
                    int* col = accb + wn * C::NTW * 32 + nt * 32 + l31;
#pragma unroll
                    for (int e = 0; e < 16; e++) atomicMax(col + doff[e], __float_as_int(acc[rt][nt][e]));
                }
            }
            if (it_c.r0 + C::TR >= it_c.n) {  // last batch of this object
                flush_g = ga + it_c.gi;
                flush_buf = abuf;
            }
            if (valid(it_d)) write_tile((t + 1) & 1, meta_d);  // W(t+1)
            fix_meta(it_m, meta_m);
            meta_d = meta_m;
            it_c = it_d;
            it_d = it_m;
            it_m = advance(it_m);
            __syncthreads();
        }
        if (flush_g >= 0) flush(flush_g, flush_buf);
    }
}

template <int K, int N, int WN, int RT>
int launch_sa_cfg(const SaParams& p_in, hipStream_t st, const char* name) {
    using C = SaCfg<K, N, WN, RT>;
    auto kern = k_ws_sa<K, N, WN, RT>;
    T2P_TRY(reserve_lds((const void*)kern, C::lds_bytes(), "ws_sa"));
    if (p_in.n_obj <= 0) return 0;
    T2P_CHECK_ARG(p_in.n_obj < (1 << 30) && p_in.n_obj * p_in.n_dense < 0x7fffffffLL, "ws_sa: chunk too large for 32-bit rows");
    int n_wg = num_cus();
    if (n_wg > p_in.n_obj) n_wg = (int)p_in.n_obj;
    const SaParams& p = p_in;
    if (!p.balanced) {
        ProfScope ps_("sa_balance", st);
        hipLaunchKernelGGL(k_balance, dim3(1), dim3(1024), 0, st, p.n_rows, (int)p.n_obj, C::TR, n_wg, p.prefix_ws,
                           p.bounds_ws);
        T2P_CHECK_LAUNCH("sa_balance");
    }
    ProfScope ps_(name, st);
    hipLaunchKernelGGL(kern, dim3(n_wg), dim3(NT), C::lds_bytes(), st, p);
    T2P_CHECK_LAUNCH("ws_sa");
    return 0;
}

}  // namespace


int launch_sa_balance(const SaParams& p, int tile_rows, int n_wg, hipStream_t st) {
    ProfScope ps_("sa_balance", st);
    hipLaunchKernelGGL(k_balance, dim3(1), dim3(1024), 0, st, p.n_rows, (int)p.n_obj, tile_rows, n_wg, p.prefix_ws,
                       p.bounds_ws);
    T2P_CHECK_LAUNCH("sa_balance");
    return 0;
}


// tile rows / workgroup count of the kernel launch_ws_sa will pick for (H, Cout)
static int sa_launch_shape(int H, int Cout, const SaParams& p, int64_t n_obj, int* tile_rows, int* n_wg) {
    if (sa_rows_selected(H, Cout, p)) return sa_rows_launch_shape(n_obj, tile_rows, n_wg);
    if (sa_points_selected(H, Cout, p)) return sa_points_launch_shape(n_obj, tile_rows, n_wg);
    if (sa3_selected(H, Cout, p)) return sa3_launch_shape(n_obj, tile_rows, n_wg);
    if (p.W_x3 != nullptr) {
        set_error("ws_sa: no f16x3 kernel for H=%d C=%d (built: 32/64 over 256 points, 128/128, 256/256, each with the LDS centroid table)", H, Cout);
        return T2P_E_UNSUPPORTED;
    }
    int n = num_cus();
    if (n > n_obj) n = (int)n_obj;
    *n_wg = n;
    if (H == 32 && Cout == 64) { *tile_rows = SaCfg<32, 64, 2, 2>::TR; return 0; }
    if (H == 128 && Cout == 128) { *tile_rows = SaCfg<128, 128, 4, 1>::TR; return 0; }
    if (H == 256 && Cout == 256) { *tile_rows = SaCfg<256, 256, 8, 1>::TR; return 0; }
    set_error("ws_sa: no instantiation for H=%d C=%d", H, Cout);
    return T2P_E_UNSUPPORTED;
}

int launch_sa_balance_levels(const SaParams p[3], const int H[3], const int C[3], hipStream_t st) {
    if (p[0].n_obj <= 0) return 0;
    BalanceJobs j;
    j.n = (int)p[0].n_obj;
    for (int l = 0; l < 3; l++) {
        j.n_rows[l] = p[l].n_rows;
        j.prefix[l] = p[l].prefix_ws;
        j.bounds[l] = p[l].bounds_ws;
        T2P_TRY(sa_launch_shape(H[l], C[l], p[l], p[l].n_obj, &j.tile_rows[l], &j.n_wg[l]));
    }
    ProfScope ps_("sa_balance", st);
    hipLaunchKernelGGL(k_balance_levels, dim3(3), dim3(1024), 0, st, j);
    T2P_CHECK_LAUNCH("sa_balance");
    return 0;
}

int launch_ws_sa(int H, int Cout, const SaParams& p, hipStream_t st) {
    T2P_CHECK_ARG((((uintptr_t)p.A | (uintptr_t)p.Bc) & 15) == 0, "ws_sa: tables must be 16-byte aligned");
    if (p.W_x3 != nullptr) {  // f16x3 split-precision path: one kernel per level
        T2P_CHECK_ARG(((uintptr_t)p.W_x3 & 15) == 0, "ws_sa: packed f16x3 weights must be 16-byte aligned");
        if (sa_rows_selected(H, Cout, p)) return launch_sa_rows(H, Cout, p, st);
        if (sa_points_selected(H, Cout, p)) return launch_sa_points(H, Cout, p, st);
        if (sa3_selected(H, Cout, p)) return launch_sa3(p, st);
        set_error("ws_sa: no f16x3 kernel for H=%d C=%d (built: 32/64 over 256 points, 128/128, 256/256, each with the LDS centroid table)", H, Cout);
        return T2P_E_UNSUPPORTED;
    } else {
        if (H == 32 && Cout == 64) return launch_sa_cfg<32, 64, 2, 2>(p, st, "ws_edge_sa_k32_n64");
        if (H == 128 && Cout == 128) return launch_sa_cfg<128, 128, 4, 1>(p, st, "ws_edge_sa_k128_n128");
        if (H == 256 && Cout == 256) return launch_sa_cfg<256, 256, 8, 1>(p, st, "ws_edge_sa_k256_n256");
    }
    set_error("ws_sa: no instantiation for H=%d C=%d", H, Cout);
    return T2P_E_UNSUPPORTED;
}

}  // namespace t2p

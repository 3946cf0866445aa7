// Set-abstraction edge kernel: per-edge ReLU(A_j - B_i) -> layer-2 GEMM -> max per centroid, for sa1/sa2/sa3
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
#include <stdlib.h>
#ifndef T2P_SCHED_BARRIER
#define T2P_SCHED_BARRIER 0
#endif
#ifndef T2P_ANY_SKIP
#define T2P_ANY_SKIP 0
#endif

#include "t2p_common.h"

namespace t2p {
namespace {

constexpr int kSub = 512;  // objects whose row counts / self-loop bases are cached in LDS at a time

template <int K, int N, int WN, int RT>
struct SaCfg {
    static constexpr int WM = 4 / WN;
    static constexpr int NTW = N / (32 * WN);
    static constexpr int KS = K / 2;
    static constexpr int TR = WM * RT * 32;
    static constexpr int LDH = K + 4;
    static constexpr int HID_FLOATS = TR * LDH;
    static constexpr int ACC_INTS = 8192;  // n_cent * N for all three levels (128x64, 64x128, 32x256)
    static constexpr int F4_PER_ROW = K / 4;
    static constexpr int TOTAL_F4 = TR * F4_PER_ROW;
    static constexpr int ITERS = TOTAL_F4 / 256;
    static_assert(TOTAL_F4 % 256 == 0, "staging must divide evenly over 256 threads");
    static constexpr size_t lds_bytes() {
        return (size_t)(2 * HID_FLOATS + 2 * ACC_INTS) * 4 + 2 * TR + kSub * 2 + kSub * 4;
    }
};

// Tile-count prefix sums over the objects and balanced contiguous ranges for n_wg workgroups.  One block.
__global__ __launch_bounds__(1024) void k_balance(const uint16_t* __restrict__ n_rows, int n, int tile_rows, int n_wg,
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

struct BatchIt {  // position in the flattened batch stream of a sub-range
    int gi;       // object index inside the cached sub-range
    int r0;       // first row of the batch inside the object
    int n;        // rows of the object
};

template <int K, int N, int WN, int RT>
__global__ __launch_bounds__(256, 1) void k_ws_sa(SaParams p) {
    using C = SaCfg<K, N, WN, RT>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* hid = lds;                                       // [2][HID_FLOATS]
    int* acc_lds = (int*)(lds + 2 * C::HID_FLOATS);         // [2][ACC_INTS]
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

    for (int i = tid; i < 2 * C::ACC_INTS; i += 256) acc_lds[i] = 0;

    const int g_begin = p.bounds_ws[blockIdx.x], g_end = p.bounds_ws[blockIdx.x + 1];

    for (int ga = g_begin; ga < g_end; ga += kSub) {
        const int cnt = (g_end - ga) < kSub ? (g_end - ga) : kSub;
        __syncthreads();
        for (int i = tid; i < cnt; i += 256) {
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

        uint32_t meta_d[C::ITERS], meta_m[C::ITERS];  // row metadata for the batch being gathered / the one after
        f32x4 sa[C::ITERS], sb[C::ITERS];

        // M: metadata of one batch -> registers (0xFFFF = padding row)
        auto load_meta = [&](const BatchIt& it, uint32_t (&m)[C::ITERS]) {
            const uint16_t* rows = p.rows + (int64_t)(ga + it.gi) * maxr;
#pragma unroll
            for (int k = 0; k < C::ITERS; k++) {
                const int lr = (k * 256 + tid) / C::F4_PER_ROW;
                const int r = it.r0 + lr;
                m[k] = (valid(it) && r < it.n) ? (uint32_t)rows[r] : 0xFFFFu;
            }
        };
        // D: gathers of the batch's A_j and B_i rows -> registers
        auto load_data = [&](const BatchIt& it, const uint32_t (&m)[C::ITERS]) {
            const int64_t g = ga + it.gi;
            const int sb0 = valid(it) ? sbase[it.gi] : 0;
#pragma unroll
            for (int k = 0; k < C::ITERS; k++) {
                const int c4 = (k * 256 + tid) % C::F4_PER_ROW;
                sa[k] = f32x4{0.f, 0.f, 0.f, 0.f};
                sb[k] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (m[k] != 0xFFFFu && !(p.ablate & 2)) {
                    const int src = m[k] & 0xFF, d = m[k] >> 8, dl = d & 127;
                    const int64_t srow = (d & 0x80) ? (int64_t)(sb0 + src) : (g * p.n_dense + src);
                    sa[k] = *(const f32x4*)(p.A + srow * K + c4 * 4);
                    sb[k] = *(const f32x4*)(p.Bc + (g * nc + dl) * (int64_t)K + c4 * 4);
                }
            }
        };
        // W: h = relu(A_j - B_i) -> LDS tile, plus the destination byte of every row
        auto write_tile = [&](int buf, const uint32_t (&m)[C::ITERS]) {
            float* dst = hid + buf * C::HID_FLOATS;
#pragma unroll
            for (int k = 0; k < C::ITERS; k++) {
                const int q = k * 256 + tid;
                const int lr = q / C::F4_PER_ROW, c4 = q % C::F4_PER_ROW;
                const f32x4 t = sa[k] - sb[k];
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = fmaxf(t[e], 0.f);
                *(f32x4*)(dst + lr * C::LDH + c4 * 4) = v;
                if (c4 == 0) dstl[buf * C::TR + lr] = m[k] == 0xFFFFu ? (uint8_t)0xFF : (uint8_t)((m[k] >> 8) & 127);
            }
        };
        // flush one finished object's accumulator (feature columns; the [xyz | 0 x 5] tail of the rows is written
        // by the centroid-table kernel), re-zero
        auto flush = [&](int64_t g, int abuf) {
            int* a = acc_lds + abuf * C::ACC_INTS;
            float* o = p.out + g * nc * (int64_t)p.ldo;
            for (int i = tid; i < nc * N; i += 256) {
                const int c = i / N, col = i % N;
                o[c * (int64_t)p.ldo + col] = __int_as_float(a[i]);
                a[i] = 0;
            }
        };

        BatchIt it_c{0, 0, cnt > 0 ? (int)nr[0] : 0};
        BatchIt it_d = advance(it_c);
        BatchIt it_m = advance(it_d);
        // prologue: M(0), M(1), D(0), W(0)
        load_meta(it_c, meta_d);
        load_meta(it_d, meta_m);
        load_data(it_c, meta_d);
        write_tile(0, meta_d);
#pragma unroll
        for (int k = 0; k < C::ITERS; k++) meta_d[k] = meta_m[k];
        __syncthreads();

        int64_t flush_g = -1;
        int flush_buf = 0;
        for (int t = 0; valid(it_c); t++) {
            // the object finished in the previous batch drains to HBM underneath this batch's MFMAs
            if (flush_g >= 0) {
                if (!(p.ablate & 8)) flush(flush_g, flush_buf);
                flush_g = -1;
            }
            if (valid(it_d)) load_data(it_d, meta_d);      // D(t+1): gathers go out first ...
            if (!(p.ablate & 32)) load_meta(it_m, meta_m);  // M(t+2): ... the younger metadata loads stay in flight

            // C(t): MFMA block on tile t & 1
            f32x16 acc[RT][C::NTW];
#pragma unroll
            for (int rt = 0; rt < RT; rt++)
#pragma unroll
                for (int nt = 0; nt < C::NTW; nt++)
#pragma unroll
                    for (int e = 0; e < 16; e++) acc[rt][nt][e] = bias[nt];  // bias rides in the accumulator
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
            if (!(p.ablate & 4))
#pragma unroll
            for (int ch = 0; ch < NCH; ch++) {
                if (ch + 1 < NCH) {
#pragma unroll
                    for (int rt = 0; rt < RT; rt++)
#pragma unroll
                        for (int qi = 0; qi < QC; qi++)
                            a_nxt[rt][qi] = *(const f32x4*)(hrow + rt * 32 * C::LDH + ((ch + 1) * QC + qi) * 4);
                }
                // keep the next chunk's ds_read_b128s ABOVE this chunk's MFMAs (hipcc otherwise sinks each read to just
                // before its first use and exposes the LDS latency once per 8 MFMAs)
#if T2P_SCHED_BARRIER
                __builtin_amdgcn_sched_barrier(0);
#endif
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
            // segmented max into the object's accumulator (runs of equal destination folded in registers first)
            const int abuf = it_c.gi & 1;
            int* accb = acc_lds + abuf * C::ACC_INTS;
            const uint8_t* dl = dstl + buf * C::TR;
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                const int trow0 = (wm * RT + rt) * 32;
                if (it_c.r0 + trow0 >= it_c.n || (p.ablate & 1)) continue;
                int dq[16];
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const int b = dl[trow0 + 8 * (e >> 2) + 4 * h + (e & 3)];
                    dq[e] = b == 0xFF ? -1 : b;
                }
                // Segmented max.  Rows are sorted by destination, so after a forward running-max pass the last row of
                // every run holds the run's maximum and only those rows touch LDS.  No explicit ReLU: the accumulator is
                // a signed-integer max against +0, which a negative float (negative as an integer) never beats.
                bool same[16], is_end[16];
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    same[e] = e > 0 && dq[e] == dq[e - 1];
                    is_end[e] = dq[e] >= 0 && (e == 15 || dq[e] != dq[e + 1]);
                }
#pragma unroll
                for (int nt = 0; nt < C::NTW; nt++) {
                    int* col = accb + wn * C::NTW * 32 + nt * 32 + l31;
                    float v[16];
#pragma unroll
                    for (int e = 0; e < 16; e++) {
                        v[e] = acc[rt][nt][e];
                        if (e > 0) v[e] = same[e] ? fmaxf(v[e], v[e - 1]) : v[e];
                    }
#pragma unroll
                    for (int e = 0; e < 16; e++) {
#if T2P_ANY_SKIP
                        if (__any(is_end[e]))  // wave-uniform skip: most slots are run interiors in both lane halves
#endif
                        {
                            if (is_end[e]) atomicMax(col + dq[e] * N, __float_as_int(v[e]));
                        }
                    }
                }
            }
            if (it_c.r0 + C::TR >= it_c.n) {  // last batch of this object
                flush_g = ga + it_c.gi;
                flush_buf = abuf;
            }
            if (valid(it_d) && !(p.ablate & 16)) write_tile((t + 1) & 1, meta_d);  // W(t+1)
#pragma unroll
            for (int k = 0; k < C::ITERS; k++) meta_d[k] = meta_m[k];
            it_c = it_d;
            it_d = it_m;
            it_m = advance(it_m);
            if (!(p.ablate & 64)) __syncthreads();
        }
        if (flush_g >= 0) flush(flush_g, flush_buf);
    }
}

template <int K, int N, int WN, int RT>
int launch_sa_cfg(const SaParams& p_in, hipStream_t st, const char* name) {
    using C = SaCfg<K, N, WN, RT>;
    auto kern = k_ws_sa<K, N, WN, RT>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)C::lds_bytes());
        if (e != hipSuccess) {
            set_error("ws_sa: cannot reserve %zu B of LDS: %s", C::lds_bytes(), hipGetErrorString(e));
            return (int)e;
        }
        attr_set = true;
    }
    if (p_in.n_obj <= 0) return 0;
    T2P_CHECK_ARG(p_in.n_obj < (1 << 30) && p_in.n_obj * p_in.n_dense < 0x7fffffffLL, "ws_sa: chunk too large for 32-bit rows");
    int n_wg = num_cus();
    if (n_wg > p_in.n_obj) n_wg = (int)p_in.n_obj;
    SaParams p = p_in;
    {
        const char* ab = getenv("T2P_ABLATE");  // debug: 1 = no epilogue, 2 = no gathers, 4 = no MFMA
        p.ablate = ab ? atoi(ab) : 0;
    }
    {
        ProfScope ps_("sa_balance", st);
        hipLaunchKernelGGL(k_balance, dim3(1), dim3(1024), 0, st, p.n_rows, (int)p.n_obj, C::TR, n_wg, p.prefix_ws,
                           p.bounds_ws);
    }
    T2P_CHECK_LAUNCH("sa_balance");
    ProfScope ps_(name, st);
    hipLaunchKernelGGL(kern, dim3(n_wg), dim3(256), C::lds_bytes(), st, p);
    T2P_CHECK_LAUNCH("ws_sa");
    return 0;
}

}  // namespace

int launch_ws_sa(int H, int Cout, const SaParams& p, hipStream_t st) {
    T2P_CHECK_ARG((((uintptr_t)p.A | (uintptr_t)p.Bc) & 15) == 0, "ws_sa: tables must be 16-byte aligned");
    if (H == 32 && Cout == 64) return launch_sa_cfg<32, 64, 2, 4>(p, st, "ws_edge_sa_k32_n64");
    if (H == 128 && Cout == 128) return launch_sa_cfg<128, 128, 4, 2>(p, st, "ws_edge_sa_k128_n128");
    if (H == 256 && Cout == 256) return launch_sa_cfg<256, 256, 4, 1>(p, st, "ws_edge_sa_k256_n256");
    set_error("ws_sa: no instantiation for H=%d C=%d", H, Cout);
    return T2P_E_UNSUPPORTED;
}

}  // namespace t2p

// Set-abstraction edge kernel, f16x3 path, INDEPENDENT waves over centroid groups (SA level 1: H = 32, C = 64).
// (reference: gnn.PointConv(local_nn)(x, (pos, pos[idx]), edge_index), models/pointcloud/pointnet2.py:31-35).
//
// SA level 1 multiplies [32 x 64] weights: 12 MFMAs per 32-row tile.  The column-slice kernel (ws_sa2.hip) spends ~330
// instructions per 6 MFMAs there (shared row batches, staging through LDS planes, a workgroup barrier per batch) and is bound by
// instruction issue.  Here nothing is shared between waves:
//   * the whole weight matrix sits in 32 registers of EVERY wave (sa_rows.hip's row-owning form), 12 waves per CU;
//   * the unit of work is (object, group of GS = 16 centroids).  The row list is sorted by centroid, so a group is one
//     contiguous piece of it (its ends are found by a two-round 64-lane search in the list itself: no offset tables from the
//     producers), and its max-accumulator [16][64] and centroid table [16][32] are PRIVATE to the wave: no atomics between
//     waves, no object barrier, no drain phase - the wave that finishes a group writes its 16 output rows;
//   * rows travel by LDS-DMA (8 rows x 128 B per instruction, XOR-swizzled on the source address) into a 4 KB tile buffer of
//     the wave; the buffer is read into registers at once, the next tile's DMA goes out behind the reads and lands under this
//     tile's arithmetic and under the other waves of the SIMD (three per SIMD hide what a lone wave would wait for);
//   * ReLU(A_j - B_i), the fp16 hi / lo split and the three MFMAs per 16 k are sa_rows.hip's; float max into the private LDS
//     accumulator, bias + ReLU at the group's end.
// Results agree with ws_sa2.hip to fp32 rounding (another k grouping inside the MFMAs, bias added last).
#include "t2p_common.h"

namespace t2p {
int launch_sa_balance(const SaParams& p, int tile_rows, int n_wg, hipStream_t st);  // ws_sa.hip

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gl_void;

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)

template <int K, int N, int NC, int GS, int NW>
struct GrpCfg {
    static constexpr int NT = 64 * NW;
    static constexpr int ND = 2 * NC;
    static constexpr int S16 = K / 16, NTW = N / 32;
    static constexpr int NG = NC / GS;                   // groups per object
    static constexpr int MAXR = NC * 33;
    static constexpr int TILE_BYTES = 32 * K * 4;
    static constexpr int BT_STRIDE = K * 4 + 16;         // centroid-table row pitch: +16 B keeps ds_read_b128 conflict-free
    static constexpr int LPR = 64 / GS;                  // lanes per centroid row in the table build / the drain
    static constexpr int ACC_BYTES = GS * N * 4;
    static constexpr int BT_BYTES = GS * BT_STRIDE;
    static constexpr int WAVE_BYTES = (ACC_BYTES + BT_BYTES + 64 + TILE_BYTES + 15) / 16 * 16;
    static constexpr size_t lds_bytes() { return (size_t)NW * WAVE_BYTES + 16; }   // + the object counter
    static_assert(K == 32, "one 128-byte piece per row: the tile buffer is read into registers at once");
    static_assert(N % 32 == 0 && NC % GS == 0 && 64 % GS == 0 && (K / LPR) % 4 == 0 && (N / LPR) % 4 == 0, "shape");
};

template <int SEL>
__device__ __forceinline__ float sub_half_g(float v, fp16x2 h) {   // v - (float)h[SEL] in one VALU op (exact)
    float r;
    if constexpr (SEL == 0)
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(v));
    else
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(v));
    return r;
}

template <int K, int N, int NC, int GS, int NW>
__global__ __launch_bounds__(64 * NW, (NW + 3) / 4) void k_sa_groups(SaParams p) {
    using C = GrpCfg<K, N, NC, GS, NW>;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, rr = lane & 31;
    // this wave's LDS: accumulator [GS][N] | centroid table [GS][K (+pad)] | row offsets [32] u16 | tile buffer
    const uint32_t acc_off = (uint32_t)(wave * C::WAVE_BYTES);
    const uint32_t bt_off = acc_off + C::ACC_BYTES;
    const uint32_t dst_off = bt_off + C::BT_BYTES;
    const uint32_t tile_off = dst_off + 64;

    // ---- stationary weights (natural k order: this lane's operand of step s covers k = 16 s + 8 h + e) -------------------
    half8 w_hi[C::NTW][C::S16], w_lo[C::NTW][C::S16];
    {
        const uint4* wp = (const uint4*)p.W_x3;
        constexpr int PLANE_U4 = (N / 32) * C::S16 * 64;
#pragma unroll
        for (int nt = 0; nt < C::NTW; nt++)
#pragma unroll
            for (int s = 0; s < C::S16; s++) {
                const int kb = 16 * s + 8 * h;
                const int half_ = kb / (K / 2), step_ = (kb % (K / 2)) / 8;
                const int idx = (((nt * C::S16 + step_) * 2 + half_) * 32) + rr;
                w_hi[nt][s] = __builtin_bit_cast(half8, wp[idx]);
                w_lo[nt][s] = __builtin_bit_cast(half8, wp[PLANE_U4 + idx]);
            }
    }
    constexpr f32x16 kZero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // ---- per-lane constants ---------------------------------------------------------------------------------------------
    // tile buffer: row r at r * 128, its 16-byte chunk c at position c ^ ((r >> 1) & 7); this lane reads chunks 4 s + 2 h + j
    uint32_t rd[C::S16][2];
#pragma unroll
    for (int s = 0; s < C::S16; s++)
#pragma unroll
        for (int j = 0; j < 2; j++)
            rd[s][j] = tile_off + (uint32_t)(rr * 128 + (((4 * s + 2 * h + j) ^ ((rr >> 1) & 7)) * 16));
    // DMA instruction q: lane i fetches row 8 q + (i >> 3), LDS position i & 7 <- source chunk (i & 7) ^ swizzle(row)
    uint32_t dma_sel[4], dma_chunk[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int r = 8 * q + (lane >> 3);
        dma_sel[q] = (uint32_t)(r * 4);
        dma_chunk[q] = (uint32_t)((((lane & 7) ^ ((r >> 1) & 7))) * 16);
    }
    // table build / drain: lane = (centroid row cr of the group, column slice cs)
    const int cr = lane / C::LPR, cs = lane % C::LPR;
    constexpr int KPL = K / C::LPR, NPL = N / C::LPR;      // table / output columns per lane
    float bias_l[NPL];
#pragma unroll
    for (int e = 0; e < NPL; e++) bias_l[e] = p.bias[cs * NPL + e];

    int gtop = 0;         // fp16-range guard: this lane's maximum (bit pattern, before out_scale) of the drained outputs; reduced over
                          // the wave once, at the end (six ds_bpermute round trips per drain otherwise)
    const int g_begin = p.bounds_ws[blockIdx.x], g_end = p.bounds_ws[blockIdx.x + 1];
    // objects are handed out dynamically (an LDS counter): a wave that draws a light object comes back sooner
    int* ctr = (int*)(lds + NW * C::WAVE_BYTES);
    if (tid == 0) *ctr = 0;
    __syncthreads();

    for (;;) {
        int gi = 0;
        if (lane == 0) gi = atomicAdd(ctr, 1);
        const int g = g_begin + __builtin_amdgcn_readfirstlane(gi);
        if (g >= g_end) break;
        const int n = __builtin_amdgcn_readfirstlane((int)p.n_rows[g]);
        const uint16_t* list = p.rows + (int64_t)g * C::MAXR;
        const int first = __builtin_amdgcn_readfirstlane(p.first[g]);
        const uint32_t sb0 = (uint32_t)(first * C::ND + (g - first) * NC);

        // ---- group bounds: lane j of `bnd` = first list entry whose centroid is >= j GS (j = 0 .. NG; sorted list) -----------
        // two rounds of 64-lane probing for all NG - 1 inner boundaries at once: every lane samples the list at stride
        // ceil(n / 64), a ballot per boundary brackets it, a second pair of probes per lane and boundary pins it down
        int bnd = lane == 0 ? 0 : n;
        if (n > 0) {
            const int s1 = (n + 63) >> 6;               // <= 66
            const int i1 = lane * s1;
            int key1 = 255;
            if (i1 < n) key1 = (int)((list[i1] >> 8) & 127);
            int base[C::NG], k0[C::NG], k1[C::NG];
#pragma unroll
            for (int j = 1; j < C::NG; j++) {
                const unsigned long long b1 = __ballot(key1 >= j * GS);
                const int l1 = b1 ? (int)__builtin_ctzll(b1) : 64;   // first probe at or behind the boundary (64: none)
                base[j] = l1 == 0 ? -1 : (l1 - 1) * s1 + 1;           // the boundary lies in [base, base + s1 - 1]; -1: it is 0
                const int i2 = base[j] + 2 * lane;
                k0[j] = k1[j] = 255;
                if (base[j] >= 0 && i2 < n) k0[j] = (int)((list[i2] >> 8) & 127);
                if (base[j] >= 0 && i2 + 1 < n) k1[j] = (int)((list[i2 + 1] >> 8) & 127);
            }
#pragma unroll
            for (int j = 1; j < C::NG; j++) {
                int r = 0;
                if (base[j] >= 0) {
                    const bool g0 = k0[j] >= j * GS, g1 = k1[j] >= j * GS;
                    const unsigned long long b2 = __ballot(g0 || g1);   // (never empty: entries past n count as >= c)
                    const int cand = g0 ? base[j] + 2 * lane : base[j] + 2 * lane + 1;
                    r = __builtin_amdgcn_readlane(cand, (int)__builtin_ctzll(b2));
                    r = r < n ? r : n;
                }
                bnd = lane == j ? r : bnd;
            }
        }
        auto bound = [&](int j) { return __builtin_amdgcn_readlane(bnd, j); };

        // ---- the object's tile stream: (group, first row) in list order, empty groups skipped ------------------------------------
        auto first_tile_from = [&](int j, int& jo, int& ro) {
            while (j < C::NG && bound(j) >= bound(j + 1)) j++;
            jo = j;
            ro = j < C::NG ? bound(j) : 0;
        };
        auto next_tile = [&](int j, int r0, int& jo, int& ro) {
            if (r0 + 32 < bound(j + 1)) { jo = j; ro = r0 + 32; }
            else first_tile_from(j + 1, jo, ro);
        };
        auto tile_meta = [&](int j, int r0) -> uint32_t {   // entry min(r0 + rr, hi - 1): rows past the end repeat the last row
            const int hi = bound(j + 1);
            int idx = r0 + rr;
            idx = idx < hi ? idx : hi - 1;
            return (uint32_t)list[idx];
        };
        auto issue_tile = [&](uint32_t m) {
            const uint32_t src = m & 0xFFu, d = m >> 8;
            const uint32_t srow = (d & 0x80u) ? (sb0 + src) : ((uint32_t)g * (uint32_t)C::ND + src);
            const uint32_t rowbyte = srow * (uint32_t)(K * 4);
            uint32_t voff[4];
#pragma unroll
            for (int q = 0; q < 4; q++)
                voff[q] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)dma_sel[q], (int)rowbyte) + dma_chunk[q];
#pragma unroll
            for (int q = 0; q < 4; q++)
                __builtin_amdgcn_global_load_lds((gl_void*)((const char*)p.A + voff[q]), (lds_void*)(lds + tile_off + q * 1024), 16, 0, 0);
        };
        // positions of a group's centroids (this lane: row cr), fetched one group ahead
        const float* pos0 = p.out + ((int64_t)g * NC + cr) * (int64_t)p.ldo + N;
        float px = pos0[0], py = pos0[1], pz = pos0[2];

        // tile T+1 ("nx": its DMA is in flight while T is multiplied) and T+2 ("n2": its list entry is being fetched)
        int j_nx, r_nx, j_n2 = C::NG, r_n2 = 0;
        uint32_t m_nx = 0, m_n2 = 0;
        first_tile_from(0, j_nx, r_nx);
        if (j_nx < C::NG) {
            m_nx = tile_meta(j_nx, r_nx);
            issue_tile(m_nx);
            next_tile(j_nx, r_nx, j_n2, r_n2);
            if (j_n2 < C::NG) m_n2 = tile_meta(j_n2, r_n2);
        }

        for (int j = 0; j < C::NG; j++) {
            const int c0 = j * GS, lo = bound(j), hi = bound(j + 1);
            // ---- centroid table of the group B_i = W1p pos_i, accumulator at -inf -----------------------------------------------
            {
#pragma unroll
                for (int q4 = 0; q4 < KPL / 4; q4++) {
                    const int k0_ = cs * KPL + q4 * 4;
                    const f32x4 w0 = *(const f32x4*)(p.wp + k0_), w1 = *(const f32x4*)(p.wp + K + k0_),
                                w2 = *(const f32x4*)(p.wp + 2 * K + k0_);
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; e++) {   // same order as k_sample_group's table: ((x w0) + y w1) + z w2
                        float a = px * w0[e];
                        a = fmaf(py, w1[e], a);
                        a = fmaf(pz, w2[e], a);
                        v[e] = a;
                    }
                    *(f32x4*)(lds + bt_off + cr * C::BT_STRIDE + k0_ * 4) = v;
                }
                typedef int i32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
                for (int q4 = 0; q4 < NPL / 4; q4++)
                    *(i32x4*)(lds + acc_off + (cr * N + cs * NPL + q4 * 4) * 4) =
                        i32x4{(int)0xFF800000, (int)0xFF800000, (int)0xFF800000, (int)0xFF800000};
                if (j + 1 < C::NG) {
                    const float* pn = pos0 + (int64_t)(c0 + GS) * (int64_t)p.ldo;
                    px = pn[0], py = pn[1], pz = pn[2];
                }
            }
            for (int r0 = lo; r0 < hi; r0 += 32) {
                const uint32_t m = m_nx;       // (this tile is "nx" of the previous iteration)
                // the tile has landed: all of it goes to registers, then the buffer is free for the next tile's DMA.
                // (LDS accesses between a DMA and its wait are inline asm: hipcc would put s_waitcnt vmcnt(0) in front of an
                // ordinary one, i.e. wait for the DMA it cannot tell apart from the access)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                f32x4 x[C::S16][2];
#pragma unroll
                for (int s = 0; s < C::S16; s++)
#pragma unroll
                    for (int jj = 0; jj < 2; jj++) asm volatile("ds_read_b128 %0, %1" : "=v"(x[s][jj]) : "v"(rd[s][jj]) : "memory");
                const uint32_t dl = ((m >> 8) & 127u) - (uint32_t)c0;
                const uint32_t brow = bt_off + dl * (uint32_t)C::BT_STRIDE + (uint32_t)(h * 32);
                f32x4 b[C::S16][2];
#pragma unroll
                for (int s = 0; s < C::S16; s++) {
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b[s][0]) : "v"(brow), "n"(s * 64) : "memory");
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b[s][1]) : "v"(brow), "n"(s * 64 + 16) : "memory");
                }
                asm volatile("ds_write_b16 %0, %1" ::"v"(dst_off + (uint32_t)(rr * 2)), "v"(dl * (uint32_t)(N * 4)) : "memory");
                uint2 four[4];   // accumulator-row byte offsets of this lane's 16 result rows 8 q + 4 h + {0..3}
                {
                    const uint32_t a4 = dst_off + (uint32_t)(h * 8);
                    asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:16\n\tds_read_b64 %2, %4 offset:32\n\t"
                                 "ds_read_b64 %3, %4 offset:48\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(four[0]), "=&v"(four[1]), "=&v"(four[2]), "=&v"(four[3]) : "v"(a4) : "memory");
                }
                // (the wait above also covers the tile and table reads: the asm below ties their registers to it)
#pragma unroll
                for (int s = 0; s < C::S16; s++)
                    asm volatile("" : "+v"(x[s][0]), "+v"(x[s][1]), "+v"(b[s][0]), "+v"(b[s][1]));
                // next tile: DMA out; the tile after it: list entry requested (both land under this tile's arithmetic)
                j_nx = j_n2;
                r_nx = r_n2;
                m_nx = m_n2;
                if (j_nx < C::NG) {
                    issue_tile(m_nx);
                    next_tile(j_nx, r_nx, j_n2, r_n2);
                    if (j_n2 < C::NG) m_n2 = tile_meta(j_n2, r_n2);
                }

                f32x16 acc[C::NTW];
#pragma unroll
                for (int s = 0; s < C::S16; s++) {
                    uint32_t nh[4], nl[4];
#pragma unroll
                    for (int pr = 0; pr < 4; pr++) {
                        const int jj = pr >> 1, e0 = (pr & 1) * 2;
                        const float v0 = fmaxf(x[s][jj][e0] - b[s][jj][e0], 0.f), v1 = fmaxf(x[s][jj][e0 + 1] - b[s][jj][e0 + 1], 0.f);
                        const fp16x2 hh = __builtin_amdgcn_cvt_pkrtz(v0, v1);
                        const fp16x2 ll = __builtin_amdgcn_cvt_pkrtz(sub_half_g<0>(v0, hh), sub_half_g<1>(v1, hh));
                        nh[pr] = __builtin_bit_cast(uint32_t, hh);
                        nl[pr] = __builtin_bit_cast(uint32_t, ll);
                    }
                    const half8 a_hi = __builtin_bit_cast(half8, u32x4{nh[0], nh[1], nh[2], nh[3]});
                    const half8 a_lo = __builtin_bit_cast(half8, u32x4{nl[0], nl[1], nl[2], nl[3]});
#pragma unroll
                    for (int nt = 0; nt < C::NTW; nt++) acc[nt] = MFMA16(a_hi, w_hi[nt][s], s == 0 ? kZero16 : acc[nt]);
#pragma unroll
                    for (int nt = 0; nt < C::NTW; nt++) acc[nt] = MFMA16(a_hi, w_lo[nt][s], acc[nt]);
#pragma unroll
                    for (int nt = 0; nt < C::NTW; nt++) acc[nt] = MFMA16(a_lo, w_hi[nt][s], acc[nt]);
                }
                // float max into the wave's accumulator (nobody else touches it)
                {
                    uint32_t ad[16];
#pragma unroll
                    for (int e = 0; e < 16; e++) {
                        const uint32_t pair = (e & 2) ? four[e >> 2].y : four[e >> 2].x;
                        ad[e] = acc_off + (uint32_t)(rr * 4) + ((e & 1) ? (pair >> 16) : (pair & 0xFFFFu));
                    }
                    // MFMA -> LDS-data hazard: hipcc does not see that the asm below reads MFMA results, and it may move MFMAs
                    // (no memory operation) across a plain asm fence - so the results themselves pass THROUGH the s_nop
                    static_assert(C::NTW == 2, "the hazard fence below names both result blocks");
                    asm volatile("s_nop 15" : "+v"(acc[0]), "+v"(acc[1])::"memory");
#pragma unroll
                    for (int nt = 0; nt < C::NTW; nt++)
#pragma unroll
                        for (int e = 0; e < 16; e++)
                            asm volatile("ds_max_f32 %0, %1 offset:%2" ::"v"(ad[e]), "v"(acc[nt][e]), "n"(nt * 128) : "memory");
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // ---- drain: relu(max + bias) of the group's GS centroids ---------------------------------------------------------
            {
                float* o = p.out + ((int64_t)g * NC + c0 + cr) * (int64_t)p.ldo + cs * NPL;
                int top = gtop;
#pragma unroll
                for (int q4 = 0; q4 < NPL / 4; q4++) {
                    const f32x4 raw = *(const f32x4*)(lds + acc_off + (cr * N + cs * NPL + q4 * 4) * 4);
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const float r = fmaxf(raw[e] + bias_l[q4 * 4 + e], 0.f);   // (a centroid without rows stays at -inf: 0)
                        const int bits = __float_as_int(r);
                        top = bits > top ? bits : top;
                        v[e] = r * p.out_scale;
                    }
                    *(f32x4*)(o + q4 * 4) = v;
                }
                gtop = top;
            }
        }
    }
    uint32_t gbits = 0;
    guard_track_bits(gbits, gtop);
    if (p.amax_out != nullptr && lane == 0 && gbits != 0u)
        atomicMax(p.amax_out, __float_as_uint(__uint_as_float(gbits) * p.out_scale));
}

#ifndef T2P_GRP_WAVES
#define T2P_GRP_WAVES 12
#endif
constexpr int kGrpWaves = T2P_GRP_WAVES;

}  // namespace

bool sa_groups_selected(int H, int Cout, const SaParams& p) {
    return H == 32 && Cout == 64 && p.W_x3 != nullptr && p.wp != nullptr && !(p.plan & 2);
}

// (tile rows, workgroups) for the range balancing: one 12-wave workgroup per CU, cost = rows
int sa_groups_launch_shape(int64_t n_obj, int* tile_rows, int* n_wg) {
    int n = num_cus();
    if (n > 1024) n = 1024;
    if (n > n_obj) n = (int)n_obj;
    *tile_rows = 32;
    *n_wg = n;
    return 0;
}

int launch_sa_groups(int H, int Cout, const SaParams& p, hipStream_t st) {
    if (!(H == 32 && Cout == 64 && p.n_cent == 128 && p.n_dense == 256 && p.wp && p.W_x3)) {
        set_error("sa_groups: built for SA level 1 (H = 32, C = 64, 128 centroids of 256 points, f16x3)");
        return T2P_E_UNSUPPORTED;
    }
    using C = GrpCfg<32, 64, 128, 16, kGrpWaves>;
    auto kern = k_sa_groups<32, 64, 128, 16, kGrpWaves>;
    T2P_TRY(reserve_lds((const void*)kern, C::lds_bytes(), "sa_groups"));
    if (p.n_obj <= 0) return 0;
    T2P_CHECK_ARG(p.n_obj < (1 << 30) && p.n_obj * p.n_dense * (int64_t)H * 4 < 0xffffffffLL,
                  "sa_groups: chunk too large for 32-bit table offsets");
    T2P_CHECK_ARG((((uintptr_t)p.A | (uintptr_t)p.out | (uintptr_t)p.W_x3 | (uintptr_t)p.wp) & 15) == 0 && p.ldo % 4 == 0,
                  "sa_groups: tables must be 16-byte aligned");
    int tr, n_wg;
    sa_groups_launch_shape(p.n_obj, &tr, &n_wg);
    if (!p.balanced) T2P_TRY(launch_sa_balance(p, tr, n_wg, st));
    ProfScope ps_("ws_edge_sa_k32_n64", st);
    hipLaunchKernelGGL(kern, dim3(n_wg), dim3(C::NT), C::lds_bytes(), st, p);
    T2P_CHECK_LAUNCH("sa_groups");
    return 0;
}

}  // namespace t2p

// Set-abstraction edge kernel of SA level 1 (6 -> 32 -> 64), f16x3 path: BOTH layers per edge, the object's points in LDS.
// (reference: gnn.PointConv(local_nn)(x, (pos, pos[idx]), edge_index), models/pointcloud/pointnet2.py:31-35).
//
// Levels 2 and 3 split layer 1 algebraically (a table row per POINT, gathered per edge) because their layer 1 costs as much as
// layer 2.  At level 1 it is 6 inputs wide: 3 small MFMAs against layer 2's 12, while the gathered table row is 128 B per edge -
// sa_groups.hip (the gather form of this kernel) ran into the ~5 TB/s the row-gather path delivers, 33 GB per step.  Here
// nothing is gathered: a wave stages its object's 256 points ([rgb x | y z 1 0], 32 B each) in LDS once and forms every edge's
// layer-1 input [rgb_j | pos_j - pos_i | 1] from two 16-byte LDS reads, exactly the reference's order of operations:
//   * layer 1 runs TRANSPOSED on v_mfma_f32_32x32x8_f16: D[hidden][row] = W1^T[hidden][k] X^T[k][row], k = (r g b dx | dy dz 1 0),
//     the bias rides on the constant-1 input; three products (hi hi, lo hi, hi lo) as everywhere in the f16x3 path;
//   * its result registers are already a layer-2 A operand: lane (row, half) holds hidden 8 q + 4 half + {0..3}, q = 0..3 - the
//     layer-2 weight image is re-ordered once per wave to that k order (two 16-byte pieces of the natural-order image per
//     operand), so ReLU + fp16 split feed the 12 layer-2 MFMAs without touching LDS;
//   * self-loop rows take their point from ANOTHER object's rows (PyG's index aliasing: row sbase + c of the flattened batch):
//     16 such points per centroid group, fetched one group ahead into a 512-byte side table;
//   * 12 independent waves per CU, objects handed out through an LDS counter, no barrier, no atomics between waves.  The row list
//     (sorted by centroid) streams through a 256-entry LDS ring; a wave's max-accumulator is a WINDOW of 16 centroid slots (slot =
//     centroid & 15) in two halves: a tile takes the next 32 rows as long as they stay within the half of its first row and the
//     next one (31.6 live rows per tile), a half is drained - relu(max + bias) -> its 8 output rows - once the list has passed it.
// k_sample_group no longer writes the A_1 table (6.3 GB per step) and no B_1 table exists at all.
// T2P_PABL (development only, results are wrong): 1 = no atomics, 2 = no layer-2 MFMAs, 4 = no layer-1 MFMAs, 8 = no point reads
#ifndef T2P_PABL
#define T2P_PABL 0
#endif
// T2P_PPROF = w + 1: wave w of block 0 sums s_memtime differences of its phases over the launch (t2p_debug_pprof reads them)
#ifndef T2P_PPROF
#define T2P_PPROF 0
#endif
#include "t2p_common.h"

namespace t2p {
int launch_sa_balance(const SaParams& p, int tile_rows, int n_wg, hipStream_t st);  // ws_sa.hip

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define MFMA8(a, b, c) __builtin_amdgcn_mfma_f32_32x32x8f16(a, b, c, 0, 0, 0)

template <int NW>
struct PtsCfg {
    static constexpr int K = 32, N = 64, NC = 128, ND = 256, GS = 16;
    static constexpr int NT = 64 * NW;
    static constexpr int S16 = K / 16, NTW = N / 32;
    static constexpr int HS = GS / 2, NH = NC / HS;      // the accumulator window's halves: 8 centroids, 16 halves per object
    static constexpr int MAXR = NC * 33;
    static constexpr int LPR = 64 / HS;                  // lanes per centroid row in the drain (of a half)
    static constexpr int ACC_BYTES = GS * N * 4;
    static constexpr int PTS_BYTES = ND * 24;            // [r g b x y z] per point
    static constexpr int SELF_BYTES = GS * 24;           // the points the group's self-loop rows name (same record)
    static constexpr int CEN_BYTES = GS * 32;            // the group's centroids [0 0 0 x | y z 0 0]
    static constexpr int RING_BYTES = 4 * 64 * 2;        // four 64-entry windows of the row list
    static constexpr int WAVE_BYTES = ACC_BYTES + PTS_BYTES + SELF_BYTES + CEN_BYTES + 16 + 64 + RING_BYTES;
    static_assert(WAVE_BYTES % 16 == 0, "16-byte accesses");
    static constexpr size_t lds_bytes() { return (size_t)NW * WAVE_BYTES + 16 + N * 4; }   // + the object counter, the bias
};

template <int SEL>
__device__ __forceinline__ float sub_half_p(float v, fp16x2 h) {   // v - (float)h[SEL] in one VALU op (exact)
    float r;
    if constexpr (SEL == 0)
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(v));
    else
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(v));
    return r;
}
// max(v, 0) in one instruction: fmaxf() puts a canonicalising v_max v, v in front of the maximum (the operand is an MFMA result
// of unknown NaN class).  As SIGNED INTEGERS the bit patterns of the non-negative floats order like the floats and every
// negative float (and -0) is below 0: one v_max_i32.  (Not inline asm: hipcc places the MFMA -> VALU wait states only in front
// of instructions it knows.)
__device__ __forceinline__ float relu1(float v) {
    const int b = __float_as_int(v);
    return __int_as_float(b > 0 ? b : 0);
}
// (v0, v1) -> fp16 pair hi (round to nearest) and the fp16 pair of the exact residuals
__device__ __forceinline__ void split2(float v0, float v1, uint32_t& hi, uint32_t& lo) {
    const fp16x2 hh = cvt_pk_f16(v0, v1);
    const fp16x2 ll = cvt_pk_f16(sub_half_p<0>(v0, hh), sub_half_p<1>(v1, hh));
    hi = __builtin_bit_cast(uint32_t, hh);
    lo = __builtin_bit_cast(uint32_t, ll);
}

#if T2P_PPROF
__device__ unsigned long long t2p_pprof_sums[16];
#define PPROF_DECL unsigned long long pp_t = __builtin_amdgcn_s_memtime(), pp_sum[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; const unsigned long long pp_begin = pp_t
#define PPROF_MARK(i)                                                  \
    do {                                                               \
        const unsigned long long n_ = __builtin_amdgcn_s_memtime();    \
        pp_sum[i] += n_ - pp_t;                                        \
        pp_t = n_;                                                     \
    } while (0)
#define PPROF_COUNT(i) pp_sum[i] += 1
#else
#define PPROF_DECL
#define PPROF_MARK(i)
#define PPROF_COUNT(i)
#endif

// PARTS = 1: a wave owns whole objects.  PARTS = 4 (calls of a few hundred objects - the reference's 64-cell batches -, where a
// wave per object leaves most of the chip idle and the call waits for one object's ~130 tiles): a wave owns a QUARTER of an
// object's centroids = four halves of the window below; it finds its stretch of the (centroid-sorted) row list by a wave-wide
// search.  Every row goes through the same arithmetic either way and the maximum does not care who owns it: same bits.
template <int NW, int PARTS>
__global__ __launch_bounds__(64 * NW, (NW + 3) / 4) void k_sa_points(SaParams p) {
    using C = PtsCfg<NW>;
    static_assert(PARTS == 1 || (C::NH % PARTS == 0 && C::NH / PARTS >= 3), "a part is a whole number (>= 3: two open + one fetched ahead) of halves");
    constexpr int K = C::K, N = C::N, NC = C::NC;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, rr = lane & 31;
    // this wave's LDS: accumulator [GS][N] | points [256][6] | self-loop points [GS][6] | centroids [GS][8] | (1, 0) | row offsets
    // [32] u16 | list ring [256] u16
    const uint32_t acc_off = (uint32_t)(wave * C::WAVE_BYTES);
    const uint32_t pts_off = acc_off + C::ACC_BYTES;
    const uint32_t self_off = pts_off + C::PTS_BYTES;
    const uint32_t cen_off = self_off + C::SELF_BYTES;
    const uint32_t one_off = cen_off + C::CEN_BYTES;
    const uint32_t dst_off = one_off + 16;
    const uint32_t ring_off = dst_off + 64;
    if (lane == 0) *(f32x4*)(lds + one_off) = f32x4{1.f, 0.f, 0.f, 0.f};

    // ---- layer-2 weights, re-ordered: operand element e of step s in lane (n, half) = hidden 16 s + 8 (e >> 2) + 4 half + (e & 3)
    // = elements 4 half .. 4 half + 3 of the natural-order pieces (k = 16 s + 8 h' + e') h' = 0 (e < 4) and h' = 1 (e >= 4)
    half8 w_hi[C::NTW][C::S16], w_lo[C::NTW][C::S16];
    {
        const uint4* wp = (const uint4*)p.W_x3;
        constexpr int PLANE_U4 = (N / 32) * C::S16 * 64;
#pragma unroll
        for (int nt = 0; nt < C::NTW; nt++)
#pragma unroll
            for (int s = 0; s < C::S16; s++) {
                uint4 pc[2][2];   // [plane][h']
#pragma unroll
                for (int hp = 0; hp < 2; hp++) {
                    const int kb = 16 * s + 8 * hp;
                    const int half_ = kb / (K / 2), step_ = (kb % (K / 2)) / 8;
                    const int idx = (((nt * C::S16 + step_) * 2 + half_) * 32) + rr;
                    pc[0][hp] = wp[idx];
                    pc[1][hp] = wp[PLANE_U4 + idx];
                }
                const u32x4 vh = h ? u32x4{pc[0][0].z, pc[0][0].w, pc[0][1].z, pc[0][1].w} : u32x4{pc[0][0].x, pc[0][0].y, pc[0][1].x, pc[0][1].y};
                const u32x4 vl = h ? u32x4{pc[1][0].z, pc[1][0].w, pc[1][1].z, pc[1][1].w} : u32x4{pc[1][0].x, pc[1][0].y, pc[1][1].x, pc[1][1].y};
                w_hi[nt][s] = __builtin_bit_cast(half8, vh);
                w_lo[nt][s] = __builtin_bit_cast(half8, vl);
            }
    }
    // ---- layer-1 weights as the A operand of the transposed product: lane (hidden m, half): k = 4 half + e over
    // (W1[r] W1[g] W1[b] W1[dx] | W1[dy] W1[dz] b1 0)[m]
    half4 w1_hi, w1_lo;
    {
        float wv[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int k = 4 * h + e;
            wv[e] = k < 6 ? p.w1[k * K + rr] : (k == 6 ? p.b1[rr] : 0.f);
        }
        uint32_t hi0, lo0, hi1, lo1;
        split2(wv[0], wv[1], hi0, lo0);
        split2(wv[2], wv[3], hi1, lo1);
        w1_hi = __builtin_bit_cast(half4, u32x2{hi0, hi1});
        w1_lo = __builtin_bit_cast(half4, u32x2{lo0, lo1});
    }
    constexpr f32x16 kZero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // drain of a half: lane = (centroid row cr of the half, column slice cs)
    const int cr = lane / C::LPR, cs = lane % C::LPR;
    constexpr int NPL = N / C::LPR;
    const uint32_t bias_off = (uint32_t)(NW * C::WAVE_BYTES + 16 + cs * NPL * 4);   // (this lane's NPL bias values, shared table)
    if (tid < N) *(float*)(lds + NW * C::WAVE_BYTES + 16 + tid * 4) = p.bias[tid];

    int gtop = 0;         // fp16-range guard: this lane's maximum (bit pattern, before out_scale) of the drained outputs; reduced
                          // over the wave once, at the end
    PPROF_DECL;
    const int g_begin = p.bounds_ws[blockIdx.x], g_end = p.bounds_ws[blockIdx.x + 1];
    int* ctr = (int*)(lds + NW * C::WAVE_BYTES);
    if (tid == 0) *ctr = 0;
    __syncthreads();

    f32x4 pout[NPL / 4];      // drained rows of the last closed group, not stored yet
    float* pptr = nullptr;
    bool has_pout = false;
    for (;;) {
        int gi = 0;
        if (lane == 0) gi = atomicAdd(ctr, 1);
        gi = __builtin_amdgcn_readfirstlane(gi);
        const int g = g_begin + (PARTS > 1 ? gi / PARTS : gi);
        if (g >= g_end) break;
        constexpr int HP = C::NH / PARTS;                        // halves per part
        const int hs = PARTS > 1 ? (gi % PARTS) * HP : 0;        // this wave's halves: [hs, hend)
        const int hend = hs + HP;
        PPROF_MARK(0);      // drawing an object
        PPROF_COUNT(9);
        int n = __builtin_amdgcn_readfirstlane((int)p.n_rows[g]);
        const uint16_t* list = p.rows + (int64_t)g * C::MAXR;
        if constexpr (PARTS > 1) {
            // first list index whose centroid is >= target (n if none): 64 probes `step` apart, then the stretch between two probes
            auto lower = [&](int target) -> int {
                if (target <= 0 || n <= 0) return 0;
                if (target >= NC) return n;
                const int step = (n + 63) >> 6;
                const int i0 = lane * step;
                const uint32_t c0 = i0 < n ? (((uint32_t)list[i0] >> 8) & 127u) : 255u;
                const unsigned long long ge = __ballot(c0 >= (uint32_t)target);
                const int fl = ge ? (int)__builtin_ctzll(ge) : 64;      // first probe at or past the target
                if (fl == 0) return 0;
                int hi = fl * step;
                hi = hi < n ? hi : n;                                    // list[hi] >= target (or hi == n); list[(fl - 1) step] < target
                const int base = (fl - 1) * step + 1;
                for (int b = base; b < hi; b += 64) {
                    const int i = b + lane;
                    const uint32_t c = i < hi ? (((uint32_t)list[i] >> 8) & 127u) : 255u;
                    const unsigned long long m = __ballot(c >= (uint32_t)target);
                    if (m) return b + (int)__builtin_ctzll(m);
                }
                return hi;
            };
            const int r_lo = __builtin_amdgcn_readfirstlane(lower(hs * C::HS));
            const int r_hi = __builtin_amdgcn_readfirstlane(lower(hend * C::HS));
            list += r_lo;
            n = r_hi - r_lo;
        }
        const int first = __builtin_amdgcn_readfirstlane(p.first[g]);
        const uint32_t sb0 = (uint32_t)(first * C::ND + (g - first) * NC);

        // ---- the row list travels through a ring of four 64-entry windows in LDS, one window ahead of the tiles ---------------
        const int nwin = (n + 63) >> 6;
        auto load_win = [&](int w) -> uint32_t {      // (entries past the end repeat the last one; never used)
            int i = w * 64 + lane;
            i = i < n ? i : n - 1;
            i = i > 0 ? i : 0;
            return (uint32_t)list[i];
        };
        auto ring_write = [&](int w, uint32_t v) { *(uint16_t*)(lds + ring_off + (((w & 3) * 64 + lane) * 2)) = (uint16_t)v; };
        auto ring_read = [&](int r) -> uint32_t {     // entry r + rr
            return (uint32_t) * (const uint16_t*)(lds + ring_off + (((r + rr) & 255) * 2));
        };
        const uint32_t win0 = load_win(0), win1 = load_win(1);
        uint32_t pend = load_win(2);
        int w_loaded = 2;

        // side data of a half (8 centroids), fetched ahead; every lane fetches centroid lane & 7 (no masked merges, no early
        // waits): centroid position (the [xyz 0] tail of the output row) and the point its self-loop row names (row sb0 + c)
        // (three 12-byte vectors: the loop-carried registers are the load's own register triples - no copy that would wait for the
        // load on the spot)
        typedef float f32x3 __attribute__((ext_vector_type(3)));
        struct Side { f32x3 q, sp, sc; };      // centroid xyz | self-loop point xyz | rgb
        auto fetch_side = [&](int hh) -> Side {
            const int c = hh * C::HS + (lane & (C::HS - 1));
            const float* pc = p.out + ((int64_t)g * NC + c) * (int64_t)p.ldo + N;
            const int64_t ai = (int64_t)sb0 + c;
            return Side{*(const f32x3*)pc, *(const f32x3*)(p.pos_src + ai * 3), *(const f32x3*)(p.feat_src + ai * 3)};
        };
        const Side side0 = fetch_side(hs), side1 = fetch_side(hs + 1);     // the first two halves are opened before the first tile

        // ---- the object's points -> LDS records [r g b x y z] (lane l: points 4 l .. 4 l + 3 = 96 contiguous bytes) --------------
        {
            const f32x4* px4 = (const f32x4*)(p.pos_src + (int64_t)g * C::ND * 3) + lane * 3;
            const f32x4* pc4 = (const f32x4*)(p.feat_src + (int64_t)g * C::ND * 3) + lane * 3;
            const f32x4 a0 = px4[0], a1 = px4[1], a2 = px4[2];      // x0 y0 z0 x1 | y1 z1 x2 y2 | z2 x3 y3 z3
            const f32x4 c0_ = pc4[0], c1_ = pc4[1], c2_ = pc4[2];   // r0 g0 b0 r1 | g1 b1 r2 g2 | b2 r3 g3 b3
            f32x4* rec = (f32x4*)(lds + pts_off + lane * 96);
            rec[0] = f32x4{c0_[0], c0_[1], c0_[2], a0[0]};
            rec[1] = f32x4{a0[1], a0[2], c0_[3], c1_[0]};
            rec[2] = f32x4{c1_[1], a0[3], a1[0], a1[1]};
            rec[3] = f32x4{c1_[2], c1_[3], c2_[0], a1[2]};
            rec[4] = f32x4{a1[3], a2[0], c2_[1], c2_[2]};
            rec[5] = f32x4{c2_[3], a2[1], a2[2], a2[3]};
        }
        ring_write(0, win0);
        ring_write(1, win1);
        PPROF_MARK(1);      // points, list windows -> LDS

        // The accumulator is a WINDOW of 16 centroid slots (slot = centroid & 15) in two halves of 8.  A tile may hold rows of the
        // half of its first row and the next one, so tiles are full (32 rows) wherever the list goes on; a half is drained (relu(max
        // + bias) -> its 8 output rows) once the list has passed it, and its slots start over for the half two further on.
        // The drained rows wait in registers and leave at the next list-window event (flush_out): vmcnt counts stores and loads
        // alike, a store issued here would sit in front of the next load the loop has to wait for.
        auto flush_out = [&]() {
            if (has_pout) {
#pragma unroll
                for (int q4 = 0; q4 < NPL / 4; q4++) *(f32x4*)(pptr + q4 * 4) = pout[q4];
                has_pout = false;
            }
        };
        auto drain = [&](int hh, bool live) {        // live: the half's slots were initialised (else: no rows at all, zeros)
            flush_out();
            pptr = p.out + ((int64_t)g * NC + hh * C::HS + cr) * (int64_t)p.ldo + cs * NPL;
            const uint32_t a0 = acc_off + (uint32_t)((((hh & 1) * C::HS + cr) * N + cs * NPL) * 4);
            int top = gtop;
#pragma unroll
            for (int q4 = 0; q4 < NPL / 4; q4++) {
                f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
                if (live) {
                    const f32x4 raw = *(const f32x4*)(lds + a0 + q4 * 16);
                    const f32x4 bq = *(const f32x4*)(lds + bias_off + q4 * 16);
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const float r = fmaxf(raw[e] + bq[e], 0.f);   // (a centroid without rows stays at -inf: 0)
                        const int bits = __float_as_int(r);
                        top = bits > top ? bits : top;
                        v[e] = r * p.out_scale;
                    }
                }
                pout[q4] = v;
            }
            has_pout = true;
            gtop = top;
        };
        auto open_half = [&](int hh, const Side& sd) {     // slots of half hh: accumulator at -inf, side tables
            const uint32_t s0 = (uint32_t)((hh & 1) * C::HS);
            if (lane < C::HS) {
                uint2* sp = (uint2*)(lds + self_off + (s0 + lane) * 24);
                sp[0] = uint2{__float_as_uint(sd.sc[0]), __float_as_uint(sd.sc[1])};
                sp[1] = uint2{__float_as_uint(sd.sc[2]), __float_as_uint(sd.sp[0])};
                sp[2] = uint2{__float_as_uint(sd.sp[1]), __float_as_uint(sd.sp[2])};
                f32x4* cp = (f32x4*)(lds + cen_off + (s0 + lane) * 32);
                cp[0] = f32x4{0.f, 0.f, 0.f, sd.q[0]};
                cp[1] = f32x4{sd.q[1], sd.q[2], 0.f, 0.f};
            }
            typedef int i32x4 __attribute__((ext_vector_type(4)));
            const uint32_t a0 = acc_off + (uint32_t)(((s0 + cr) * N + cs * NPL) * 4);
#pragma unroll
            for (int q4 = 0; q4 < NPL / 4; q4++)
                *(i32x4*)(lds + a0 + q4 * 16) = i32x4{(int)0xFF800000, (int)0xFF800000, (int)0xFF800000, (int)0xFF800000};
        };

        open_half(hs, side0);
        open_half(hs + 1, side1);
        Side side_n = fetch_side(hs + 2);     // runs one half ahead of the window
        int hf = hs + 2;                      // the half `side_n` belongs to
        int hd = hs, ho = hs + 2;             // halves drained / opened (hd <= ho <= hd + 2)
        int r0 = 0;
        uint32_t e_cur = n > 0 ? ring_read(0) : 0u;
        while (r0 < n) {
            PPROF_COUNT(10);
            // ---- this tile: rows r0 .. r0 + c - 1, all in halves h0 and h0 + 1 (the list is sorted by centroid) -----------------------
            const uint32_t key = r0 + rr < n ? ((e_cur >> 8) & 127u) : 255u;
            const int h0 = __builtin_amdgcn_readfirstlane((int)key) >> 3;
            const uint32_t over = (uint32_t)__ballot(key >= (uint32_t)((h0 + 2) * C::HS));    // (lanes 32 .. 63 repeat 0 .. 31)
            const int c = over ? (int)__builtin_ctz(over) : 32;
            const uint32_t e_nxt = ring_read(r0 + c);     // the next tile's entries (the ring always holds 64 entries past r0)
            if (hd < h0 || ho < h0 + 2) {
                // ---- the window moves: close the halves the list has passed, open those this tile may touch ----------------------------
                if (hd < h0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (their atomics)
                while (hd < h0) {
                    drain(hd, hd < ho);
                    hd++;
                }
                if (ho < hd) ho = hd;        // (halves skipped without a row were never opened)
                const int want = h0 + 2 < hend ? h0 + 2 : hend;
                while (ho < want) {
                    if (hf != ho) side_n = fetch_side(ho);
                    open_half(ho, side_n);
                    ho++;
                    if (ho < hend) {
                        side_n = fetch_side(ho);
                        hf = ho;
                    }
                }
                PPROF_MARK(3);  // window move
            }
            // list ring: one more window into LDS whenever less than 96 entries lie ahead (every other tile); behind its wait the
            // pending output rows leave, then the next window's load
            if (w_loaded * 64 < r0 + 96 && w_loaded < nwin) {
                ring_write(w_loaded, pend);
                w_loaded++;
                flush_out();
                if (w_loaded < nwin) pend = load_win(w_loaded);
            }
            // rows past the tile's end repeat its last row (same maximum)
            const uint32_t m = rr < c ? e_cur : (uint32_t)__builtin_amdgcn_readlane((int)e_cur, c - 1);
            // ---- layer-1 input of this lane's row: [r g b dx] (half 0) / [dy dz 1 0] (half 1) --------------------------------
            const uint32_t src = m & 0xFFu, d = m >> 8;
            const uint32_t dl = d & 15u;                  // slot of the row's centroid in the 16-centroid window
            const uint32_t rec = (d & 0x80u) ? self_off + dl * 24u : pts_off + src * 24u;
            const uint32_t pa0 = rec + (uint32_t)(h * 16), pa1 = h ? one_off : rec + 8u;
            f32x4 pv, cv;
            if constexpr (T2P_PABL & 8) {
                pv = f32x4{(float)pa0, 1.f, 2.f, (float)pa1};
                cv = f32x4{(float)dl, 1.f, 2.f, 3.f};
            } else {
                const f32x2 p0 = *(const f32x2*)(lds + pa0), p1 = *(const f32x2*)(lds + pa1);
                pv = f32x4{p0[0], p0[1], p1[0], p1[1]};
                cv = *(const f32x4*)(lds + cen_off + dl * 32u + (uint32_t)(h * 16));
            }
            *(uint16_t*)(lds + dst_off + rr * 2) = (uint16_t)(dl * (uint32_t)(N * 4));
            uint32_t xh[2], xl[2];
            split2(pv[0] - cv[0], pv[1] - cv[1], xh[0], xl[0]);
            split2(pv[2] - cv[2], pv[3] - cv[3], xh[1], xl[1]);
            const half4 x_hi = __builtin_bit_cast(half4, u32x2{xh[0], xh[1]});
            const half4 x_lo = __builtin_bit_cast(half4, u32x2{xl[0], xl[1]});
            f32x16 hid;
            if constexpr (T2P_PABL & 4) {
#pragma unroll
                for (int e = 0; e < 16; e++) hid[e] = (float)x_hi[e & 3] + (float)x_lo[e & 3];
            } else {
                hid = MFMA8(w1_hi, x_hi, kZero16);
                hid = MFMA8(w1_lo, x_hi, hid);
                hid = MFMA8(w1_hi, x_lo, hid);
            }
            PPROF_MARK(4);  // look-ahead, point reads, layer 1
            uint2 four[4];   // accumulator-row byte offsets of this lane's 16 result rows 8 q + 4 h + {0..3}
            {
                const uint2* f4 = (const uint2*)(lds + dst_off + h * 8);
                four[0] = f4[0], four[1] = f4[2], four[2] = f4[4], four[3] = f4[6];
            }
            // ---- ReLU, fp16 split: the result registers are the layer-2 operand -----------------------------------------------
            f32x16 acc[C::NTW];
#pragma unroll
            for (int s = 0; s < C::S16; s++) {
                uint32_t nh[4], nl[4];
#pragma unroll
                for (int pr = 0; pr < 4; pr++)
                    split2(relu1(hid[8 * s + 2 * pr]), relu1(hid[8 * s + 2 * pr + 1]), nh[pr], nl[pr]);
                const half8 a_hi = __builtin_bit_cast(half8, u32x4{nh[0], nh[1], nh[2], nh[3]});
                const half8 a_lo = __builtin_bit_cast(half8, u32x4{nl[0], nl[1], nl[2], nl[3]});
                if constexpr (T2P_PABL & 2) {
#pragma unroll
                    for (int nt = 0; nt < C::NTW; nt++)
#pragma unroll
                        for (int e = 0; e < 16; e++) acc[nt][e] = (float)a_hi[e & 7] + (float)a_lo[e & 7];
                    continue;
                }
#pragma unroll
                for (int nt = 0; nt < C::NTW; nt++) acc[nt] = MFMA16(a_hi, w_hi[nt][s], s == 0 ? kZero16 : acc[nt]);
#pragma unroll
                for (int nt = 0; nt < C::NTW; nt++) acc[nt] = MFMA16(a_hi, w_lo[nt][s], acc[nt]);
#pragma unroll
                for (int nt = 0; nt < C::NTW; nt++) acc[nt] = MFMA16(a_lo, w_hi[nt][s], acc[nt]);
            }
            PPROF_MARK(5);  // ReLU, split, layer 2
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
                        if constexpr (!(T2P_PABL & 1))
                            asm volatile("ds_max_f32 %0, %1 offset:%2" ::"v"(ad[e]), "v"(acc[nt][e]), "n"(nt * 128) : "memory");
                        else
                            asm volatile("; %0 %1" ::"v"(ad[e]), "v"(acc[nt][e]));
                PPROF_MARK(6);  // atomics
            }
#if T2P_PPROF
            pp_sum[8] += c;
#endif
            r0 += c;
            e_cur = e_nxt;
        }
        // ---- close what is open, write the halves behind the list's end -------------------------------------------------------
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (; hd < hend; hd++) drain(hd, hd < ho);
        flush_out();
        PPROF_MARK(7);      // last drains
    }
#if T2P_PPROF
    if (blockIdx.x == 0 && wave == T2P_PPROF - 1 && lane == 0) {
        for (int i = 0; i < 11; i++) atomicAdd(&t2p_pprof_sums[i], pp_sum[i]);
        atomicAdd(&t2p_pprof_sums[11], __builtin_amdgcn_s_memtime() - pp_begin);
    }
#endif
    uint32_t gbits = 0;
    guard_track_bits(gbits, gtop);
    if (p.amax_out != nullptr && lane == 0 && gbits != 0u)
        atomicMax(p.amax_out, __float_as_uint(__uint_as_float(gbits) * p.out_scale));
}

#ifndef T2P_PTS_WAVES
#define T2P_PTS_WAVES 12
#endif
constexpr int kPtsWaves = T2P_PTS_WAVES;
constexpr int kPtsParts = 4;

}  // namespace

bool sa_points_selected(int H, int Cout, const SaParams& p) {
    return H == 32 && Cout == 64 && p.W_x3 != nullptr && p.wp != nullptr && p.w1 != nullptr;
}

// (tile rows, workgroups) for the range balancing: one 12-wave workgroup per CU, cost = rows
int sa_points_launch_shape(int64_t n_obj, int* tile_rows, int* n_wg) {
    int n = num_cus();
    if (n > 1024) n = 1024;
    if (n > n_obj) n = (int)n_obj;
    *tile_rows = 32;
    *n_wg = n;
    return 0;
}

int launch_sa_points(int H, int Cout, const SaParams& p, hipStream_t st) {
    if (!(H == 32 && Cout == 64 && p.n_cent == 128 && p.n_dense == 256 && p.wp && p.W_x3 && p.w1 && p.b1 && p.feat_src && p.pos_src &&
          p.ld_pos == 3 && p.pos_col0 == 0)) {
        set_error("sa_points: built for SA level 1 (6 -> 32 -> 64, 128 centroids of 256 points [xyz] + [rgb], f16x3)");
        return T2P_E_UNSUPPORTED;
    }
    using C = PtsCfg<kPtsWaves>;
    auto kern = k_sa_points<kPtsWaves, 1>, kern_parts = k_sa_points<kPtsWaves, kPtsParts>;
    T2P_TRY(reserve_lds((const void*)kern, C::lds_bytes(), "sa_points"));
    T2P_TRY(reserve_lds((const void*)kern_parts, C::lds_bytes(), "sa_points"));
    if (p.n_obj <= 0) return 0;
    T2P_CHECK_ARG(p.n_obj < (1 << 22), "sa_points: chunk too large");
    T2P_CHECK_ARG((((uintptr_t)p.pos_src | (uintptr_t)p.feat_src | (uintptr_t)p.out | (uintptr_t)p.W_x3) & 15) == 0 && p.ldo % 4 == 0,
                  "sa_points: points, output rows and weights must be 16-byte aligned");
    int tr, n_wg;
    sa_points_launch_shape(p.n_obj, &tr, &n_wg);
    if (!p.balanced) T2P_TRY(launch_sa_balance(p, tr, n_wg, st));
    ProfScope ps_("ws_edge_sa_k32_n64", st);
    // fewer objects than half the chip's wave slots: a wave per QUARTER object (same bits; see the kernel)
    const bool parts = p.n_obj * 2 <= (int64_t)n_wg * kPtsWaves;
    T2P_REPEAT(ps_) hipLaunchKernelGGL(parts ? kern_parts : kern, dim3(n_wg), dim3(C::NT), C::lds_bytes(), st, p);
    T2P_CHECK_LAUNCH("sa_points");
    return 0;
}

}  // namespace t2p

#if T2P_PPROF
extern "C" void t2p_debug_pprof(unsigned long long* out) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(t2p::t2p_pprof_sums), sizeof(unsigned long long) * 16);
}
#endif

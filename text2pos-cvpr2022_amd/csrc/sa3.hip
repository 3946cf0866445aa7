// Set-abstraction level 3 edge kernel (K = N = 256), f16x3 path, SCALAR control.
// (reference: gnn.PointConv(local_nn)(x, (pos, pos[idx]), edge_index), models/pointcloud/pointnet2.py:31-35).
//
// Data flow of ws_sa2.hip (rounds 1-3; one 8-wave workgroup per CU, wave w owns output columns [32 w, 32 w + 32) with its 256 x 32 weight
// slice in 128 registers, all waves share one staged 32-row batch, four batches in flight, deferred LDS float-max atomics,
// centroid table built in LDS per object) - with everything that is the same for all 64 lanes of a wave moved off the vector
// unit.  At K = 256 one staged row is exactly one wave-wide 16-byte load (64 lanes x 4 floats), so a wave stages WHOLE rows
// (4 w .. 4 w + 3 of the batch) and every per-row quantity is wave-uniform:
//   * the object iterators of the four pipeline stages live in SGPRs; an object's row count and self-loop base come from
//     scalar loads (constant address space: these tables are written by earlier kernels) issued one object ahead - no
//     per-sub-range LDS cache of them, no outer loop around the batch loop;
//   * a batch's row metadata for this wave is ONE s_load_dwordx2 (4 u16 entries), decoded on the scalar unit (source row,
//     centroid, self-loop flag, padding); the row gathers are `global_load_dwordx4 v, v_lane16, s[base]` with a scalar
//     64-bit row base, the centroid-table reads add one scalar to the lane offset;
//   * the destination offsets of the batch's rows go to the 4-batch LDS ring as one 8-byte store per wave.
// ws_sa2.hip's hot loop issued ~11.9 instructions per MFMA (ISA census, profiles/isa_count.py: 467 VALU + 56 moves + 174 LDS
// + 187 SALU + 107 waits per 96 MFMAs), of which only ~64 VALU + ~60 LDS per 48 MFMAs are the batch's arithmetic; on gfx950 every
// VALU instruction next to the MFMA stream costs issue time on the same SIMD (profiles/microbench/mix2.hip), scalar ones do not.
#include "t2p_common.h"

// T2P_SA3_ABL, development only (results wrong, timing valid): 1 = no atomics, 2 = no row gathers, 4 = no staging stores to LDS,
// 8 = no operand reads inside the MFMA loop, 16 = no third MFMA, 32 = no per-batch barrier, 64 = no staging arithmetic,
// 128 = no per-object phases (drain, centroid table).  (Round 4 raised the wave priority around the MFMA pairs of the 32x32x16 form
// with s_setprio; with the 16x16x32 slots that costs 1.5 %: 25.46-25.54 against 25.08-25.15 ms per step in three A/B pairs - removed.)
#ifndef T2P_SA3_ABL
#define T2P_SA3_ABL 0
#endif
// The products run on v_mfma_f32_16x16x32_f16: four 16 x 16 blocks (2 row blocks x 2 column blocks) per wave and batch, 96 MFMAs of
// half the size instead of 48 v_mfma_f32_32x32x16_f16 - the same FLOPs, LDS operand reads and weight registers (a row block's A operand
// feeds both column blocks), but the smaller shape sustains 3-4 % more at the power cap (profiles/microbench/mfma_shapes.hip), and here:
// 26.45-26.64 -> 25.40-25.68 ms per step in three interleaved A/B pairs (round 5).  Lane = 16 q + i holds row 16 rb + i of the A
// operand, column 16 cb + i of the B operand, k = 64 q + 8 s + e of step s (s < 8), and rows 16 rb + 4 q + v of a result block.
#ifndef T2P_SA3_SPLIT_ASM
#define T2P_SA3_SPLIT_ASM 1
#endif
#define SB() __builtin_amdgcn_sched_barrier(0)
#define AS4 __attribute__((address_space(4)))

namespace t2p {
namespace {

constexpr int K = 256, N = 256, NC = 32, ND = 64, TR = 32, NW = 8, NT = 64 * NW;
constexpr int LDHH = K + 8;            // halves per plane row (16-byte pad: conflict-free ds_read_b128)
constexpr int PLANE = TR * LDHH;       // halves per plane
constexpr int S16 = K / 16;
constexpr int MAXR = NC * 33;          // row-list slots per object

// LDS map (bytes)
constexpr int OFF_TILE = 0;                              // [2 buffers][hi plane | lo plane]
constexpr int OFF_ACC = OFF_TILE + 2 * 2 * PLANE * 2;    // [NC + 1][N] fp32 running maxima (row NC: padding rows)
constexpr int OFF_BTAB = OFF_ACC + (NC + 1) * N * 4;     // [NC + 1][K] fp32 centroid table (row NC: zeros)
constexpr int OFF_WP = OFF_BTAB + (NC + 1) * K * 4;      // [3][K] position rows of the layer-1 weights
constexpr int OFF_CPOS = OFF_WP + 3 * K * 4;             // [NC][3] centroid positions of the object being built
constexpr int OFF_BIAS = OFF_CPOS + 512;                 // [N] bias x weight scale
constexpr int OFF_DST = OFF_BIAS + N * 4;                // [4 batches][TR] u16 accumulator byte offset of every staged row
constexpr int LDS_BYTES = OFF_DST + 4 * TR * 2;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef t2p_fp16x2 fp16x2;
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)

template <typename T>
__device__ __forceinline__ const AS4 T* as_const(const T* p) {
    return (const AS4 T*)p;
}
__device__ __forceinline__ void lds_fmax(float* p, float v) {
    (void)__hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <int SEL>
__device__ __forceinline__ float sub_half(float v, fp16x2 h) {   // v - (float)h[SEL] in one VOP3P mixed-precision FMA
    float r;
    if constexpr (SEL == 0)
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(v));
    else
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(v));
    return r;
}
// LDS-only workgroup barrier (no vmcnt wait: the drain's global stores need not have landed)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct It {      // one pipeline stage's position in the workgroup's object range; every member is wave-uniform (SGPRs)
    int g;       // object
    int r0;      // first row of the batch inside the object
    int n;       // rows of the object (0 past the end of the range)
    int sb;      // self-loop base: table row of dense point 0 of the object's cell batch + n_cent * (object's rank in the cell)
};

__global__ __launch_bounds__(NT, 2) void k_sa3(SaParams p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    _Float16* const tile = (_Float16*)(lds + OFF_TILE);
    float* const accl = (float*)(lds + OFF_ACC);
    float* const btab = (float*)(lds + OFF_BTAB);
    float* const wpl = (float*)(lds + OFF_WP);
    float* const cposl = (float*)(lds + OFF_CPOS);
    float* const biasl = (float*)(lds + OFF_BIAS);
    uint16_t* const dstl = (uint16_t*)(lds + OFF_DST);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform: everything derived from it stays on the scalar unit

    // (constant address space = scalar loads: legal because earlier KERNELS wrote these tables - the scalar cache is invalidated at
    // kernel start - and this kernel never writes them.  n_rows holds u16 entries, read as the dword that holds them: the last
    // dword of an odd-length array reaches 2 bytes into the workspace's 256-byte alignment gap behind it)
    const AS4 uint32_t* const n_rows_c = as_const((const uint32_t*)p.n_rows);
    const AS4 int32_t* const first_c = as_const(p.first);
    const AS4 uint32_t* const rows_c = as_const((const uint32_t*)p.rows);
    const AS4 int32_t* const bounds_c = as_const(p.bounds_ws);

    // ---- stationary weights: columns [32 wave, 32 wave + 32), all K, hi / lo planes (packing.py::pack_f16x3 order) ----
    // 16x16x32 operands: lane = 16 q + i holds k = 64 q + 8 s + e of MFMA step s (s < 8), column 16 cb + i of the wave's 32
    constexpr int S32 = K / 32;
    const int q16 = lane >> 4, i16 = lane & 15;
    half8 w_hi[2][S32], w_lo[2][S32];
    {
        const uint4* wp = (const uint4*)p.W_x3;
        constexpr int PLANE_U4 = (N / 32) * S16 * 64;
#pragma unroll
        for (int cb = 0; cb < 2; cb++)
#pragma unroll
            for (int s = 0; s < S32; s++) {
                const int k0 = 64 * q16 + 8 * s;                       // image: k = half' K/2 + 8 step' + e
                const int idx = ((wave * S16 + (k0 % (K / 2)) / 8) * 2 + k0 / (K / 2)) * 32 + 16 * cb + i16;
                w_hi[cb][s] = __builtin_bit_cast(half8, wp[idx]);
                w_lo[cb][s] = __builtin_bit_cast(half8, wp[PLANE_U4 + idx]);
            }
    }
    for (int i = tid; i < N; i += NT) biasl[i] = p.bias[i];
    for (int i = tid; i < (NC + 1) * N; i += NT) accl[i] = -__builtin_inff();
    for (int i = tid; i < 3 * K; i += NT) wpl[i] = p.wp[i];
    for (int i = tid; i < K; i += NT) btab[NC * K + i] = 0.f;          // row NC: what padding rows subtract
    for (int i = tid; i < 4 * TR; i += NT) dstl[i] = (uint16_t)(NC * N * 4);   // "batches before the first": all padding
    int gtop = 0;   // fp16-range guard: this lane's largest drained output (bit pattern of a non-negative float)

    const int g_begin = bounds_c[blockIdx.x], g_end = bounds_c[blockIdx.x + 1];

    auto obj_n = [&](int g) -> int { return g < g_end ? (int)((n_rows_c[g >> 1] >> ((g & 1) * 16)) & 0xFFFFu) : 0; };
    auto obj_sb = [&](int g) -> int {
        if (g >= g_end) return 0;
        const int f = first_c[g];
        return f * ND + (g - f) * NC;
    };
    auto advance = [&](It it) -> It {
        it.r0 += TR;
        if (it.r0 >= it.n) {
            it.g++;
            it.r0 = 0;
            it.n = obj_n(it.g);
            it.sb = obj_sb(it.g);
        }
        return it;
    };
    auto valid = [&](const It& it) { return it.g < g_end; };

    // ---- centroid table B_i = W1p pos_i of an object, built in LDS from positions prefetched one object ahead -------
    const int c4 = lane;                         // this lane's column quad (columns 4 c4 .. 4 c4 + 3)
    float npos = 0.f;
    auto fetch_pos = [&](int g) {
        if (tid < 3 * NC && g < g_end) npos = p.out[((int64_t)g * NC + tid / 3) * (int64_t)p.ldo + N + tid % 3];
    };
    auto build_b = [&](int g) {                  // ends with a barrier
        if (tid < 3 * NC) cposl[tid] = npos;
        fetch_pos(g + 1);
        lds_barrier();
        const f32x4 w0 = *(const f32x4*)(wpl + c4 * 4), w1 = *(const f32x4*)(wpl + K + c4 * 4), w2 = *(const f32x4*)(wpl + 2 * K + c4 * 4);
#pragma unroll
        for (int i = 0; i < NC / NW; i++) {
            const int c = wave + NW * i;
            const float px = cposl[3 * c], py = cposl[3 * c + 1], pz = cposl[3 * c + 2];
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; e++) {        // same order as k_sample_group's table: ((x w0) + y w1) + z w2
                float a = px * w0[e];
                a = fmaf(py, w1[e], a);
                a = fmaf(pz, w2[e], a);
                v[e] = a;
            }
            *(f32x4*)(btab + c * K + c4 * 4) = v;
        }
        lds_barrier();
    };
    // ---- drain of a finished object: relu(max + bias) x out_scale -> HBM, accumulator back to -inf -----------------
    auto flush = [&](int g) {
        float* o = p.out + (int64_t)g * NC * (int64_t)p.ldo;
        int top = gtop;
        typedef int i32x4 __attribute__((ext_vector_type(4)));
#pragma unroll 2
        for (int i = tid; i < NC * (N / 4); i += NT) {
            const int c = i / (N / 4), col = (i % (N / 4)) * 4;
            const f32x4 raw = *(const f32x4*)(accl + c * N + col);
            const f32x4 bq = *(const f32x4*)(biasl + col);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float r = fmaxf(raw[e] + bq[e], 0.f);
                const int bits = __float_as_int(r);
                top = bits > top ? bits : top;
                v[e] = r * p.out_scale;
            }
            *(f32x4*)(o + c * (int64_t)p.ldo + col) = v;
            *(i32x4*)(accl + c * N + col) = i32x4{(int)0xFF800000, (int)0xFF800000, (int)0xFF800000, (int)0xFF800000};
        }
        gtop = top;
    };

    // ---- per-row state of the staging pipeline ---------------------------------------------------------------------
    f32x4 sa[4];         // gathered A_j rows of the batch being staged (this lane's column quad)
    f32x4 bq;            // centroid-table entries of the row staged next
    uint32_t boff[4];    // (uniform) byte offset of each staged row's centroid inside the LDS table
    f32x4 vv;
    fp16x2 vh01, vh23;
    const uint32_t lane16 = (uint32_t)lane * 16u;

    // row metadata of a batch for THIS wave: 4 u16 entries = one 8-byte scalar load
    auto load_meta = [&](const It& it) -> uint2 {
        uint2 m{0xFFFFFFFFu, 0xFFFFFFFFu};
        if (valid(it)) {
            const uint32_t e = (uint32_t)it.g * (uint32_t)MAXR + (uint32_t)(it.r0 + 4 * wave);   // (even: 8-byte aligned pair of dwords)
            m.x = rows_c[e / 2];
            m.y = rows_c[e / 2 + 1];
        }
        return m;
    };
    // gather of row k of a batch + its centroid offset; the 4 destination offsets of the wave go to the ring in one store
    auto issue = [&](const It& it, const uint2& m, int k, uint32_t& dpack_lo, uint32_t& dpack_hi) {
        const uint32_t word = (k & 2) ? m.y : m.x;
        const uint32_t e16 = (k & 1) ? (word >> 16) : (word & 0xFFFFu);
        const bool pad = !valid(it) || it.r0 + 4 * wave + k >= it.n;
        const uint32_t mm = pad ? 0u : e16;
        const uint32_t src = mm & 0xFFu, d = mm >> 8, dl = pad ? (uint32_t)NC : (d & 127u);
        const uint32_t g = (uint32_t)(valid(it) ? it.g : g_end - 1);
        const uint32_t srow = (d & 0x80u) ? ((uint32_t)it.sb + src) : (g * (uint32_t)ND + src);
        const float* rowp = p.A + (size_t)srow * K;                       // scalar 64-bit row base
        if constexpr (T2P_SA3_ABL & 2) {
            const float f = __uint_as_float(((uint32_t)(uintptr_t)rowp & 0xFFFFu) | 0x3f000000u);
            sa[k] = f32x4{f, f, f, f};
        } else
            sa[k] = *(const f32x4*)((const char*)rowp + lane16);
        boff[k] = dl * (uint32_t)(K * 4);
        const uint32_t dv = dl * (uint32_t)(N * 4);
        if (k == 0) dpack_lo = dv;
        else if (k == 1) dpack_lo |= dv << 16;
        else if (k == 2) dpack_hi = dv;
        else dpack_hi |= dv << 16;
    };
    auto load_b = [&](int k) { bq = *(const f32x4*)((const char*)btab + boff[k] + lane16); };
    auto stage_a = [&](int buf, int k) {         // v = relu(A_j - B_i), hi = fp16(v) -> hi plane
        _Float16* dsth = tile + buf * 2 * PLANE;
        uint2 ph;
        if constexpr (T2P_SA3_ABL & 64) {
            ph.x = __float_as_uint(sa[k][0]) ^ __float_as_uint(bq[0]);
            ph.y = __float_as_uint(sa[k][1]) ^ __float_as_uint(bq[1]);
            vv = sa[k];
        } else {
            const f32x4 t = sa[k] - bq;
#pragma unroll
            for (int e = 0; e < 4; e++) vv[e] = fmaxf(t[e], 0.f);
            vh01 = cvt_pk_f16(vv[0], vv[1]);
            vh23 = cvt_pk_f16(vv[2], vv[3]);
            ph.x = __builtin_bit_cast(uint32_t, vh01);
            ph.y = __builtin_bit_cast(uint32_t, vh23);
        }
        if constexpr (T2P_SA3_ABL & 4) asm volatile("" ::"v"(ph.x), "v"(ph.y));
        else *(uint2*)(dsth + (4 * wave + k) * LDHH + c4 * 4) = ph;
    };
    auto stage_b = [&](int buf, int k) {         // lo = fp16(v - hi) -> lo plane
        _Float16* dsth = tile + buf * 2 * PLANE;
        uint2 pl;
        if constexpr (T2P_SA3_ABL & 64) {
            pl.x = __float_as_uint(vv[2]);
            pl.y = __float_as_uint(vv[3]);
        } else {
#if T2P_SA3_SPLIT_ASM
            pl.x = split_lo_pk(vh01, vv[0], vv[1]);     // (one asm statement per pair: no s_nop between the pieces, t2p_common.h)
            pl.y = split_lo_pk(vh23, vv[2], vv[3]);
#else
            const fp16x2 l01 = cvt_pk_f16(sub_half<0>(vv[0], vh01), sub_half<1>(vv[1], vh01));
            const fp16x2 l23 = cvt_pk_f16(sub_half<0>(vv[2], vh23), sub_half<1>(vv[3], vh23));
            pl.x = __builtin_bit_cast(uint32_t, l01);
            pl.y = __builtin_bit_cast(uint32_t, l23);
#endif
        }
        if constexpr (T2P_SA3_ABL & 4) asm volatile("" ::"v"(pl.x), "v"(pl.y));
        else *(uint2*)(dsth + PLANE + (4 * wave + k) * LDHH + c4 * 4) = pl;
    };
    auto put_dst = [&](int slot, uint32_t lo, uint32_t hi) {
        if (lane == 0) *(uint2*)(dstl + slot * TR + 4 * wave) = uint2{lo, hi};
    };

    if (g_begin >= g_end) return;   // (uniform)

    // ---- prologue: tile 0 staged, gathers of batch 1 in flight, metadata of batch 2 in registers --------------------
    It it_c{g_begin, 0, obj_n(g_begin), obj_sb(g_begin)};
    It it_s = advance(it_c);
    It it_g = advance(it_s);
    It it_m = advance(it_g);
    fetch_pos(g_begin);
    uint32_t dlo = 0, dhi = 0;
    {
        const uint2 m0 = load_meta(it_c);
#pragma unroll
        for (int k = 0; k < 4; k++) issue(it_c, m0, k, dlo, dhi);
    }
    __syncthreads();                 // LDS initialisation above
    put_dst(0, dlo, dhi);
    build_b(g_begin);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        load_b(k);
        stage_a(0, k);
        stage_b(0, k);
    }
    {
        const uint2 m1 = load_meta(it_s);
#pragma unroll
        for (int k = 0; k < 4; k++) issue(it_s, m1, k, dlo, dhi);
        put_dst(1, dlo, dhi);
    }
    uint2 meta_g = load_meta(it_g);
    __syncthreads();

    int flush_g = -1, flush_g1 = -1;
    // deferred atomics: the maxima of batch t-1 ride between the MFMAs of batch t, fed from the result array batch t-1 filled;
    // two result arrays swap roles from batch to batch (the loop is unrolled by two)
    f32x16 rr[2];
#pragma unroll
    for (int e = 0; e < 16; e++) rr[0][e] = rr[1][e] = -__builtin_inff();
    float* const acc_col = accl + wave * 32 + i16;       // this lane's column (of column block 0) in accumulator row 0

    int t = 0;
    int newest = 0;
    for (; valid(it_c);) {
#pragma unroll
        for (int half = 0; half < 2; half++, t++) {
            if (half > 0 && !valid(it_c)) break;
            {
                bool fence = false;
                if (flush_g >= 0 && !(T2P_SA3_ABL & 128)) {     // the object whose last atomics ran in the previous batch drains
                    flush(flush_g);
                    flush_g = -1;
                    fence = true;                       // the next object's atomics (inside this batch) must not overtake the drain
                }
                if (valid(it_s) && it_s.r0 == 0 && !(T2P_SA3_ABL & 128)) {   // batch t+1 opens a new object: its centroid table replaces the current one
                    build_b(it_s.g);
                    fence = false;
                }
                if (fence) lds_barrier();
            }
            load_b(0);
            const int buf = half, sbuf = half ^ 1;     // (t is even whenever half == 0: the loop runs two batches per trip)
            const int dslot = (t + 2) & 3;
            It it_n = it_m;
            f32x16& acc = rr[half ^ 1];
            const f32x16& prev = rr[half];
            // result element e = 8 rb + 4 cb + v of this lane: row 16 rb + 4 q16 + v, column 16 cb + i16 of the wave's 32
            uint2 four[2];   // destination offsets of rows 16 rb + 4 q16 + {0..3} of batch t-1 (written three batches ago)
            {
                const uint16_t* dl = dstl + ((t + 3) & 3) * TR;
#pragma unroll
                for (int rb = 0; rb < 2; rb++) four[rb] = *(const uint2*)(dl + 16 * rb + 4 * q16);
            }
            const _Float16* hrow = tile + buf * 2 * PLANE + i16 * LDHH + q16 * 64;
            half8 a_hi[2], a_lo[2];
#pragma unroll
            for (int rb = 0; rb < 2; rb++) {
                a_hi[rb] = *(const half8*)(hrow + rb * 16 * LDHH);
                a_lo[rb] = *(const half8*)(hrow + PLANE + rb * 16 * LDHH);
            }
            uint2 meta_m{0xFFFFFFFFu, 0xFFFFFFFFu};
            typedef float f32x4v __attribute__((ext_vector_type(4)));
            const f32x4v kZero4 = {0.f, 0.f, 0.f, 0.f};
            f32x4v blk[4];   // [2 rb + cb]
#pragma unroll
            // slot = (step, block): its 3 MFMAs back to back, then the slot's share of the staging work (round 5 A/B, three pairs each:
            // [hh, hl | work | lh] 25.28-25.31 ms per step, this order 24.78-24.95, no sched_barrier fences 25.96-26.11)
            for (int sl = 0; sl < 4 * S32; sl++) {
                const int s32 = sl >> 2, b = sl & 3, rb = b >> 1, cb = b & 1;
                SB();
                // operands of the NEXT step: row block 1 - rb is idle while block rb multiplies (no second operand set)
                if (s32 + 1 < S32 && !(T2P_SA3_ABL & 8)) {
                    if (b == 2) {
                        a_hi[0] = *(const half8*)(hrow + (s32 + 1) * 8);
                        a_lo[0] = *(const half8*)(hrow + PLANE + (s32 + 1) * 8);
                    }
                }
                if (s32 > 0 && b == 0 && !(T2P_SA3_ABL & 8)) {
                    a_hi[1] = *(const half8*)(hrow + 16 * LDHH + s32 * 8);
                    a_lo[1] = *(const half8*)(hrow + PLANE + 16 * LDHH + s32 * 8);
                }
                blk[b] = MFMA32(a_hi[rb], w_hi[cb][s32], s32 == 0 ? kZero4 : blk[b]);
                blk[b] = MFMA32(a_hi[rb], w_lo[cb][s32], blk[b]);
                if constexpr (!(T2P_SA3_ABL & 16)) blk[b] = MFMA32(a_lo[rb], w_hi[cb][s32], blk[b]);
                SB();
                {
                    const int w = sl;
                    if (w == 0) meta_m = load_meta(it_m);          // M(t+3)
                    // the 12 staging chunks ride in the FIRST 12 of the 32 slots (the gathers of batch t + 2 leave early, the staged planes
                    // are complete long before the barrier), the 16 deferred atomics in the LAST 16: 23.1-23.3 ms per step against
                    // 24.2-24.3 with both spread evenly over the slots (three A/B pairs; other placements in docs/notebook.md)
#pragma unroll
                    for (int c = (w < 12 ? w : 12); c < (w < 12 ? w + 1 : 12); c++) {
                        const int k = c / 3, part = c % 3;
                        if (part == 0) stage_a(sbuf, k);
                        else if (part == 1) stage_b(sbuf, k);
                        else {
                            issue(it_g, meta_g, k, dlo, dhi);
                            if (k + 1 < 4) load_b(k + 1);
                            else put_dst(dslot, dlo, dhi);
                        }
                    }
                    if (w >= 16) {                                 // one deferred atomic per slot of the second half: 16 per batch
                        const int e = w - 16, erb = e >> 3, ecb = (e >> 2) & 1, v = e & 3;
                        const uint32_t pair = (v & 2) ? four[erb].y : four[erb].x;
                        const uint32_t off = (v & 1) ? (pair >> 16) : (pair & 0xFFFFu);
                        if constexpr (T2P_SA3_ABL & 1) asm volatile("" ::"v"(prev[e]), "v"(off));
                        else lds_fmax((float*)((char*)acc_col + off) + 16 * ecb, prev[e]);
                    }
                    if (w == 2 * S32 - 1) it_n = advance(it_m);
                }
            }
            // (row block 1's operands for step s are fetched at the first slot of step s: see above - the read of step 0 is the
            // initial one)
#pragma unroll
            for (int b = 0; b < 4; b++)
#pragma unroll
                for (int v = 0; v < 4; v++) acc[8 * (b >> 1) + 4 * (b & 1) + v] = blk[b][v];
            SB();
            const bool obj_done = it_c.r0 + TR >= it_c.n;
            newest = half ^ 1;
            flush_g = flush_g1;
            flush_g1 = obj_done ? it_c.g : -1;
            meta_g = meta_m;
            it_c = it_s;
            it_s = it_g;
            it_g = it_m;
            it_m = it_n;
            if constexpr (!(T2P_SA3_ABL & 32)) __syncthreads();
        }
    }
    if (flush_g >= 0) {
        flush(flush_g);
        lds_barrier();
    }
    {   // drain: atomics of the last batch, then its object
        const uint16_t* dl = dstl + ((t + 3) & 3) * TR;
        const f32x16& last = rr[newest];
        uint2 four[2];
#pragma unroll
        for (int rb = 0; rb < 2; rb++) four[rb] = *(const uint2*)(dl + 16 * rb + 4 * q16);
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int erb = e >> 3, ecb = (e >> 2) & 1, v = e & 3;
            const uint32_t pair = (v & 2) ? four[erb].y : four[erb].x;
            const uint32_t off = (v & 1) ? (pair >> 16) : (pair & 0xFFFFu);
            if constexpr (T2P_SA3_ABL & 1) asm volatile("" ::"v"(last[e]), "v"(off));
            else lds_fmax((float*)((char*)acc_col + off) + 16 * ecb, last[e]);
        }
        __syncthreads();
        if (flush_g1 >= 0) flush(flush_g1);
    }
    uint32_t gbits = 0;
    guard_track_bits(gbits, gtop);
    if (p.amax_out != nullptr && lane == 0 && gbits != 0u)
        atomicMax(p.amax_out, __float_as_uint(__uint_as_float(gbits) * p.out_scale));
}

}  // namespace

bool sa3_selected(int H, int C, const SaParams& p) {
    return H == 256 && C == 256 && p.W_x3 != nullptr && p.wp != nullptr;
}

int sa3_launch_shape(int64_t n_obj, int* tile_rows, int* n_wg) {
    int n = matrix_wgs();
    if (n > 1024) n = 1024;
    if (n > n_obj) n = (int)n_obj;
    *tile_rows = TR;
    *n_wg = n;
    return 0;
}

int launch_sa3(const SaParams& p, hipStream_t st) {
    T2P_CHECK_ARG(p.n_cent == NC && p.n_dense == ND, "sa3: built for %d dense points / %d centroids per object (got %d / %d)", ND, NC,
                  p.n_dense, p.n_cent);
    T2P_CHECK_ARG(p.balanced, "sa3: needs the balanced object ranges of launch_sa_balance_levels");
    T2P_TRY(reserve_lds((const void*)k_sa3, LDS_BYTES, "sa3"));
    if (p.n_obj <= 0) return 0;
    T2P_CHECK_ARG(((uintptr_t)p.out & 15) == 0 && p.ldo % 4 == 0 && ((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.rows & 7) == 0 &&
                      ((uintptr_t)p.n_rows & 3) == 0,
                  "sa3: tables and output rows must be 16-byte aligned (ldo = %d)", p.ldo);
    T2P_CHECK_ARG(p.n_obj * (int64_t)ND < (1LL << 31), "sa3: chunk too large for 32-bit row indices");
    int tr, n_wg;
    T2P_TRY(sa3_launch_shape(p.n_obj, &tr, &n_wg));
    ProfScope ps_("ws_edge_sa_k256_n256", st);
    T2P_REPEAT(ps_) hipLaunchKernelGGL(k_sa3, dim3(n_wg), dim3(NT), LDS_BYTES, st, p);
    T2P_CHECK_LAUNCH("sa3");
    return 0;
}

}  // namespace t2p

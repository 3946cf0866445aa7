// Set-abstraction edge kernel, f16x3 path, instruction-interleaved schedule.
// (reference: gnn.PointConv(local_nn)(x, (pos, pos[idx]), edge_index), models/pointcloud/pointnet2.py:31-35).
//
// Same data flow as ws_sa.hip (balanced contiguous object ranges, flattened batch stream, register-resident weights,
// LDS atomic-max accumulator) but the batch loop is ONE straight-line block whose instruction order is pinned by hand:
//
//   * four batches are in flight per workgroup: MFMA on tile t (LDS), staging of tile t+1 (gathered rows held in
//     registers -> ReLU(A_j - B_i) -> fp16 hi/lo planes in the other LDS buffer), gathers of batch t+2 (re-issued into
//     each register quad as soon as its t+1 content is staged), row metadata of batch t+3;
//   * the staging / gather work is cut into 12 chunks that are placed BETWEEN the MFMAs of the 16 (or 8, or 4) MFMA
//     groups with __builtin_amdgcn_sched_barrier fences,
//     so VALU, LDS, the texture path and the matrix pipe work at the same time inside every wave instead of taking
//     turns between barriers (measured on the phase-per-barrier kernel: MFMA, gather and staging time simply added up);
//   * all loads are unconditional: padding rows read row 0 of their object and are routed to the accumulator's dummy
//     row, so the block has no branches besides the (rare) accumulator flush.
// T2P_TRACE: wave 0 of block 0 stamps s_memtime at 8 points of its first 96 batches into prefix_ws; printed by the launcher
#ifndef T2P_TRACE
#define T2P_TRACE 0
#endif
// T2P_ABL (development only, results are wrong): 1 = no A_j gathers, 2 = no B_i gathers, 4 = no atomics, 8 = no third MFMA,
// 16 = no special-casing of padding rows (metadata taken as it comes, no clean-up past the object's end)
#ifndef T2P_ABL
#define T2P_ABL 0
#endif
#ifndef T2P_DEFER256   // deferred atomics at K = 256 too (the float-max form freed the bias block's registers)
#define T2P_DEFER256 1
#endif
#ifndef T2P_SA1_SWIZ   // 0: K = 32 stages its rows in plain order (A/B of the bank-conflict fix)
#define T2P_SA1_SWIZ 1
#endif
#define SB() __builtin_amdgcn_sched_barrier(0)
#include "t2p_common.h"

namespace t2p {
int launch_sa_balance(const SaParams& p, int tile_rows, int n_wg, hipStream_t st);  // ws_sa.hip

namespace {

constexpr int kSub = 512;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));

// NW = waves per workgroup.  8: one workgroup per CU (two waves per SIMD that share every barrier).  4: TWO independent
// workgroups per CU (one wave per SIMD each): the two waves of a SIMD then belong to different workgroups, drift apart
// and cover each other's non-MFMA phases; needs <= 80 KB of LDS, which the single-buffered accumulator allows for K <= 128.
// WGS = workgroups per CU (1 or 2; 2 needs <= 80 KB of LDS each, hence a single accumulator buffer).
// BL = 1: the object's centroid table B_i = W1p pos_i is BUILT IN LDS by the kernel (3 fmas per entry, once per object)
// instead of being written to HBM by k_sample_group and gathered row by row through the vector-memory path: that path
// delivers ~30 B/clk/CU and B_i rows were half of its bytes (profiles/microbench/README.md).  The 33 KB table takes
// the place of the second accumulator buffer, so a finished object drains behind a barrier instead of under the next
// object's MFMAs (~1 % of an object's time).
template <int K, int N, int WN, int RT, int NW, int WGS, int BL = 0>
struct Cfg2 {
    static constexpr int NT = 64 * NW;
    static constexpr int ACC_BUFS = (WGS == 1 && !BL) ? 2 : 1;
    static constexpr int WM = NW / WN;
    static constexpr int NTW = N / (32 * WN);
    static constexpr int TR = WM * RT * 32;
    static constexpr int F4_PER_ROW = K / 4;
    static constexpr int ITERS = TR * F4_PER_ROW / NT;
    static_assert((ITERS == 4 || ITERS == 2) && TR * F4_PER_ROW == ITERS * NT,
                  "every thread stages 2 or 4 consecutive rows of one column quad");
    static constexpr int LDHH = K + 8;       // halves; 16-byte pad keeps ds_read_b128 conflict-free
    static constexpr int PLANE = TR * LDHH;  // halves per plane
    static constexpr int S16 = K / 16;
    static constexpr int ACC_INTS = 8192 + N;
    static constexpr int NG = S16 * RT;      // MFMA groups per batch and wave
    static constexpr int NCH = 3 * ITERS;    // staging chunks per batch and thread
    static constexpr int BT_FLOATS = BL ? 8192 + K + 3 * K + 256 : 0;  // [n_cent + 1][K] centroid table + W1p [3][K] + [n_cent][3]
    static_assert(!BL || K == N, "the staged rows address the centroid table with the accumulator's byte offsets");
    static constexpr size_t lds_bytes() {
        return (size_t)2 * 2 * PLANE * 2 + (size_t)ACC_BUFS * ACC_INTS * 4 + 4 * TR * 2 + kSub * 2 + kSub * 4 +
               (size_t)BT_FLOATS * 4 + (size_t)N * 4;
    }
};

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
// float max into LDS without a return value (ds_max_f32; a builtin, so that hipcc places the MFMA -> LDS-data wait states itself)
__device__ __forceinline__ void lds_fmax(float* p, float v) {
    (void)__hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

#if T2P_TRACE
#define STAMP(i)                                                                                           \
    if (blockIdx.x == 0 && tid == 0 && trace_n < 96) p.prefix_ws[trace_n * 8 + (i)] = (int)__builtin_amdgcn_s_memtime();
#else
#define STAMP(i)
#endif

// v - (float)h[SEL] with the conversion folded into the FMA (VOP3P mixed-precision FMA, op_sel picks the half)
template <int SEL>
__device__ __forceinline__ float sub_half(float v, fp16x2 h) {
    float r;
    if constexpr (SEL == 0)
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(v));
    else
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(v));
    return r;
}

__device__ __forceinline__ void abl_keep(const f32x16& v, uint32_t a, const void* q) {  // T2P_ABL: keeps the operands live
    asm volatile("" ::"v"(v), "v"(a), "v"(q));
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a workgroup-scope fence, which on gfx950 waits
// for every outstanding global STORE (vmcnt(0), and with it for the gathers queued behind them): right after an object's
// drain to HBM that exposes a full store round trip.  Where only LDS contents are handed between waves, waiting for the
// wave's own LDS operations and meeting at s_barrier is enough.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct BatchIt {
    int gi, r0, n;
    int sb;  // source row of centroid 0's self loop for the current object (kept from the last object past the end)
};

#define CHUNKS()                                                                              \
    _Pragma("unroll") for (int c = (j * C::NCH) / C::NG; c < ((j + 1) * C::NCH) / C::NG; c++) { \
        const int k = c / 3, part = c % 3;                                                     \
        if (part == 0) stage_a(sbuf, k);                                                       \
        else if (part == 1) stage_b(sbuf, k);                                                  \
        else {                                                                                 \
            issue(it_g, meta_g, k, dbuf);                                                      \
            if constexpr (BL) { if (k + 1 < C::ITERS) load_b(k + 1); }                         \
        }                                                                                      \
    }

template <int K, int N, int WN, int RT, int NW, int WGS, int BL>
__global__ __launch_bounds__(64 * NW, NW * WGS / 4) void k_ws_sa2(SaParams p) {
    using C = Cfg2<K, N, WN, RT, NW, WGS, BL>;
    constexpr int IT = C::ITERS;
    constexpr int NT = C::NT;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    _Float16* hidh = (_Float16*)lds;                            // [2][hi plane | lo plane]
    int* acc_lds = (int*)(hidh + 2 * 2 * C::PLANE);             // [2][ACC_INTS]
    // [4][TR] destination of every staged row as the BYTE offset of its accumulator row (centroid * N * 4 <= 0x8000):
    // the atomics then need one add per address (SDWA picks the 16-bit half), not extract + multiply + add
    uint16_t* dstl = (uint16_t*)(acc_lds + C::ACC_BUFS * C::ACC_INTS);
    uint16_t* nr = dstl + 4 * C::TR;                            // [kSub]
    int* sbase = (int*)(nr + kSub);                             // [kSub]
    float* btab = (float*)(sbase + kSub);                       // BL: [n_cent + 1][K] centroid table of the current object
    float* wpl = btab + 8192 + K;                               // BL: [3][K] position rows of the layer-1 weights
    float* cposl = wpl + 3 * K;                                 // BL: [n_cent][3] centroid positions of the object being built
    float* biasl = (float*)((char*)lds + C::lds_bytes()) - N;   // [N] bias (pre-multiplied by the weight scale), for the drain

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave % WN, wm = wave / WN, h = lane >> 5, l31 = lane & 31;
    const int nc = p.n_cent;
    const int maxr = nc * 33;

    half8 w_hi[C::NTW][C::S16], w_lo[C::NTW][C::S16];
    {
        const uint4* wp = (const uint4*)p.W_x3;
        constexpr int PLANE_U4 = (N / 32) * C::S16 * 64;
#pragma unroll
        for (int nt = 0; nt < C::NTW; nt++)
#pragma unroll
            for (int s = 0; s < C::S16; s++) {
                const int idx = (((wn * C::NTW + nt) * C::S16 + s) * 2 + h) * 32 + l31;
                const uint4 a = wp[idx], b = wp[PLANE_U4 + idx];
                w_hi[nt][s] = __builtin_bit_cast(half8, a);
                w_lo[nt][s] = __builtin_bit_cast(half8, b);
            }
    }
    // The bias is NOT part of the accumulation (sa_rows.hip's form): a batch's first MFMA starts from a literal 0, the LDS accumulator
    // takes a FLOAT max of the raw products from a -inf start, and the drain forms relu(max + bias).  A bias block as the first
    // MFMA's C operand costs 16 registers per column tile - the registers the deferred atomics need at K = 256.
    constexpr f32x16 kZero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < N; i += NT) biasl[i] = p.bias[i];
    for (int i = tid; i < C::ACC_BUFS * C::ACC_INTS; i += NT) acc_lds[i] = (int)0xFF800000;   // -inf
    int gtop = 0;        // fp16-range guard: this lane's maximum (bit pattern, before out_scale) of the drained outputs; reduced
                         // over the wave once, at the end

    const int g_begin = p.bounds_ws[blockIdx.x], g_end = p.bounds_ws[blockIdx.x + 1];
    // first of this thread's ITERS staged rows inside a batch.  K = 32: a row is 8 lanes x 8 bytes, a ds_write_b64 is served in
    // groups of 16 lanes = two rows, and with the 80-byte row stride rows r and r + 2 overlap in half of the 32 banks
    // (SQ_LDS_BANK_CONFLICT: 15 % of SA1's LDS cycles) while r and r + 4 are disjoint: neighbouring lane groups swap row pairs
    // so that a 16-lane group covers rows r and r + 4.
    const int tgrp = tid / C::F4_PER_ROW;
    const int rgrp = ((C::F4_PER_ROW == 8 && T2P_SA1_SWIZ) ? ((tgrp & ~3) | ((tgrp & 1) << 1) | ((tgrp >> 1) & 1)) : tgrp) * C::ITERS;
    const int c4 = tid % C::F4_PER_ROW;
    const uint32_t c4b = (uint32_t)c4 * 16u;
    typedef uint16_t metav __attribute__((ext_vector_type(C::ITERS)));

    // ---- BL: centroid table in LDS ---------------------------------------------------------------------------------
    // The positions of an object's centroids sit in the [xyz | 0] tail of this level's output rows (written by
    // k_sample_group).  Thread t < 3 n_cent keeps ONE coordinate of the NEXT object in a register, fetched one object ahead,
    // so a build waits for no memory: it drops the coordinates into LDS, and thread (row group cg, column quad c4) then
    // forms the entries [c][4 c4 .. 4 c4 + 3] of the centroids c = cg + CGS i.
    constexpr int CGS = NT / C::F4_PER_ROW;              // row groups of the workgroup
    constexpr int CPT = BL ? (8192 / K) / CGS : 1;       // centroids per thread (n_cent / CGS)
    const int cg = tid / C::F4_PER_ROW;
    float npos = 0.f;
    auto fetch_pos = [&](int64_t g) {
        if constexpr (BL) {
            if (tid < 3 * nc && g < g_end) npos = p.out[(g * nc + tid / 3) * (int64_t)p.ldo + N + tid % 3];
        }
    };
    auto build_b = [&](int64_t g) {  // from the prefetched positions of object g; then prefetch g + 1.  Ends with a barrier.
        if constexpr (BL) {
            if (tid < 3 * nc) cposl[tid] = npos;
            fetch_pos(g + 1);
            lds_barrier();
            const f32x4 w0 = *(const f32x4*)(wpl + c4 * 4), w1 = *(const f32x4*)(wpl + K + c4 * 4),
                        w2 = *(const f32x4*)(wpl + 2 * K + c4 * 4);
#pragma unroll
            for (int i = 0; i < CPT; i++) {
                const int c = cg + CGS * i;
                const float px = cposl[3 * c], py = cposl[3 * c + 1], pz = cposl[3 * c + 2];
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; e++) {   // same order as k_sample_group's table: ((x w0) + y w1) + z w2
                    float a = px * w0[e];
                    a = fmaf(py, w1[e], a);
                    a = fmaf(pz, w2[e], a);
                    v[e] = a;
                }
                *(f32x4*)(btab + c * K + c4 * 4) = v;
            }
            lds_barrier();
        }
    };
    if constexpr (BL) {
        for (int i = tid; i < 3 * K; i += NT) wpl[i] = p.wp[i];
        for (int i = tid; i < K; i += NT) btab[8192 + i] = 0.f;   // row n_cent: what padding rows subtract
        fetch_pos(g_begin);
    }

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
                if (it.gi < cnt) it.sb = sbase[it.gi];
            }
            return it;
        };
        auto valid = [&](const BatchIt& it) { return it.gi < cnt; };

        metav meta_g, meta_m;
        f32x4 sa[IT], sb[BL ? 1 : IT];
        uint32_t boff[BL ? IT : 1];   // BL: byte offset of the staged row's centroid inside the LDS table
        f32x4 bq;                     // BL: that centroid's table entries, fetched one staging chunk ahead of their use
        auto load_b = [&](int k) {
            if constexpr (BL) bq = *(const f32x4*)((const char*)btab + boff[k]);
        };
        f32x4 vv;            // staged values between the two halves of a staging step
        fp16x2 vh01, vh23;

        // row metadata of a batch: the load is issued early, the clean-up of rows past the object's end runs at the
        // end of the batch (touching the value earlier would put a vmcnt wait in front of the MFMAs)
        auto load_meta = [&](const BatchIt& it, metav& m) {
#pragma unroll
            for (int k = 0; k < IT; k++) m[k] = 0xFFFF;
            if (valid(it) && it.r0 + rgrp < it.n) {
                const uint32_t off = (uint32_t)(ga + it.gi) * (uint32_t)maxr + (uint32_t)(it.r0 + rgrp);
                m = *(const metav*)(p.rows + off);
            }
        };
        auto fix_meta = [&](const BatchIt& it, metav& m) {
            if constexpr (T2P_ABL & 16) return;
#pragma unroll
            for (int k = 0; k < IT; k++)
                if (it.r0 + rgrp + k >= it.n) m[k] = 0xFFFF;
        };
        // gathers of row k of a batch (A_j and B_i) + its destination byte.  Unconditional: padding rows (and batches
        // past the end of the range) read row 0 of a valid object and go to the dummy accumulator row n_cent.
        auto issue = [&](const BatchIt& it, const metav& m, int k, int dbuf) {
            const int gi = it.gi < cnt ? it.gi : cnt - 1;
            const uint32_t g = (uint32_t)(ga + gi);
            const uint32_t sb0 = (uint32_t)it.sb;
            const bool pad = (T2P_ABL & 16) ? false : m[k] == 0xFFFF;
            const uint32_t mm = pad ? 0u : (uint32_t)m[k];
            const uint32_t src = mm & 0xFF, d = mm >> 8, dl = d & 127;
            const uint32_t srow = (d & 0x80) ? (sb0 + src) : (g * (uint32_t)p.n_dense + src);
            {
                // 32-bit BYTE offsets from the uniform table bases (checked by the launcher): SGPR base + VGPR offset loads
                if constexpr (T2P_ABL & 1) {
                    const float f = __uint_as_float((srow * (uint32_t)(K * 4) + c4b) | 0x3f000000u);
                    sa[k] = f32x4{f, f, f, f};
                } else
                    sa[k] = *(const f32x4*)((const char*)p.A + (srow * (uint32_t)(K * 4) + c4b));
                if constexpr (BL) {
                    boff[k] = (pad ? (uint32_t)nc : dl) * (uint32_t)(K * 4) + c4b;
                } else if constexpr (T2P_ABL & 2) {
                    const float f = __uint_as_float(((g * (uint32_t)nc + dl) * (uint32_t)(K * 4) + c4b) | 0x3e000000u);
                    sb[k] = f32x4{f, f, f, f};
                } else
                    sb[k] = *(const f32x4*)((const char*)p.Bc + ((g * (uint32_t)nc + dl) * (uint32_t)(K * 4) + c4b));
            }
            if (c4 == 0) dstl[dbuf * C::TR + rgrp + k] = (uint16_t)((pad ? (uint32_t)nc : dl) * (uint32_t)(N * 4));
        };
        // staging of row k, first half: v = relu(A_j - B_i), hi = fp16(v) to nearest -> hi plane
        auto stage_a = [&](int buf, int k) {
            _Float16* dsth = hidh + buf * 2 * C::PLANE;
            f32x4 t;
            if constexpr (BL) t = sa[k] - bq;
            else t = sa[k] - sb[k];
#pragma unroll
            for (int e = 0; e < 4; e++) vv[e] = fmaxf(t[e], 0.f);
            vh01 = cvt_pk_f16(vv[0], vv[1]);
            vh23 = cvt_pk_f16(vv[2], vv[3]);
            uint2 ph;
            ph.x = __builtin_bit_cast(uint32_t, vh01);
            ph.y = __builtin_bit_cast(uint32_t, vh23);
            *(uint2*)(dsth + (rgrp + k) * C::LDHH + c4 * 4) = ph;
        };
        // second half: lo = fp16(v - hi) -> lo plane (no scale factor: the matrix cores honour fp16 denormals)
        auto stage_b = [&](int buf, int k) {
            _Float16* dsth = hidh + buf * 2 * C::PLANE;
            // v - float(hi) in one VALU op each: v_fma_mix_f32 reads the fp16 half directly (hi * -1 + v, exact)
            const fp16x2 l01 = cvt_pk_f16(sub_half<0>(vv[0], vh01), sub_half<1>(vv[1], vh01));
            const fp16x2 l23 = cvt_pk_f16(sub_half<0>(vv[2], vh23), sub_half<1>(vv[3], vh23));
            uint2 pl;
            pl.x = __builtin_bit_cast(uint32_t, l01);
            pl.y = __builtin_bit_cast(uint32_t, l23);
            *(uint2*)(dsth + C::PLANE + (rgrp + k) * C::LDHH + c4 * 4) = pl;
        };
        auto flush = [&](int64_t g, int abuf) {
            int* a = acc_lds + abuf * C::ACC_INTS;
            float* o = p.out + g * nc * (int64_t)p.ldo;
            int top = gtop;  // fp16-range guard: the largest output (bit patterns of non-negative floats order like ints)
            typedef int i32x4 __attribute__((ext_vector_type(4)));
            for (int i = tid; i < nc * (N / 4); i += NT) {      // 16 bytes per thread and trip
                const int c = i / (N / 4), col = (i % (N / 4)) * 4;
                const f32x4 raw = *(const f32x4*)(a + c * N + col);
                const f32x4 bq = *(const f32x4*)(biasl + col);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float r = fmaxf(raw[e] + bq[e], 0.f);     // (a centroid without rows stays at -inf: 0)
                    const int bits = __float_as_int(r);
                    top = bits > top ? bits : top;
                    v[e] = r * p.out_scale;  // weight image scale (power of 2)
                }
                *(f32x4*)(o + c * (int64_t)p.ldo + col) = v;
                *(i32x4*)(a + c * N + col) = i32x4{(int)0xFF800000, (int)0xFF800000, (int)0xFF800000, (int)0xFF800000};
            }
            // (the next dense kernel splits these rows to fp16: the magnitude goes into the lane's running maximum)
            gtop = top;
        };

        BatchIt it_c{0, 0, (int)nr[0], sbase[0]};
        BatchIt it_s = advance(it_c);
        BatchIt it_g = advance(it_s);
        BatchIt it_m = advance(it_g);
        // prologue: tile 0 staged, gathers of batch 1 in flight, metadata of batch 2 in registers
        load_meta(it_c, meta_g);
        fix_meta(it_c, meta_g);
#pragma unroll
        for (int k = 0; k < IT; k++) issue(it_c, meta_g, k, 0);
        if constexpr (BL) build_b(ga);   // (the sub-range's first barrier pair above ordered any earlier reader of the table)
#pragma unroll
        for (int k = 0; k < IT; k++) {
            load_b(k);
            stage_a(0, k);
            stage_b(0, k);
        }
        load_meta(it_s, meta_g);
        fix_meta(it_s, meta_g);
#pragma unroll
        for (int k = 0; k < IT; k++) issue(it_s, meta_g, k, 1);
        load_meta(it_g, meta_g);
        fix_meta(it_g, meta_g);
        for (int i = tid; i < C::TR; i += NT) dstl[3 * C::TR + i] = (uint16_t)(nc * N * 4);  // "batch -1": all padding
        __syncthreads();

        int64_t flush_g = -1, flush_g1 = -1;
        int flush_buf = 0, flush_buf1 = 0;
        int trace_n = 0;
        (void)trace_n;
        // DEFER (K <= 128, registers to spare): the atomics of batch t-1 ride between the MFMAs of batch t, fed from a
        // copy of its results; one more batch passes before a finished object may be flushed.
        constexpr bool DEFER = K <= 128 || T2P_DEFER256;
        // PINGPONG (= DEFER): two result arrays that swap roles from batch to batch - the MFMAs of batch t write one while the
        // atomics of batch t-1 read the other, the batch loop is unrolled by two - instead of one array and a 16-register copy
        // per batch (which also put an s_nop 11 behind the last MFMA): SA2 -2..3 %, SA1 -5 %.
        constexpr bool PINGPONG = DEFER;
        constexpr int HALVES = PINGPONG ? 2 : 1;
        f32x16 rr[HALVES][DEFER ? RT : 1][DEFER ? C::NTW : 1];
        auto& res = rr[0];
        int prev_abuf = 0;
        if constexpr (DEFER) {
#pragma unroll
            for (int rt = 0; rt < RT; rt++)
#pragma unroll
                for (int nt = 0; nt < C::NTW; nt++)
#pragma unroll
                    for (int e = 0; e < 16; e++) rr[0][rt][nt][e] = rr[HALVES - 1][rt][nt][e] = -__builtin_inff();   // (a no-op maximum)
        }
        // atomics e0 .. e1-1 (flattened over row tile, column tile, accumulator register) of a finished batch
        auto atomics = [&](const f32x16 (&v)[DEFER ? RT : 1][DEFER ? C::NTW : 1], const uint2 (&four)[RT][4], int abuf,
                           int e0, int e1) {
            int* accb = acc_lds + abuf * C::ACC_INTS;
            if constexpr (T2P_ABL & 4) {
                if (e0 == 0) abl_keep(v[0][0], four[0][0].x, accb);
                return;
            }
#pragma unroll
            for (int idx = e0; idx < e1; idx++) {
                const int rt = idx / (C::NTW * 16), nt = (idx / 16) % C::NTW, e = idx % 16;
                const uint32_t pair = (e & 2) ? four[rt][e >> 2].y : four[rt][e >> 2].x;
                const uint32_t off = (e & 1) ? (pair >> 16) : (pair & 0xFFFFu);
                lds_fmax((float*)((char*)(accb + wn * C::NTW * 32 + nt * 32 + l31) + off), v[rt][nt][e]);
            }
        };
        int t_end = 0;
        int newest = 0;   // PINGPONG: the array the last batch filled
        for (int t = 0; valid(it_c);) {
#pragma unroll
          for (int half = 0; half < HALVES; half++, t++) {
            if (half > 0 && !valid(it_c)) break;
            t_end = t + 1;
            STAMP(0);
            {
                bool fence = false;
                if (flush_g >= 0) {  // the object finished in the previous batch drains to HBM
                    flush(flush_g, flush_buf);
                    flush_g = -1;
                    // one accumulator buffer: the next object's atomics (issued inside this batch) must not overtake the drain
                    fence = C::ACC_BUFS == 1;
                }
                if constexpr (BL) {
                    // batch t+1 (staged during this batch) opens a new object: its centroid table replaces the current one;
                    // nobody reads the table right now (batch t was staged in the previous iteration)
                    if (valid(it_s) && it_s.r0 == 0) {
                        build_b(ga + it_s.gi);   // (its barriers also order the drain above)
                        fence = false;
                    }
                }
                if (fence) lds_barrier();   // (LDS only: the drain's global stores need not have landed)
            }
            load_b(0);   // BL: first row of batch t+1; the others follow one staging chunk ahead of their use

            const int buf = t & 1, sbuf = buf ^ 1, dbuf = (t + 2) & 3;
            BatchIt it_n = it_m;
            // ONE accumulator for hi.hi + hi.lo + lo.hi (same scale); it starts at the bias block.  PINGPONG: it IS the result
            // array of this batch
            f32x16 acc_loc[PINGPONG ? 1 : RT][PINGPONG ? 1 : C::NTW];
            f32x16 (*acc)[C::NTW];
            if constexpr (PINGPONG) acc = rr[half ^ 1];
            else acc = acc_loc;
            STAMP(1);
            // destination bytes of this lane's 16 accumulator rows (4 quads of 4 consecutive rows per row tile); written
            // two batches ago, fetched here so that the atomics behind the MFMAs do not start with an LDS round trip
            uint2 four[RT][4];
            constexpr bool HOIST = DEFER;  // K = 256 has no registers to spare (weights alone take 128)
            auto load_four = [&]() {
                const uint16_t* dl = dstl + ((DEFER ? t + 3 : t) & 3) * C::TR;
#pragma unroll
                for (int rt = 0; rt < RT; rt++)
#pragma unroll
                    for (int q = 0; q < 4; q++) four[rt][q] = *(const uint2*)(dl + (wm * RT + rt) * 32 + 8 * q + 4 * h);
            };
            if constexpr (HOIST) load_four();
            const _Float16* hrow = hidh + buf * 2 * C::PLANE + ((wm * RT) * 32 + l31) * C::LDHH + h * (K / 2);
            half8 a_hi = *(const half8*)(hrow), a_lo = *(const half8*)(hrow + C::PLANE), n_hi = a_hi, n_lo = a_lo;
#pragma unroll
            for (int j = 0; j < C::NG; j++) {
                const int s = j / RT, rt = j % RT;
                SB();
                if (j + 1 < C::NG) {
                    const int s2 = (j + 1) / RT, rt2 = (j + 1) % RT;
                    n_hi = *(const half8*)(hrow + rt2 * 32 * C::LDHH + s2 * 8);
                    n_lo = *(const half8*)(hrow + C::PLANE + rt2 * 32 * C::LDHH + s2 * 8);
                }
                __builtin_amdgcn_s_setprio(1);  // a wave inside its MFMA pair wins issue arbitration over its SIMD neighbour's
                                                // staging work (measured: -2 % at K = 256; on the third MFMA too: 0)
#pragma unroll
                for (int nt = 0; nt < C::NTW; nt++) {
                    acc[rt][nt] = MFMA16(a_hi, w_hi[nt][s], s == 0 ? kZero16 : acc[rt][nt]);
                    acc[rt][nt] = MFMA16(a_hi, w_lo[nt][s], acc[rt][nt]);
                }
                __builtin_amdgcn_s_setprio(0);
                SB();
                if (j == 0) load_meta(it_m, meta_m);  // M(t+3); issued behind the first MFMAs so nothing waits on it
                CHUNKS();
                if constexpr (DEFER) {
                    constexpr int TOT = RT * C::NTW * 16;
                    atomics(rr[PINGPONG ? half : 0], four, prev_abuf, (j * TOT) / C::NG, ((j + 1) * TOT) / C::NG);
                }
                SB();
#pragma unroll
                for (int nt = 0; nt < C::NTW; nt++)
                    if constexpr (!(T2P_ABL & 8)) acc[rt][nt] = MFMA16(a_lo, w_hi[nt][s], acc[rt][nt]);

                a_hi = n_hi;
                a_lo = n_lo;
                if (j == C::NG / 2 - 1) {
                    STAMP(2);
                    it_n = advance(it_m);  // iterator bookkeeping (LDS reads of the object table) off the loop's tail
                }
            }
            STAMP(3);
            SB();
            const int abuf = C::ACC_BUFS == 2 ? (it_c.gi & 1) : 0;
            const bool obj_done = it_c.r0 + C::TR >= it_c.n;
            if constexpr (DEFER) {
                if constexpr (PINGPONG) {
                    newest = half ^ 1;
                } else {
#pragma unroll
                    for (int rt = 0; rt < RT; rt++)
#pragma unroll
                        for (int nt = 0; nt < C::NTW; nt++) res[rt][nt] = acc[rt][nt];
                }
                prev_abuf = abuf;
                // the object whose atomics ran in this batch may be flushed next; the one that just ended waits a batch
                flush_g = flush_g1;
                flush_buf = flush_buf1;
                flush_g1 = obj_done ? (int64_t)(ga + it_c.gi) : -1;
                flush_buf1 = abuf;
            } else {
                STAMP(4);
                load_four();
                // max-aggregation: float atomic max into the object's LDS accumulator (bias + ReLU at the drain)
                int* accb = acc_lds + abuf * C::ACC_INTS;
#pragma unroll
                for (int rt = 0; rt < RT; rt++) {
                    const int trow0 = (wm * RT + rt) * 32;
                    if (it_c.r0 + trow0 >= it_c.n) continue;
#pragma unroll
                    for (int nt = 0; nt < C::NTW; nt++) {
                        char* col = (char*)(accb + wn * C::NTW * 32 + nt * 32 + l31);
                        if constexpr (T2P_ABL & 4) {
                            abl_keep(acc[rt][nt], four[rt][0].x, col);
                        } else {
#pragma unroll
                            for (int e = 0; e < 16; e++) {
                                const uint32_t pair = (e & 2) ? four[rt][e >> 2].y : four[rt][e >> 2].x;
                                lds_fmax((float*)(col + ((e & 1) ? (pair >> 16) : (pair & 0xFFFFu))), acc[rt][nt][e]);
                            }
                        }
                    }
                }
                STAMP(5);
                if (obj_done) {
                    flush_g = ga + it_c.gi;
                    flush_buf = abuf;
                }
            }
            fix_meta(it_m, meta_m);
            meta_g = meta_m;
            it_c = it_s;
            it_s = it_g;
            it_g = it_m;
            it_m = it_n;
            STAMP(6);
            __syncthreads();
            STAMP(7);
            trace_n++;
          }
        }
        if (flush_g >= 0) {
            flush(flush_g, flush_buf);
            if constexpr (C::ACC_BUFS == 1) lds_barrier();
        }
        if constexpr (DEFER) {  // drain: atomics of the last batch, then its object
            uint2 four[RT][4];
            const uint16_t* dl = dstl + ((t_end + 3) & 3) * C::TR;
#pragma unroll
            for (int rt = 0; rt < RT; rt++)
#pragma unroll
                for (int q = 0; q < 4; q++) four[rt][q] = *(const uint2*)(dl + (wm * RT + rt) * 32 + 8 * q + 4 * h);
            if (PINGPONG && newest == 1) atomics(rr[HALVES - 1], four, prev_abuf, 0, RT * C::NTW * 16);
            else atomics(rr[0], four, prev_abuf, 0, RT * C::NTW * 16);
            __syncthreads();
            if (flush_g1 >= 0) flush(flush_g1, flush_buf1);
        }
    }
    uint32_t gbits = 0;
    guard_track_bits(gbits, gtop);
    if (p.amax_out != nullptr && lane == 0 && gbits != 0u)
        atomicMax(p.amax_out, __float_as_uint(__uint_as_float(gbits) * p.out_scale));
}

template <int K, int N, int WN, int RT, int NW, int WGS, int BL = 0>
int launch_cfg2(const SaParams& p, hipStream_t st, const char* name) {
    using C = Cfg2<K, N, WN, RT, NW, WGS, BL>;
    static_assert(WGS == 1 || K <= 128, "two workgroups per CU rely on the deferred atomics (K <= 128)");
    static_assert(C::lds_bytes() * WGS <= 160 * 1024, "LDS budget");
    auto kern = k_ws_sa2<K, N, WN, RT, NW, WGS, BL>;
    if (BL) T2P_CHECK_ARG(p.wp != nullptr && p.n_cent == 8192 / K && ((uintptr_t)p.out & 3) == 0,
                          "ws_sa2: the LDS centroid table needs wp and n_cent = %d (got %d)", 8192 / K, p.n_cent);
    T2P_TRY(reserve_lds((const void*)kern, C::lds_bytes(), "ws_sa2"));
    if (p.n_obj <= 0) return 0;
    T2P_CHECK_ARG(((uintptr_t)p.out & 15) == 0 && p.ldo % 4 == 0, "ws_sa2: output rows must be 16-byte aligned (ldo = %d)", p.ldo);
    T2P_CHECK_ARG(p.n_obj < (1 << 30) && p.n_obj * p.n_dense * (int64_t)K * 4 < 0xffffffffLL &&
                      p.n_obj * p.n_cent * (int64_t)K * 4 < 0xffffffffLL,
                  "ws_sa: chunk too large for 32-bit table offsets");
    int n_wg = num_cus() * WGS;
    if (n_wg > 1024) n_wg = 1024;
    if (n_wg > p.n_obj) n_wg = (int)p.n_obj;
    if (!p.balanced) {
        int rc = launch_sa_balance(p, C::TR, n_wg, st);
        if (rc != 0) return rc;
    }
    ProfScope ps_(name, st);
    hipLaunchKernelGGL(kern, dim3(n_wg), dim3(C::NT), C::lds_bytes(), st, p);
    T2P_CHECK_LAUNCH("ws_sa2");
#if T2P_TRACE
    {
        static int printed = 0;
        if (!printed && K == T2P_TRACE) {
            printed = 1;
            hipStreamSynchronize(st);
            int h[96 * 8];
            hipMemcpy(h, p.prefix_ws, sizeof(h), hipMemcpyDeviceToHost);
            double sum[8] = {0};
            for (int b = 8; b < 88; b++)
                for (int i = 0; i < 8; i++) sum[i] += (double)(unsigned)(h[b * 8 + (i + 1) % 8 + (i == 7 ? 8 : 0)] - h[b * 8 + i]);
            fprintf(stderr, "[trace %s] mean s_memtime ticks per segment over 80 batches:", name);
            for (int i = 0; i < 8; i++) fprintf(stderr, " s%d->%d=%.0f", i, (i + 1) % 8, sum[i] / 80);
            fprintf(stderr, "\n");
            for (int b = 8; b < 14; b++) {
                fprintf(stderr, "[trace] batch %d:", b);
                for (int i = 0; i < 7; i++) fprintf(stderr, " %u", (unsigned)(h[b * 8 + i + 1] - h[b * 8 + i]));
                fprintf(stderr, " | next %u\n", (unsigned)(h[(b + 1) * 8] - h[b * 8 + 7]));
            }
        }
    }
#endif
    return 0;
}

}  // namespace

// (tile rows, workgroups) of the configuration launch_ws_sa2 uses for (H, Cout): keep in step with the table below
int sa2_launch_shape(int H, int Cout, int64_t n_obj, int* tile_rows, int* n_wg) {
    int tr, wgs;
    if (H == 32 && Cout == 64) { tr = Cfg2<32, 64, 2, 1, 8, 2>::TR; wgs = 2; }
    else if (H == 128 && Cout == 128) { tr = Cfg2<128, 128, 4, 1, 8, 1>::TR; wgs = 1; }
    else if (H == 256 && Cout == 256) { tr = Cfg2<256, 256, 8, 1, 8, 1>::TR; wgs = 1; }
    else {
        set_error("ws_sa2: no instantiation for H=%d C=%d", H, Cout);
        return T2P_E_UNSUPPORTED;
    }
    int n = num_cus() * wgs;
    if (n > 1024) n = 1024;
    if (n > n_obj) n = (int)n_obj;
    *tile_rows = tr;
    *n_wg = n;
    return 0;
}

int launch_ws_sa2(int H, int Cout, const SaParams& p, hipStream_t st) {
    // SA2: two 4-wave workgroups per CU measured no faster than one 8-wave workgroup (kept: its double-buffered accumulator
    // drains under the MFMAs); SA3 cannot split (128 weight registers per 32 columns).
    // SA1: the weights take 16 registers, so 16 waves per CU fit (two 8-wave workgroups of <= 128 registers, 2 rows per
    // thread): 4 waves per SIMD measured 10 % faster than 2 (two 4-wave workgroups) and 14 % faster than one 8-wave one
    if (H == 32 && Cout == 64) return launch_cfg2<32, 64, 2, 1, 8, 2>(p, st, "ws_edge_sa_k32_n64");
    // wp given: the centroid table is built in LDS per object (BL); otherwise gathered from the Bc table in HBM
    if (H == 128 && Cout == 128 && p.wp) return launch_cfg2<128, 128, 4, 1, 8, 1, 1>(p, st, "ws_edge_sa_k128_n128");
    if (H == 256 && Cout == 256 && p.wp) return launch_cfg2<256, 256, 8, 1, 8, 1, 1>(p, st, "ws_edge_sa_k256_n256");
    if (H == 128 && Cout == 128) return launch_cfg2<128, 128, 4, 1, 8, 1>(p, st, "ws_edge_sa_k128_n128");
    if (H == 256 && Cout == 256) return launch_cfg2<256, 256, 8, 1, 8, 1>(p, st, "ws_edge_sa_k256_n256");
    set_error("ws_sa2: no instantiation for H=%d C=%d", H, Cout);
    return T2P_E_UNSUPPORTED;
}

}  // namespace t2p

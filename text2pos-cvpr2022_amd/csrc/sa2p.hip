// Set-abstraction level 2 edge kernel (K = N = 128), f16x3 path, in the organisation of sa3.hip (round 6 experiment).
// (reference: gnn.PointConv(local_nn)(x, (pos, pos[idx]), edge_index), models/pointcloud/pointnet2.py:31-35, sa2 = get_mlp([67, 128, 128]) :58).
//
// sa_rows.hip gives SA2 one wave per SIMD that holds the whole weight matrix and owns rows; everything that is not an MFMA comes out of
// that wave's MFMA stream (46 % MFMA-busy).  Here SA2 runs the way SA3 does: 8 waves per CU = two per SIMD, all waves share one staged
// batch, a wave's staging work and atomics ride under its partner's MFMAs.  What K = 128 changes against sa3.hip:
//   * a batch is 64 rows; wave w = (row group rg = w / 4, column group cg = w % 4) multiplies batch rows [32 rg, +32) by output columns
//     [32 cg, +32): 2 x 2 blocks of 16 x 16, four k-steps of 32, 48 MFMAs per wave and batch, 64 weight registers; a staged row is read
//     by the four waves of its row group only (LDS operand traffic per MFMA as at K = 256);
//   * a staged row is 512 bytes = HALF a wave-wide 16-byte load: a wave stages row PAIRS (batch rows 8 w + 2 k, + 1 in lanes 0-31 /
//     32-63).  The pair's metadata is still decoded on the scalar unit (one s_load_dwordx4 per wave and batch = 8 u16 entries); the
//     two rows' table offsets reach the lanes through one v_cndmask each (gather offset, centroid-table offset);
//   * 16 slots per batch (4 steps x 4 blocks): the 12 staging chunks ride in the first 12, one deferred atomic in every slot.
#include "t2p_common.h"

#ifndef T2P_SA2P_ABL
#define T2P_SA2P_ABL 0
#endif
#define SB() __builtin_amdgcn_sched_barrier(0)
#define AS4 __attribute__((address_space(4)))

namespace t2p {
namespace {

constexpr int K = 128, N = 128, NC = 64, ND = 128, TR = 64, NW = 8, NT = 64 * NW;
constexpr int LDHH = K + 8;            // halves per plane row (16-byte pad: conflict-free ds_read_b128)
constexpr int PLANE = TR * LDHH;       // halves per plane
constexpr int S16 = K / 16, S32 = K / 32;
constexpr int MAXR = NC * 33;          // row-list slots per object

// LDS map (bytes)
constexpr int OFF_TILE = 0;                              // [2 buffers][hi plane | lo plane]
constexpr int OFF_ACC = OFF_TILE + 2 * 2 * PLANE * 2;    // [NC + 1][N] fp32 running maxima (row NC: padding rows)
constexpr int OFF_BTAB = OFF_ACC + (NC + 1) * N * 4;     // [NC + 1][K] fp32 centroid table (row NC: zeros)
constexpr int OFF_WP = OFF_BTAB + (NC + 1) * K * 4;      // [3][K] position rows of the layer-1 weights
constexpr int OFF_CPOS = OFF_WP + 3 * K * 4;             // [NC][3] centroid positions of the object being built
constexpr int OFF_BIAS = OFF_CPOS + 1024;                // [N] bias x weight scale
constexpr int OFF_DST = OFF_BIAS + N * 4;                // [4 batches][TR] u16 accumulator byte offset of every staged row
constexpr int LDS_BYTES = OFF_DST + 4 * TR * 2;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
static_assert((NC + 1) * N * 4 < 65536, "accumulator byte offsets travel as u16");

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
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct It {      // one pipeline stage's position in the workgroup's object range; every member is wave-uniform (SGPRs)
    int g;       // object
    int r0;      // first row of the batch inside the object
    int n;       // rows of the object (0 past the end of the range)
    int sb;      // self-loop base: table row of dense point 0 of the object's cell batch + n_cent * (object's rank in the cell)
};

__global__ __launch_bounds__(NT, 2) void k_sa2p(SaParams p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    _Float16* const tile = (_Float16*)(lds + OFF_TILE);
    float* const accl = (float*)(lds + OFF_ACC);
    float* const btab = (float*)(lds + OFF_BTAB);
    float* const wpl = (float*)(lds + OFF_WP);
    float* const cposl = (float*)(lds + OFF_CPOS);
    float* const biasl = (float*)(lds + OFF_BIAS);
    uint16_t* const dstl = (uint16_t*)(lds + OFF_DST);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = wave & 3, rg = wave >> 2;       // column group (32 output columns), row group (32 batch rows)
    const int hf = lane >> 5, l32 = lane & 31;     // staging: row of the pair, column quad (columns 4 l32 .. + 3)

    const AS4 uint32_t* const n_rows_c = as_const((const uint32_t*)p.n_rows);
    const AS4 int32_t* const first_c = as_const(p.first);
    const AS4 uint32_t* const rows_c = as_const((const uint32_t*)p.rows);
    const AS4 int32_t* const bounds_c = as_const(p.bounds_ws);

    // ---- stationary weights: columns [32 cg, +32), all K, hi / lo planes (packing.py::pack_f16x3 order) ----
    // 16x16x32 operands: lane = 16 q + i holds k = 32 q + 8 s + e of MFMA step s (s < 4), column 16 cb + i of the wave's 32
    const int q16 = lane >> 4, i16 = lane & 15;
    half8 w_hi[2][S32], w_lo[2][S32];
    {
        const uint4* wp = (const uint4*)p.W_x3;
        constexpr int PLANE_U4 = (N / 32) * S16 * 64;
#pragma unroll
        for (int cb = 0; cb < 2; cb++)
#pragma unroll
            for (int s = 0; s < S32; s++) {
                const int k0 = 32 * q16 + 8 * s;                       // image: k = half' K/2 + 8 step' + e
                const int idx = ((cg * S16 + (k0 % (K / 2)) / 8) * 2 + k0 / (K / 2)) * 32 + 16 * cb + i16;
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
    float npos = 0.f;
    auto fetch_pos = [&](int g) {
        if (tid < 3 * NC && g < g_end) npos = p.out[((int64_t)g * NC + tid / 3) * (int64_t)p.ldo + N + tid % 3];
    };
    auto build_b = [&](int g) {                  // ends with a barrier
        if (tid < 3 * NC) cposl[tid] = npos;
        fetch_pos(g + 1);
        lds_barrier();
        const f32x4 w0 = *(const f32x4*)(wpl + l32 * 4), w1 = *(const f32x4*)(wpl + K + l32 * 4), w2 = *(const f32x4*)(wpl + 2 * K + l32 * 4);
#pragma unroll
        for (int i = 0; i < NC / (2 * NW); i++) {
            const int c = 2 * (wave + NW * i) + hf;      // a half-wave per centroid row (K / 4 = 32 column quads)
            const float px = cposl[3 * c], py = cposl[3 * c + 1], pz = cposl[3 * c + 2];
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; e++) {        // same order as k_sample_group's table: ((x w0) + y w1) + z w2
                float a = px * w0[e];
                a = fmaf(py, w1[e], a);
                a = fmaf(pz, w2[e], a);
                v[e] = a;
            }
            *(f32x4*)(btab + c * K + l32 * 4) = v;
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

    // ---- per-pair state of the staging pipeline --------------------------------------------------------------------
    f32x4 sa[4];         // gathered A_j row pieces of the batch being staged: pair k = batch rows 8 w + 2 k (lanes 0-31), + 1 (lanes 32-63)
    f32x4 bq;            // centroid-table entries of the pair staged next
    uint32_t boff[4];    // per lane: byte offset of the pair's centroid row inside the LDS table (+ the lane's column quad)
    f32x4 vv;
    fp16x2 vh01, vh23;
    const uint32_t lane16 = (uint32_t)l32 * 16u;
    const bool lo_half = hf == 0;

    // row metadata of a batch for THIS wave: 8 u16 entries = one 16-byte scalar load
    auto load_meta = [&](const It& it) -> uint4 {
        uint4 m{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        if (valid(it)) {
            const uint32_t e = (uint32_t)it.g * (uint32_t)MAXR + (uint32_t)(it.r0 + 8 * wave);   // (multiple of 8: 16-byte aligned)
            m.x = rows_c[e / 2];
            m.y = rows_c[e / 2 + 1];
            m.z = rows_c[e / 2 + 2];
            m.w = rows_c[e / 2 + 3];
        }
        return m;
    };
    // gather of pair k of a batch + its centroid offsets; the 8 destination offsets of the wave go to the ring in one store
    auto issue = [&](const It& it, const uint4& m, int k, uint4& dpack) {
        const uint32_t word = k == 0 ? m.x : (k == 1 ? m.y : (k == 2 ? m.z : m.w));
        uint32_t srow[2], dl[2];
#pragma unroll
        for (int r = 0; r < 2; r++) {          // (scalar unit)
            const uint32_t e16 = r ? (word >> 16) : (word & 0xFFFFu);
            const bool pad = !valid(it) || it.r0 + 8 * wave + 2 * k + r >= it.n;
            const uint32_t mm = pad ? 0u : e16;
            const uint32_t src = mm & 0xFFu, d = mm >> 8;
            dl[r] = pad ? (uint32_t)NC : (d & 127u);
            const uint32_t g = (uint32_t)(valid(it) ? it.g : g_end - 1);
            srow[r] = (d & 0x80u) ? ((uint32_t)it.sb + src) : (g * (uint32_t)ND + src);
        }
        const uint32_t voff = (lo_half ? srow[0] : srow[1]) * (uint32_t)(K * 4) + lane16;     // < 2^32: checked at launch
        if constexpr (T2P_SA2P_ABL & 2) {
            const float f = __uint_as_float((voff & 0xFFFFu) | 0x3f000000u);
            sa[k] = f32x4{f, f, f, f};
        } else
            sa[k] = *(const f32x4*)((const char*)p.A + (size_t)voff);
        boff[k] = (lo_half ? dl[0] : dl[1]) * (uint32_t)(K * 4) + lane16;
        const uint32_t dv = (dl[0] * (uint32_t)(N * 4)) | ((dl[1] * (uint32_t)(N * 4)) << 16);
        if (k == 0) dpack.x = dv;
        else if (k == 1) dpack.y = dv;
        else if (k == 2) dpack.z = dv;
        else dpack.w = dv;
    };
    auto load_b = [&](int k) { bq = *(const f32x4*)((const char*)btab + boff[k]); };
    const uint32_t stage_row = (uint32_t)(8 * wave + hf);      // + 2 k
    auto stage_a = [&](int buf, int k) {         // v = relu(A_j - B_i), hi = fp16(v) -> hi plane
        _Float16* dsth = tile + buf * 2 * PLANE;
        const f32x4 t = sa[k] - bq;
#pragma unroll
        for (int e = 0; e < 4; e++) vv[e] = fmaxf(t[e], 0.f);
        vh01 = cvt_pk_f16(vv[0], vv[1]);
        vh23 = cvt_pk_f16(vv[2], vv[3]);
        uint2 ph;
        ph.x = __builtin_bit_cast(uint32_t, vh01);
        ph.y = __builtin_bit_cast(uint32_t, vh23);
        if constexpr (T2P_SA2P_ABL & 4) asm volatile("" ::"v"(ph.x), "v"(ph.y));
        else *(uint2*)(dsth + (stage_row + 2 * k) * LDHH + l32 * 4) = ph;
    };
    auto stage_b = [&](int buf, int k) {         // lo = fp16(v - hi) -> lo plane
        _Float16* dsth = tile + buf * 2 * PLANE;
        uint2 pl;
        pl.x = split_lo_pk(vh01, vv[0], vv[1]);
        pl.y = split_lo_pk(vh23, vv[2], vv[3]);
        if constexpr (T2P_SA2P_ABL & 4) asm volatile("" ::"v"(pl.x), "v"(pl.y));
        else *(uint2*)(dsth + PLANE + (stage_row + 2 * k) * LDHH + l32 * 4) = pl;
    };
    auto put_dst = [&](int slot, const uint4& d) {
        if (lane == 0) *(uint4*)(dstl + slot * TR + 8 * wave) = d;
    };

    if (g_begin >= g_end) return;   // (uniform)

    // ---- prologue: tile 0 staged, gathers of batch 1 in flight, metadata of batch 2 in registers --------------------
    It it_c{g_begin, 0, obj_n(g_begin), obj_sb(g_begin)};
    It it_s = advance(it_c);
    It it_g = advance(it_s);
    It it_m = advance(it_g);
    fetch_pos(g_begin);
    uint4 dpk{0, 0, 0, 0};
    {
        const uint4 m0 = load_meta(it_c);
#pragma unroll
        for (int k = 0; k < 4; k++) issue(it_c, m0, k, dpk);
    }
    __syncthreads();                 // LDS initialisation above
    put_dst(0, dpk);
    build_b(g_begin);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        load_b(k);
        stage_a(0, k);
        stage_b(0, k);
    }
    {
        const uint4 m1 = load_meta(it_s);
#pragma unroll
        for (int k = 0; k < 4; k++) issue(it_s, m1, k, dpk);
        put_dst(1, dpk);
    }
    uint4 meta_g = load_meta(it_g);
    __syncthreads();

    int flush_g = -1, flush_g1 = -1;
    // deferred atomics: the maxima of batch t-1 ride between the MFMAs of batch t, fed from the result array batch t-1 filled;
    // two result arrays swap roles from batch to batch (the loop is unrolled by two)
    f32x16 rr[2];
#pragma unroll
    for (int e = 0; e < 16; e++) rr[0][e] = rr[1][e] = -__builtin_inff();
    float* const acc_col = accl + cg * 32 + i16;       // this lane's column (of column block 0) in accumulator row 0

    int t = 0;
    int newest = 0;
    for (; valid(it_c);) {
#pragma unroll
        for (int half = 0; half < 2; half++, t++) {
            if (half > 0 && !valid(it_c)) break;
            {
                bool fence = false;
                if (flush_g >= 0 && !(T2P_SA2P_ABL & 128)) {     // the object whose last atomics ran in the previous batch drains
                    flush(flush_g);
                    flush_g = -1;
                    fence = true;                       // the next object's atomics (inside this batch) must not overtake the drain
                }
                if (valid(it_s) && it_s.r0 == 0 && !(T2P_SA2P_ABL & 128)) {   // batch t+1 opens a new object: its centroid table replaces the current one
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
            // result element e = 8 rb + 4 cb + v of this lane: batch row 32 rg + 16 rb + 4 q16 + v, column 16 cb + i16 of the wave's 32
            uint2 four[2];   // destination offsets of those rows of batch t-1 (written three batches ago)
            {
                const uint16_t* dl = dstl + ((t + 3) & 3) * TR + 32 * rg;
#pragma unroll
                for (int rb = 0; rb < 2; rb++) four[rb] = *(const uint2*)(dl + 16 * rb + 4 * q16);
            }
            const _Float16* hrow = tile + buf * 2 * PLANE + (32 * rg + i16) * LDHH + q16 * 32;
            half8 a_hi[2], a_lo[2];
#pragma unroll
            for (int rb = 0; rb < 2; rb++) {
                a_hi[rb] = *(const half8*)(hrow + rb * 16 * LDHH);
                a_lo[rb] = *(const half8*)(hrow + PLANE + rb * 16 * LDHH);
            }
            uint4 meta_m{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
            typedef float f32x4v __attribute__((ext_vector_type(4)));
            const f32x4v kZero4 = {0.f, 0.f, 0.f, 0.f};
            f32x4v blk[4];   // [2 rb + cb]
#pragma unroll
            for (int sl = 0; sl < 4 * S32; sl++) {
                const int s32 = sl >> 2, b = sl & 3, rb = b >> 1, cb = b & 1;
                SB();
                // operands of the NEXT step: row block 1 - rb is idle while block rb multiplies (no second operand set)
                if (s32 + 1 < S32 && !(T2P_SA2P_ABL & 8)) {
                    if (b == 2) {
                        a_hi[0] = *(const half8*)(hrow + (s32 + 1) * 8);
                        a_lo[0] = *(const half8*)(hrow + PLANE + (s32 + 1) * 8);
                    }
                }
                if (s32 > 0 && b == 0 && !(T2P_SA2P_ABL & 8)) {
                    a_hi[1] = *(const half8*)(hrow + 16 * LDHH + s32 * 8);
                    a_lo[1] = *(const half8*)(hrow + PLANE + 16 * LDHH + s32 * 8);
                }
                blk[b] = MFMA32(a_hi[rb], w_hi[cb][s32], s32 == 0 ? kZero4 : blk[b]);
                blk[b] = MFMA32(a_hi[rb], w_lo[cb][s32], blk[b]);
                blk[b] = MFMA32(a_lo[rb], w_hi[cb][s32], blk[b]);
                SB();
                {
                    const int w = sl;
                    if (w == 0) meta_m = load_meta(it_m);          // M(t+3)
                    // the 12 staging chunks ride in the first 12 of the 16 slots, one deferred atomic in every slot
                    if (w < 12) {
                        const int k = w / 3, part = w % 3;
                        if (part == 0) stage_a(sbuf, k);
                        else if (part == 1) stage_b(sbuf, k);
                        else {
                            issue(it_g, meta_g, k, dpk);
                            if (k + 1 < 4) load_b(k + 1);
                            else put_dst(dslot, dpk);
                        }
                    }
                    {
                        const int e = w, erb = e >> 3, ecb = (e >> 2) & 1, v = e & 3;
                        const uint32_t pair = (v & 2) ? four[erb].y : four[erb].x;
                        const uint32_t off = (v & 1) ? (pair >> 16) : (pair & 0xFFFFu);
                        if constexpr (T2P_SA2P_ABL & 1) asm volatile("" ::"v"(prev[e]), "v"(off));
                        else lds_fmax((float*)((char*)acc_col + off) + 16 * ecb, prev[e]);
                    }
                    if (w == 2 * S32 - 1) it_n = advance(it_m);
                }
            }
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
            if constexpr (!(T2P_SA2P_ABL & 32)) __syncthreads();
        }
    }
    if (flush_g >= 0) {
        flush(flush_g);
        lds_barrier();
    }
    {   // drain: atomics of the last batch, then its object
        const uint16_t* dl = dstl + ((t + 3) & 3) * TR + 32 * rg;
        const f32x16& last = rr[newest];
        uint2 four[2];
#pragma unroll
        for (int rb = 0; rb < 2; rb++) four[rb] = *(const uint2*)(dl + 16 * rb + 4 * q16);
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int erb = e >> 3, ecb = (e >> 2) & 1, v = e & 3;
            const uint32_t pair = (v & 2) ? four[erb].y : four[erb].x;
            const uint32_t off = (v & 1) ? (pair >> 16) : (pair & 0xFFFFu);
            lds_fmax((float*)((char*)acc_col + off) + 16 * ecb, last[e]);
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

// T2P_SA2_PAIRS=1 routes SA level 2 of the f16x3 path here instead of sa_rows.hip (A/B switch; see the measurements in docs/notebook.md)
#ifndef T2P_SA2_PAIRS
#define T2P_SA2_PAIRS 0
#endif
bool sa2p_selected(int H, int C, const SaParams& p) {
    return T2P_SA2_PAIRS && H == 128 && C == 128 && p.W_x3 != nullptr && p.wp != nullptr;
}

int sa2p_launch_shape(int64_t n_obj, int* tile_rows, int* n_wg) {
    int n = matrix_wgs();
    if (n > 1024) n = 1024;
    if (n > n_obj) n = (int)n_obj;
    *tile_rows = TR;
    *n_wg = n;
    return 0;
}

int launch_sa2p(const SaParams& p, hipStream_t st) {
    T2P_CHECK_ARG(p.n_cent == NC && p.n_dense == ND, "sa2p: built for %d dense points / %d centroids per object (got %d / %d)", ND, NC,
                  p.n_dense, p.n_cent);
    T2P_CHECK_ARG(p.balanced, "sa2p: needs the balanced object ranges of launch_sa_balance_levels");
    T2P_TRY(reserve_lds((const void*)k_sa2p, LDS_BYTES, "sa2p"));
    if (p.n_obj <= 0) return 0;
    T2P_CHECK_ARG(((uintptr_t)p.out & 15) == 0 && p.ldo % 4 == 0 && ((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.rows & 15) == 0 &&
                      ((uintptr_t)p.n_rows & 3) == 0,
                  "sa2p: tables and output rows must be 16-byte aligned (ldo = %d)", p.ldo);
    T2P_CHECK_ARG(p.n_obj * (int64_t)ND * (int64_t)(K * 4) < (1LL << 32), "sa2p: chunk too large for 32-bit table offsets");
    int tr, n_wg;
    T2P_TRY(sa2p_launch_shape(p.n_obj, &tr, &n_wg));
    ProfScope ps_("ws_edge_sa_k128_n128", st);
    T2P_REPEAT(ps_) hipLaunchKernelGGL(k_sa2p, dim3(n_wg), dim3(NT), LDS_BYTES, st, p);
    T2P_CHECK_LAUNCH("sa2p");
    return 0;
}

}  // namespace t2p

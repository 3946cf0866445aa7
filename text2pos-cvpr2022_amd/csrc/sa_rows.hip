// Set-abstraction edge kernel, f16x3 path, ROW-OWNING waves (SA level 2: H = C = 128).
// (reference: gnn.PointConv(local_nn)(x, (pos, pos[idx]), edge_index), models/pointcloud/pointnet2.py:31-35).
//
// ws_sa2.hip gives every wave a 32-column slice of the layer-2 weights and lets all waves of the workgroup share one
// staged row batch (gather -> registers -> ReLU(A_j - B_i) -> fp16 split -> LDS planes -> barrier -> operand reads): a wave
// then issues ~500 instructions around 24 MFMAs per batch at K = 128 and the matrix pipe idles two thirds of the time.
// Here a wave owns ROWS instead: it holds the WHOLE 128 x 128 weight matrix (hi + lo fp16 planes: 256 registers; one wave
// per SIMD, 512-register budget), fetches its own 32-row tiles with LDS-DMA (global_load_lds, no staging registers, no
// ds_write pass), reads each row piece straight into the MFMA A-operand layout, forms ReLU(A_j - B_i) and its fp16 hi / lo
// split in registers and multiplies 32 rows by all 128 columns: 96 MFMAs per tile and wave, no workgroup barrier inside an
// object, ~5 other instructions per MFMA instead of ~20.
//
//   * one workgroup (4 waves) per CU walks a balanced contiguous object range; the tiles of an object go round-robin to
//     the waves, the object's max-accumulator [n_cent][C] and centroid table B_i = W1p pos_i [n_cent][H] live in LDS;
//   * ring of 4 slots x 4 KB per wave = one whole tile of prefetch: slot u holds the 128-byte pieces (k = 32 u .. 32 u + 31)
//     of the tile's 32 rows, 8 rows x 128 B per DMA instruction (full cache lines), XOR-swizzled on the SOURCE address so
//     that the lane-linear LDS image is read conflict-free with ds_read_b128; counted s_waitcnt vmcnt(12) (three slots stay
//     in flight), never vmcnt(0) inside the stream;
//   * everything a wave loads travels by LDS-DMA (row lists and centroid positions of the objects ahead included): an
//     ordinary VGPR load beside outstanding DMAs would make hipcc drain the whole ring at its first use;
//   * natural k order (lane half h owns k = 16 s + 8 h .. + 7 of MFMA step s): the host's register-order weight image is
//     re-indexed at load time; every fp32 accumulation runs hi.hi, hi.lo, lo.hi per step like ws_sa2.hip, with another
//     grouping of the k's (results agree to fp32 rounding, not bit for bit).
#include "t2p_common.h"

namespace t2p {
int launch_sa_balance(const SaParams& p, int tile_rows, int n_wg, hipStream_t st);  // ws_sa.hip

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gl_void;

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)

constexpr int kSubR = 1024;   // objects of a workgroup's range cached at a time (row counts, self-loop bases)

template <int K, int N, int NC, int NW>
struct RowsCfg {
    static constexpr int NT = 64 * NW;
    static constexpr int ND = 2 * NC;
    static constexpr int S16 = K / 16, NTW = N / 32;
    static constexpr int SLOTS = K / 32;                 // 128-byte row pieces (two MFMA steps each) = ring slots per tile
    static constexpr int SLOT_BYTES = 32 * 128;
    static constexpr int RING_BYTES = SLOTS * SLOT_BYTES;   // one tile per wave
    static constexpr int MAXR = NC * 33;
    static constexpr int ROWS_CHUNKS = (MAXR * 2 + 1023) / 1024;
    static constexpr int ROWS_BUF = ROWS_CHUNKS * 1024;  // the row list arrives in whole 1 KB DMA pieces
    static constexpr int BT_STRIDE = K * 4 + 16;         // centroid-table row pitch: +16 B keeps ds_read_b128 conflict-free
    static constexpr int CPOS_BUF = 3 * NC * 4;
    // LDS map (bytes)
    static constexpr int ACC_OFF = 0;
    static constexpr int BT_OFF = ACC_OFF + NC * N * 4;
    static constexpr int ROWS_OFF = BT_OFF + NC * BT_STRIDE;
    static constexpr int CPOS_OFF = ROWS_OFF + 3 * ROWS_BUF;
    static constexpr int NR_OFF = CPOS_OFF + 2 * CPOS_BUF;
    static constexpr int SB_OFF = NR_OFF + kSubR * 2;
    static constexpr int DSTL_OFF = SB_OFF + kSubR * 4;
    static constexpr int RING_OFF = (DSTL_OFF + NW * 64 + 1023) / 1024 * 1024;
    static constexpr size_t lds_bytes() { return (size_t)RING_OFF + (size_t)NW * RING_BYTES; }
    static_assert(K % 32 == 0 && N % 32 == 0 && NC % 64 == 0 && (NC * N / 4) % NT == 0, "shape");
    static_assert(SLOTS == 4, "the counted waits below assume four ring slots per tile (K = 128)");
};

template <int SEL>
__device__ __forceinline__ float sub_half_r(float v, fp16x2 h) {   // v - (float)h[SEL] in one VALU op (exact)
    float r;
    if constexpr (SEL == 0)
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(v));
    else
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(v));
    return r;
}

// workgroup barrier that orders LDS traffic only (no vmcnt: outstanding LDS-DMA pieces of the ring must survive it)
__device__ __forceinline__ void lds_barrier_r() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// counted wait for the ring: the 12 youngest DMA instructions (three slots) may stay in flight
__device__ __forceinline__ void wait_ring() { asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }
__device__ __forceinline__ void wait_all_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

struct TileRef {   // a tile of this wave: object (index inside the cached sub-range), first row, rows of the object
    int gi, r0, n;
};

template <int K, int N, int NC, int NW>
__global__ __launch_bounds__(64 * NW, 1) void k_sa_rows(SaParams p) {
    using C = RowsCfg<K, N, NC, NW>;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    int* acc_lds = (int*)(lds + C::ACC_OFF);
    uint16_t* nr = (uint16_t*)(lds + C::NR_OFF);
    int* sbase = (int*)(lds + C::SB_OFF);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, rr = lane & 31;
    const uint32_t ringb = (uint32_t)(C::RING_OFF + wave * C::RING_BYTES);
    const uint32_t dstl_addr = (uint32_t)(C::DSTL_OFF + wave * 64);   // [32] u16: accumulator-row byte offset of every tile row

    // ---- stationary weights: the whole [K][N] matrix as hi / lo fp16 planes, natural k order ---------------------------
    // image (packing.py::pack_f16x3_scaled): [plane][n-tile][step'][half'][32 lanes][8 halves] with k = half' K/2 + 8 step' + e;
    // this lane's operand of step s covers k = 16 s + 8 h + e.
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
    float biasv[C::NTW];
#pragma unroll
    for (int nt = 0; nt < C::NTW; nt++) biasv[nt] = p.bias[nt * 32 + rr];
    // position rows of the layer-1 weights, this thread's column quad (centroid table build)
    constexpr int QPR = K / 4;                 // column quads per table row
    constexpr int CGS = C::NT / QPR;           // centroid groups of the workgroup
    constexpr int CPT = NC / CGS;              // centroids per thread
    static_assert(C::NT % QPR == 0 && NC % CGS == 0, "centroid table build");
    const int cq = tid % QPR, cg = tid / QPR;
    const f32x4 wq0 = *(const f32x4*)(p.wp + cq * 4), wq1 = *(const f32x4*)(p.wp + K + cq * 4),
                wq2 = *(const f32x4*)(p.wp + 2 * K + cq * 4);

    // ---- per-lane constants --------------------------------------------------------------------------------------------
    // ring reads: row rr of a slot is 128 B = 8 chunks of 16 B; chunk c sits at position c ^ ((rr >> 1) & 7)
    uint32_t rd[2][2];
#pragma unroll
    for (int par = 0; par < 2; par++)
#pragma unroll
        for (int j = 0; j < 2; j++)
            rd[par][j] = ringb + (uint32_t)(rr * 128 + (((4 * par + 2 * h + j) ^ ((rr >> 1) & 7)) * 16));
    // DMA instruction q of a slot: lane i fetches row 8 q + (i >> 3), LDS position i & 7 <- source chunk (i & 7) ^ swizzle(row)
    uint32_t dma_sel[4], dma_chunk[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int r = 8 * q + (lane >> 3);
        dma_sel[q] = (uint32_t)(r * 4);
        dma_chunk[q] = (uint32_t)((((lane & 7) ^ ((r >> 1) & 7))) * 16);
    }

    for (int i = tid; i < NC * N; i += C::NT) acc_lds[i] = 0;
    uint32_t gbits = 0;   // fp16-range guard: wave-uniform maximum (bit pattern, before out_scale) of the drained outputs

    const int g_begin = p.bounds_ws[blockIdx.x], g_end = p.bounds_ws[blockIdx.x + 1];

    // ---- DMA helpers -----------------------------------------------------------------------------------------------------
    auto dma16 = [&](const void* base, uint32_t voff, uint32_t lds_off) {
        __builtin_amdgcn_global_load_lds((gl_void*)((const char*)base + voff), (lds_void*)(lds + lds_off), 16, 0, 0);
    };
    // row list of object g (absolute) -> rows buffer g % 3, in 1 KB pieces split over the waves
    auto dma_rows = [&](int g) {
        const char* src = (const char*)(p.rows + (int64_t)g * C::MAXR);
        const uint32_t dst = (uint32_t)(C::ROWS_OFF + (g % 3) * C::ROWS_BUF);
#pragma unroll
        for (int c = 0; c < C::ROWS_CHUNKS; c++) {
            if (c % NW != wave) continue;
            uint32_t off = (uint32_t)(c * 1024 + lane * 16);
            if (off + 16 > (uint32_t)(C::MAXR * 2)) off = 0;      // past the list's slice: fetch a valid address (value unused)
            dma16(src, off, dst + c * 1024);
        }
    };
    // centroid positions of object g -> cpos buffer g & 1, [3][NC] (4-byte DMA: lane = centroid)
    auto dma_cpos = [&](int g) {
        if (wave != (NW > 1 ? 1 : 0)) return;
        const uint32_t dst = (uint32_t)(C::CPOS_OFF + (g & 1) * C::CPOS_BUF);
#pragma unroll
        for (int e = 0; e < 3; e++)
#pragma unroll
            for (int b = 0; b < NC / 64; b++) {
                const float* src = p.out + ((int64_t)g * NC + b * 64 + lane) * (int64_t)p.ldo + N + e;
                __builtin_amdgcn_global_load_lds((gl_void*)src, (lds_void*)(lds + dst + (e * NC + b * 64) * 4), 4, 0, 0);
            }
    };

    for (int ga = g_begin; ga < g_end; ga += kSubR) {
        const int cnt = (g_end - ga) < kSubR ? (g_end - ga) : kSubR;
        wait_all_vm();
        __syncthreads();
        for (int i = tid; i < cnt; i += C::NT) {
            const int g = ga + i;
            nr[i] = p.n_rows[g];
            const int first = p.first[g];
            sbase[i] = first * C::ND + (g - first) * NC;
        }
        // prologue of the sub-range: row lists of its first two objects, positions of the first
        dma_rows(ga);
        if (cnt > 1) dma_rows(ga + 1);
        dma_cpos(ga);
        wait_all_vm();
        __syncthreads();

        // ---- this wave's tile stream ------------------------------------------------------------------------------------
        // (wave-uniform values read from LDS are moved to SGPRs: the tile stream's control flow stays scalar)
        auto rows_of = [&](int gi) { return __builtin_amdgcn_readfirstlane((int)nr[gi]); };
        auto ntile = [&](int gi) { return (rows_of(gi) + 31) >> 5; };
        // first tile of this wave at or after object gi (cnt = none)
        auto first_tile = [&](int gi) -> TileRef {
            while (gi < cnt && ntile(gi) <= wave) gi++;
            return TileRef{gi, wave * 32, gi < cnt ? rows_of(gi) : 0};
        };
        auto next_tile = [&](const TileRef& t) -> TileRef {
            if (t.r0 + NW * 32 < t.n) return TileRef{t.gi, t.r0 + NW * 32, t.n};
            return first_tile(t.gi + 1);
        };
        // row metadata of (tile, lane): list entry min(r0 + rr, n - 1): rows past the end repeat the last row (max is idempotent)
        auto tile_meta = [&](const TileRef& t) -> uint32_t {
            int idx = t.r0 + rr;
            idx = idx < t.n ? idx : t.n - 1;
            const uint16_t* rows_l = (const uint16_t*)(lds + C::ROWS_OFF + ((ga + t.gi) % 3) * C::ROWS_BUF);
            return (uint32_t)rows_l[idx];
        };
        // DMA byte offsets (into p.A) of a tile's rows for the four instructions of a slot
        auto tile_voff = [&](const TileRef& t, uint32_t m, uint32_t (&voff)[4]) {
            const uint32_t src = m & 0xFFu, d = m >> 8;
            const uint32_t g = (uint32_t)(ga + t.gi);
            const uint32_t sb0 = (uint32_t)__builtin_amdgcn_readfirstlane(sbase[t.gi]);
            const uint32_t srow = (d & 0x80u) ? (sb0 + src) : (g * (uint32_t)C::ND + src);
            const uint32_t rowbyte = srow * (uint32_t)(K * 4);
#pragma unroll
            for (int q = 0; q < 4; q++)
                voff[q] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)dma_sel[q], (int)rowbyte) + dma_chunk[q];
        };
        auto issue_slot = [&](const uint32_t (&voff)[4], int u) {
#pragma unroll
            for (int q = 0; q < 4; q++) dma16(p.A, voff[q] + (uint32_t)(u * 128), ringb + (uint32_t)(u * C::SLOT_BYTES + q * 1024));
        };

        // ---- per-object phases --------------------------------------------------------------------------------------------
        auto flush = [&](int g) {   // accumulator -> output rows of object g; leaves the accumulator at +0
            float* o = p.out + (int64_t)g * NC * (int64_t)p.ldo;
            int top = 0;
#pragma unroll
            for (int k = 0; k < NC * N / 4 / C::NT; k++) {
                const int i = tid + k * C::NT;
                const int c = i / (N / 4), c4 = i % (N / 4);
                typedef int i32x4 __attribute__((ext_vector_type(4)));
                i32x4* a = (i32x4*)(acc_lds + c * N + c4 * 4);
                const i32x4 bits = *a;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    top = bits[e] > top ? bits[e] : top;
                    v[e] = __int_as_float(bits[e]) * p.out_scale;
                }
                *(f32x4*)(o + c * (int64_t)p.ldo + c4 * 4) = v;
                *a = i32x4{0, 0, 0, 0};
            }
            guard_track_bits(gbits, top);
        };
        auto build_b = [&](int g) {   // centroid table of object g from its positions (cpos buffer g & 1)
            const float* cp = (const float*)(lds + C::CPOS_OFF + (g & 1) * C::CPOS_BUF);
#pragma unroll
            for (int i = 0; i < CPT; i++) {
                const int c = cg + CGS * i;
                const float px = cp[c], py = cp[NC + c], pz = cp[2 * NC + c];
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; e++) {   // same order as k_sample_group's table: ((x w0) + y w1) + z w2
                    float a = px * wq0[e];
                    a = fmaf(py, wq1[e], a);
                    a = fmaf(pz, wq2[e], a);
                    v[e] = a;
                }
                *(f32x4*)(lds + C::BT_OFF + c * C::BT_STRIDE + cq * 16) = v;
            }
        };

        // first object: its table; the lists / positions the next phases need are put in flight
        build_b(ga);
        if (cnt > 2) dma_rows(ga + 2);
        if (cnt > 1) dma_cpos(ga + 1);
        lds_barrier_r();

        TileRef cur = first_tile(0);
        bool cur_fetched = false;        // the four slots of `cur` are in flight / landed
        uint32_t m_cur = 0;              // metadata of this lane's row of `cur` (valid when cur_fetched)

        for (int gi = 0; gi < cnt; gi++) {
            // ---- tiles of object gi that belong to this wave -------------------------------------------------------------
            bool did_tile = false;
            half8 a_hi, a_lo;
            bool prepped = false;    // a_hi / a_lo hold step 0 of `cur`
            while (cur.gi == gi) {
                did_tile = true;
                if (!cur_fetched) {
                    m_cur = tile_meta(cur);
                    uint32_t v0[4];
                    tile_voff(cur, m_cur, v0);
#pragma unroll
                    for (int u = 0; u < C::SLOTS; u++) issue_slot(v0, u);
                    cur_fetched = true;
                }
                // look ahead: the next tile is prefetched while this one is multiplied, if its row list is in LDS already
                // (same object or the next one); otherwise the same addresses are fetched again to keep the DMA count of the
                // counted waits (the slot is dead by then)
                const TileRef nxt = next_tile(cur);
                const bool nxt_ok = nxt.gi < cnt && nxt.gi <= gi + 1;
                uint32_t m_nxt = m_cur;
                uint32_t vn[4];
                {
                    const TileRef src = nxt_ok ? nxt : cur;
                    m_nxt = tile_meta(src);
                    tile_voff(src, m_nxt, vn);
                }
                // this tile: centroid of the lane's row -> table row, accumulator row
                const uint32_t dl = (m_cur >> 8) & 127u;
                const uint32_t brow = (uint32_t)C::BT_OFF + dl * (uint32_t)C::BT_STRIDE + (uint32_t)(h * 32);
                // (inline asm: hipcc puts s_waitcnt vmcnt(0) in front of an ordinary ds_read it cannot separate from the
                // outstanding LDS-DMA pieces, which would drain the ring once per tile)
                asm volatile("ds_write_b16 %0, %1" ::"v"(dstl_addr + (uint32_t)(rr * 2)), "v"(dl * (uint32_t)(N * 4)) : "memory");

                auto read_step = [&](int s, uint32_t brow_, f32x4 (&x)[2], f32x4 (&b)[2]) {
                    const int u = s >> 1, par = s & 1;
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        x[j] = *(const f32x4*)(lds + rd[par][j] + u * C::SLOT_BYTES);
                        b[j] = *(const f32x4*)(lds + brow_ + s * 64 + j * 16);
                    }
                };
                auto prep = [&](const f32x4 (&x)[2], const f32x4 (&b)[2], half8& oh, half8& ol) {
                    fp16x2 hh[4], ll[4];
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        const f32x4 t = x[j] - b[j];
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; e++) v[e] = fmaxf(t[e], 0.f);
                        hh[2 * j] = __builtin_amdgcn_cvt_pkrtz(v[0], v[1]);
                        hh[2 * j + 1] = __builtin_amdgcn_cvt_pkrtz(v[2], v[3]);
                        ll[2 * j] = __builtin_amdgcn_cvt_pkrtz(sub_half_r<0>(v[0], hh[2 * j]), sub_half_r<1>(v[1], hh[2 * j]));
                        ll[2 * j + 1] = __builtin_amdgcn_cvt_pkrtz(sub_half_r<0>(v[2], hh[2 * j + 1]), sub_half_r<1>(v[3], hh[2 * j + 1]));
                    }
                    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                    u32x4 ph, pl;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        ph[e] = __builtin_bit_cast(uint32_t, hh[e]);
                        pl[e] = __builtin_bit_cast(uint32_t, ll[e]);
                    }
                    oh = __builtin_bit_cast(half8, ph);
                    ol = __builtin_bit_cast(half8, pl);
                };

                if (!prepped) {   // step 0 of this tile (first tile of an object, or a tile that was not prefetched)
                    wait_ring();
                    f32x4 x[2], b[2];
                    read_step(0, brow, x, b);
                    prep(x, b, a_hi, a_lo);
                }
                f32x16 acc[C::NTW];
#pragma unroll
                for (int nt = 0; nt < C::NTW; nt++)
#pragma unroll
                    for (int e = 0; e < 16; e++) acc[nt][e] = biasv[nt];

                // the next tile's first step can be prepared inside this one only when it reads the same centroid table
                const bool chain = nxt_ok && nxt.gi == gi;
                const uint32_t dl_n = (m_nxt >> 8) & 127u;
                const uint32_t brow_n = (uint32_t)C::BT_OFF + dl_n * (uint32_t)C::BT_STRIDE + (uint32_t)(h * 32);
                half8 n_hi = a_hi, n_lo = a_lo;
#pragma unroll
                for (int s = 0; s < C::S16; s++) {
                    f32x4 x[2], b[2];
                    bool have_next = true;
                    if (s + 1 < C::S16) {
                        if (((s + 1) & 1) == 0) wait_ring();      // first step of the next slot
                        read_step(s + 1, brow, x, b);
                    } else if (chain) {
                        wait_ring();                             // slot 0 of the next tile
                        read_step(0, brow_n, x, b);
                    } else {
                        have_next = false;
                    }
#pragma unroll
                    for (int nt = 0; nt < C::NTW; nt++) acc[nt] = MFMA16(a_hi, w_hi[nt][s], acc[nt]);
#pragma unroll
                    for (int nt = 0; nt < C::NTW; nt++) acc[nt] = MFMA16(a_hi, w_lo[nt][s], acc[nt]);
                    if (have_next) prep(x, b, n_hi, n_lo);
                    // slot u is consumed once the second of its two steps has been read and converted: refill it
                    // (WAR: the reads of the slot must have RETURNED before its refill is issued, not merely been issued: the
                    // explicit wait also pins the order - hipcc otherwise puts the DMA right behind the ds_reads)
                    if ((s & 1) == 0) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        issue_slot(vn, s >> 1);
                    }
#pragma unroll
                    for (int nt = 0; nt < C::NTW; nt++) acc[nt] = MFMA16(a_lo, w_hi[nt][s], acc[nt]);
                    a_hi = n_hi;
                    a_lo = n_lo;
                }
                prepped = chain;
                // max-aggregation: integer atomic max into the object's LDS accumulator (the max against +0 is the ReLU)
                {
                    uint2 four[4];   // accumulator-row byte offsets of this lane's 16 result rows 8 q + 4 h + {0..3}
                    {
                        const uint32_t a4 = dstl_addr + (uint32_t)(h * 8);
                        asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:16\n\tds_read_b64 %2, %4 offset:32\n\t"
                                     "ds_read_b64 %3, %4 offset:48\n\ts_waitcnt lgkmcnt(0)"
                                     : "=&v"(four[0]), "=&v"(four[1]), "=&v"(four[2]), "=&v"(four[3]) : "v"(a4) : "memory");
                    }
                    // (inline asm for the same reason as above; the results sit in AGPRs, which a DS instruction reads directly.
                    // The s_nop covers the MFMA -> LDS-data hazard the compiler no longer sees.)
                    const uint32_t col = (uint32_t)(C::ACC_OFF + rr * 4);
                    uint32_t ad[16];
#pragma unroll
                    for (int e = 0; e < 16; e++) {
                        const uint32_t pair = (e & 2) ? four[e >> 2].y : four[e >> 2].x;
                        ad[e] = col + ((e & 1) ? (pair >> 16) : (pair & 0xFFFFu));
                    }
                    asm volatile("s_nop 15" ::: "memory");
#pragma unroll
                    for (int nt = 0; nt < C::NTW; nt++)
#pragma unroll
                        for (int e = 0; e < 16; e++)
                            asm volatile("ds_max_i32 %0, %1 offset:%2" ::"v"(ad[e]), "a"(acc[nt][e]), "n"(nt * 128) : "memory");
                }
                cur = nxt;
                cur_fetched = nxt_ok;
                m_cur = m_nxt;
            }
            // ---- object gi is complete for this wave ------------------------------------------------------------------------
            // the DMA pieces this wave issued for later objects (row lists, positions) are older than any ring piece a
            // counted wait has since retired - unless the wave had no tile here
            if (!did_tile) wait_all_vm();
            lds_barrier_r();                                    // A: all atomics of object gi are in the accumulator
            flush(ga + gi);
            if (gi + 1 < cnt) {
                build_b(ga + gi + 1);
                if (gi + 3 < cnt) dma_rows(ga + gi + 3);
                if (gi + 2 < cnt) dma_cpos(ga + gi + 2);
            }
            lds_barrier_r();                                    // B: accumulator cleared, next table in place
        }
    }
    wait_all_vm();
    if (p.amax_out != nullptr && lane == 0 && gbits != 0u)
        atomicMax(p.amax_out, __float_as_uint(__uint_as_float(gbits) * p.out_scale));
}

}  // namespace

bool sa_rows_selected(int H, int Cout, const SaParams& p) {
    return H == 128 && Cout == 128 && p.W_x3 != nullptr && p.wp != nullptr && !(p.plan & 1);
}

// (tile rows, workgroups) for the range balancing: four waves share an object, a round of the workgroup covers 128 rows
int sa_rows_launch_shape(int64_t n_obj, int* tile_rows, int* n_wg) {
    int n = num_cus();
    if (n > 1024) n = 1024;
    if (n > n_obj) n = (int)n_obj;
    *tile_rows = 4 * 32;
    *n_wg = n;
    return 0;
}

int launch_sa_rows(int H, int Cout, const SaParams& p, hipStream_t st) {
    if (!(H == 128 && Cout == 128 && p.n_cent == 64 && p.n_dense == 128 && p.wp && p.W_x3)) {
        set_error("sa_rows: built for SA level 2 (H = C = 128, 64 centroids of 128 points, f16x3, LDS centroid table)");
        return T2P_E_UNSUPPORTED;
    }
    using C = RowsCfg<128, 128, 64, 4>;
    auto kern = k_sa_rows<128, 128, 64, 4>;
    T2P_TRY(reserve_lds((const void*)kern, C::lds_bytes(), "sa_rows"));
    if (p.n_obj <= 0) return 0;
    T2P_CHECK_ARG(p.n_obj < (1 << 30) && p.n_obj * p.n_dense * (int64_t)H * 4 < 0xffffffffLL,
                  "sa_rows: chunk too large for 32-bit table offsets");
    T2P_CHECK_ARG((((uintptr_t)p.A | (uintptr_t)p.rows | (uintptr_t)p.out | (uintptr_t)p.W_x3) & 15) == 0 && p.ldo % 4 == 0,
                  "sa_rows: tables must be 16-byte aligned");
    int tr, n_wg;
    sa_rows_launch_shape(p.n_obj, &tr, &n_wg);
    if (!p.balanced) T2P_TRY(launch_sa_balance(p, tr, n_wg, st));
    ProfScope ps_("ws_edge_sa_k128_n128", st);
    hipLaunchKernelGGL(kern, dim3(n_wg), dim3(C::NT), C::lds_bytes(), st, p);
    T2P_CHECK_LAUNCH("sa_rows");
    return 0;
}

}  // namespace t2p

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
//   * (rounds 3-5; still built with -DT2P_ROWS_DIRECT=0) ring of 4 slots x 4 KB per wave = one whole tile of prefetch: slot u holds
//     the 128-byte pieces (k = 32 u .. 32 u + 31) of the tile's 32 rows, 8 rows x 128 B per DMA instruction, XOR-swizzled on the SOURCE
//     address so that the lane-linear LDS image is read conflict-free with ds_read_b128; counted s_waitcnt vmcnt(8), never vmcnt(0)
//     inside the stream.  Round 6 (default): every lane loads its own operand bytes straight into registers - T2P_ROWS_DIRECT below;
//   * the row lists and centroid positions of the objects ahead travel by LDS-DMA;
//   * natural k order (lane half h owns k = 16 s + 8 h .. + 7 of MFMA step s): the host's register-order weight image is
//     re-indexed at load time; every fp32 accumulation runs hi.hi, hi.lo, lo.hi per step like ws_sa2.hip, with another
//     grouping of the k's (results agree to fp32 rounding, not bit for bit).
// T2P_RABL (development only, results are wrong): 1 = no ring DMA, 2 = no atomics, 4 = no drain stores / table build,
// 8 = no MFMAs (first and last of a step kept), 16 = no object barriers.  An ablation that leaves ZEROS where data was also lowers the
// matrix pipe's power and with it the time of this kernel and of those behind it: read the table with that in mind.
#ifndef T2P_RABL
#define T2P_RABL 0
#endif
#include "t2p_common.h"

namespace t2p {
int launch_sa_balance(const SaParams& p, int tile_rows, int n_wg, hipStream_t st);  // ws_sa.hip

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gl_void;

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
// T2P_ROWS_AGPR (round 6): the weight matrix does not fit the 256 architectural VGPRs next to the loop's working set, and hipcc parks
// what does not fit (39 of the 64 half8 operands) in AGPRs as SPILL space: 96 v_accvgpr_read_b32 per tile copy them back, operand by
// operand, in front of their MFMAs (ISA census: 96 of the loop's 859 non-MFMA instructions - with one wave per SIMD every one of
// them comes out of the MFMA stream).  gfx950's MFMA reads its B operand from an AGPR just as well: here the MFMAs are inline asm whose
// weight operand is constrained to the register file the operand LIVES in - the whole lo plane and the first WA_HI_STEPS steps of
// the hi plane in AGPRs (moved there once, at kernel start), the rest in VGPRs - and the accumulators stay in AGPRs, where
// ds_max_f32 reads them.  Same instructions, same order of accumulation: same bits.
#ifndef T2P_ROWS_AGPR
#define T2P_ROWS_AGPR 1
#endif
// T2P_ROWS_DIRECT (round 6): the tile's rows no longer travel through an LDS ring.  The ring (global_load_lds pieces of 8 rows x
// 128 B, read back with ds_read_b128) cost ~60 cycles of issue per piece among the MFMAs of a one-wave SIMD - 16 pieces per tile, 2.1 ms
// of SA2's 19.2 ms per step by ablation (profiles/r06_d_sa2_ablation.txt), 4.5 ms of the step through the power cap.  The MFMA A operand
// of lane (h, rr) is 32 contiguous bytes of ITS OWN row rr per step (k = 16 s + 8 h .. + 7): each lane now fetches exactly those with two
// global_load_dwordx4 per step from `row base + 64 s + 32 h` into a window of registers (the registers the AGPR-resident weights
// freed).  No ring slots, no ds_bpermute address shuffle, no M0 writes, and the 64 KB of ring leave LDS.
// The loads are ordinary loads, so hipcc places the waits - and across the tile loop's back edge it only ever waits for ALL outstanding
// loads (vmcnt(0)) at the first use of a batch.  The schedule is built around that: two bursts of eight loads per tile, each issued
// right BEHIND a wait and used three to four steps (>= 2,000 cycles) later, so that a wait never finds a young load outstanding:
//   burst Y(t)   = steps 5, 6, 7 of tile t and step 0 of tile t + 1, issued in steps 0 - 1 of tile t, first used in step 4
//   burst X(t+1) = steps 1 - 4 of tile t + 1,                        issued in steps 4 - 5 of tile t, first used in step 0 of tile t + 1
// (the conversion of step s + 1 runs inside step s; step 0 of a tile inside its predecessor's last step).
// Measured (three interleaved A/B pairs, profiles/r06_f_ab_direct.txt): 19.2-19.3 ms per step against 19.6-19.8 with the ring (-2 %) for
// 630 instead of 818 instructions per tile - and exactly as much with the same loads as inline asm under the ring's counted waits
// (vmcnt(8), 6-7 steps of prefetch): neither instruction issue nor prefetch depth is what the rows cost, the bytes are.  (The
// "no ring DMA" ablation's 2.1 ms was mostly its all-zero operands: zeros cost the matrix pipe less power.)
#ifndef T2P_ROWS_DIRECT
#define T2P_ROWS_DIRECT 1
#endif
// load issued behind conversion chunk c of step s (S16 = 8 steps): 2 L + j = piece j of logical step L (L >= 8: next tile), or -1
__device__ __forceinline__ constexpr int rows_load_slot(int s, int c) {
    if (s == 0) return c == 0 || c == 4 ? -1 : 10 + (c < 4 ? c - 1 : c - 2);      // c = 1,2,3,5,6,7 -> steps 5, 6, 7
    if (s == 1) return c == 1 ? 16 : (c == 3 ? 17 : -1);                          // next tile's step 0 (its row offset exists from step 0 on)
    if (s == 4) return (c & 1) ? 18 + (c >> 1) : -1;                              // c = 1,3,5,7 -> next tile's steps 1, 2
    if (s == 5) return (c & 1) ? 22 + (c >> 1) : -1;                              //             -> next tile's steps 3, 4
    return -1;
}
#ifndef T2P_WA_HI_STEPS
#define T2P_WA_HI_STEPS 3
#endif
constexpr int WA_HI_STEPS = T2P_WA_HI_STEPS;     // AGPR budget: 64 accumulator registers + 128 (lo plane) + 16 * WA_HI_STEPS (hi plane) <= 256
__device__ __forceinline__ constexpr bool w_in_agpr(bool lo_plane, int s) { return lo_plane || s < WA_HI_STEPS; }
// c (+)= a x w on v_mfma_f32_32x32x16_f16; `agpr`: where w lives; `zero`: start from 0 (inline constant) instead of c.
// HAZARD the compiler cannot see here: an MFMA must not read, as its A / B operand, a VGPR that a VALU instruction wrote less than two
// issue slots earlier (t2p_common.h, split_lo_pk).  The A operands of step s + 1 are assembled at the END of step s and first multiplied
// behind the centroid-table reads that open step s + 1 (>= 2 instructions, pinned by the sched_barriers); the last step has no such reads
// when the tile does not chain, so it opens with an explicit s_nop 1.  tests/test_host.py compiles this file and checks the distance
// in the generated code.
__device__ __forceinline__ void mfma_w(f32x16& c, const half8& a, const half8& w, bool agpr, bool zero) {
#if T2P_ROWS_AGPR
    if (zero) {
        if (agpr) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "a"(w));
        else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(w));
    } else {
        if (agpr) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "a"(w));
        else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(w));
    }
#else
    constexpr f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    c = MFMA16(a, w, zero ? z : c);
#endif
}
#define SB() __builtin_amdgcn_sched_barrier(0)
#if T2P_RABL & 2
#define DS_MAX_STR "; no atomic %0 %1 %2"
#else
#define DS_MAX_STR "ds_max_f32 %0, %1 offset:%2"
#endif

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
    static constexpr size_t lds_bytes() { return (size_t)RING_OFF + (T2P_ROWS_DIRECT ? (size_t)0 : (size_t)NW * RING_BYTES); }
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
// counted wait for the ring: the 8 youngest DMA instructions (two slots) may stay in flight; the slot the wait retires is
// read right away, the slot emptied one step earlier is refilled BEHIND the wait in the same step
__device__ __forceinline__ void wait_ring() { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
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
#if T2P_ROWS_AGPR
                // into the AGPR file, once (an empty asm whose AGPR output is tied to the loaded value)
                if (w_in_agpr(false, s)) asm volatile("" : "=a"(w_hi[nt][s]) : "0"(w_hi[nt][s]));
                asm volatile("" : "=a"(w_lo[nt][s]) : "0"(w_lo[nt][s]));
#endif
            }
    }
    // The bias is NOT part of the accumulation: a tile's first MFMAs start from 0, the LDS accumulator takes a FLOAT max of the
    // raw products (ds_max_f32 orders negative values correctly) from a -inf start, and the drain forms relu(max + bias).
    // (A bias block as the first MFMA's C operand costs 64 registers of a budget that is full.)
    f32x4 bias4;         // this thread's four output columns in the drain (column quad = tid % (N / 4))
    {
        const int c4 = tid % (N / 4);
        bias4 = *(const f32x4*)(p.bias + c4 * 4);
    }
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

    for (int i = tid; i < NC * N; i += C::NT) acc_lds[i] = (int)0xFF800000;   // -inf
    int gtop = 0;         // fp16-range guard: this lane's maximum (bit pattern, before out_scale) of the drained outputs; reduced over
                          // the wave once, at the end (six ds_bpermute round trips per drain otherwise)

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
        // Tiles of an object go round-robin to the waves: wave w owns rows [32 (w + NW i), +32).  All stream state is wave-uniform
        // and lives in SGPRs (row counts / self-loop bases are fetched from LDS once per object).
        auto rows_of = [&](int gi) { return __builtin_amdgcn_readfirstlane((int)nr[gi]); };
        // row metadata of (object gi, first row r0, lane): list entry min(r0 + rr, n - 1) - rows past the end repeat the last row
        // (max is idempotent)
        auto tile_meta = [&](int gi, int r0, int n) -> uint32_t {
            int idx = r0 + rr;
            idx = idx < n ? idx : n - 1;
            const uint16_t* rows_l = (const uint16_t*)(lds + C::ROWS_OFF + ((ga + gi) % 3) * C::ROWS_BUF);
            return (uint32_t)rows_l[idx];
        };
        // byte offset (into p.A) of the lane's row
        auto row_byte = [&](int gi, uint32_t sb0, uint32_t m) -> uint32_t {
            const uint32_t src = m & 0xFFu, d = m >> 8;
            const uint32_t g = (uint32_t)(ga + gi);
            const uint32_t srow = (d & 0x80u) ? (sb0 + src) : (g * (uint32_t)C::ND + src);
            return srow * (uint32_t)(K * 4);
        };
        // ... and of the rows each DMA instruction of a slot fetches
        [[maybe_unused]] auto tile_voff = [&](uint32_t rowbyte, uint32_t (&voff)[4]) {   // (ring build only: -DT2P_ROWS_DIRECT=0)
#pragma unroll
            for (int q = 0; q < 4; q++)
                voff[q] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)dma_sel[q], (int)rowbyte) + dma_chunk[q];
        };
        auto dma_piece = [&](const uint32_t (&voff)[4], int u, int q) {
            if constexpr (!(T2P_RABL & 1))
                dma16(p.A, voff[q] + (uint32_t)(u * 128), ringb + (uint32_t)(u * C::SLOT_BYTES + q * 1024));
        };
        [[maybe_unused]] auto issue_slot = [&](const uint32_t (&voff)[4], int u) {
#pragma unroll
            for (int q = 0; q < 4; q++) dma_piece(voff, u, q);
        };
#if T2P_ROWS_DIRECT
        // direct operand loads: lane (h, rr) reads bytes [64 s + 32 h, + 32) of its own row rr
        constexpr int AHEAD = 5;                       // steps between a load and its use (the window holds AHEAD x 8 registers)
        const uint32_t hoff = (uint32_t)(h * 32);
        auto load_half = [&](uint32_t roff, int s_, int j, f32x4 (&x)[2]) {
            if constexpr (!(T2P_RABL & 1)) x[j] = *(const f32x4*)((const char*)p.A + (size_t)roff + (size_t)(s_ * 64 + j * 16));
        };
        auto load_step = [&](uint32_t roff, int s_, f32x4 (&x)[2]) {
            load_half(roff, s_, 0, x);
            load_half(roff, s_, 1, x);
        };
#endif

        // ---- per-object phases --------------------------------------------------------------------------------------------
        auto flush = [&](int g) {   // accumulator -> output rows of object g: relu(max + bias); leaves the accumulator at -inf
            float* o = p.out + (int64_t)g * NC * (int64_t)p.ldo;
            int top = gtop;
            static_assert(C::NT % (N / 4) == 0, "a thread keeps its column quad over the drain");
#pragma unroll
            for (int k = 0; k < NC * N / 4 / C::NT; k++) {
                const int i = tid + k * C::NT;
                const int c = i / (N / 4), c4 = i % (N / 4);
                typedef int i32x4 __attribute__((ext_vector_type(4)));
                f32x4* a = (f32x4*)(acc_lds + c * N + c4 * 4);
                const f32x4 raw = *a;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float r = fmaxf(raw[e] + bias4[e], 0.f);      // (a centroid without rows stays at -inf: 0, as before)
                    const int bits = __float_as_int(r);
                    top = bits > top ? bits : top;
                    v[e] = r * p.out_scale;
                }
                *(f32x4*)(o + c * (int64_t)p.ldo + c4 * 4) = v;
                *(i32x4*)a = i32x4{(int)0xFF800000, (int)0xFF800000, (int)0xFF800000, (int)0xFF800000};
            }
            gtop = top;
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

        bool cur_fetched = false;        // the four slots of this wave's next tile (first of object gi) are in flight / landed
        uint32_t m_cur = 0;              // ... and this is the metadata of the lane's row of it
#if T2P_ROWS_DIRECT
        uint32_t roff_cur = 0;           // ... its byte offset in p.A (+ the lane half's 32 bytes)
        f32x4 xw[C::S16][2];             // rolling window of operand pieces: xw[s] = step s of the tile that needs it next
#pragma unroll
        for (int s0 = 0; s0 < C::S16; s0++) xw[s0][0] = xw[s0][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#endif

        for (int gi = 0; gi < cnt; gi++) {
            // ---- tiles of object gi that belong to this wave -------------------------------------------------------------
            const int n_g = rows_of(gi), n_g1 = gi + 1 < cnt ? rows_of(gi + 1) : 0;
            const uint32_t sb_g = (uint32_t)__builtin_amdgcn_readfirstlane(sbase[gi]);
            const uint32_t sb_g1 = gi + 1 < cnt ? (uint32_t)__builtin_amdgcn_readfirstlane(sbase[gi + 1]) : 0u;
            int r0 = wave * 32;
            bool have = r0 < n_g;
            const bool did_tile = have;
            half8 a_hi, a_lo;
            f32x16 acc[C::NTW];
            bool pend = false;       // the last result block of the previous tile still waits for its atomics (rows: fourp)
            uint2 fourp[4] = {};
            // ---- the wave's first tile of this object: fetched here unless the previous object's last tile already prefetched it; its step 0
            // is converted here.  (Both used to sit inside the tile loop behind `!cur_fetched` / `!prepped`; they are first-iteration-only,
            // and with ordinary loads in flight hipcc's waitcnt pass merged their register state into every iteration.)
            auto read_step = [&](int s, uint32_t brow_, f32x4 (&x)[2], f32x4 (&b)[2]) {
#if T2P_ROWS_DIRECT
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    x[j] = xw[s][j];                      // fetched AHEAD steps ago (a register rename: the loops are unrolled)
                    b[j] = *(const f32x4*)(lds + brow_ + s * 64 + j * 16);
                }
#else
                const int u = s >> 1, par = s & 1;
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    x[j] = *(const f32x4*)(lds + rd[par][j] + u * C::SLOT_BYTES);
                    b[j] = *(const f32x4*)(lds + brow_ + s * 64 + j * 16);
                }
#endif
            };
            // conversion of one step's 8 values in 8 half-chunks of 4 VALU operations (pair pr = values 2 pr, 2 pr + 1):
            // first half v = relu(x - b), second half hi = fp16(v) to nearest, lo = fp16(v - hi)
            auto prep_a = [&](int pr, const f32x4 (&x)[2], const f32x4 (&b)[2], float (&v)[2]) {
                const int j = pr >> 1, e0 = (pr & 1) * 2;
                v[0] = fmaxf(x[j][e0] - b[j][e0], 0.f);
                v[1] = fmaxf(x[j][e0 + 1] - b[j][e0 + 1], 0.f);
            };
            auto prep_b = [&](const float (&v)[2], uint32_t& wh, uint32_t& wl) {
                const fp16x2 hh = cvt_pk_f16(v[0], v[1]);
                wh = __builtin_bit_cast(uint32_t, hh);
#if T2P_ROWS_AGPR
                wl = split_lo_pk(hh, v[0], v[1]);      // (one asm statement: t2p_common.h)
#else
                const fp16x2 ll = cvt_pk_f16(sub_half_r<0>(v[0], hh), sub_half_r<1>(v[1], hh));
                wl = __builtin_bit_cast(uint32_t, ll);
#endif
            };
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            auto row_addr = [&](const uint2 (&f4)[4], int e) -> uint32_t {
                const uint32_t pair = (e & 2) ? f4[e >> 2].y : f4[e >> 2].x;
                return (uint32_t)(C::ACC_OFF + rr * 4) + ((e & 1) ? (pair >> 16) : (pair & 0xFFFFu));
            };

            if (have) {
                if (!cur_fetched) {
                    m_cur = tile_meta(gi, r0, n_g);
#if T2P_ROWS_DIRECT
                    roff_cur = row_byte(gi, sb_g, m_cur) + hoff;
#pragma unroll
                    for (int s0 = 0; s0 < AHEAD; s0++) load_step(roff_cur, s0, xw[s0]);
#else
                    uint32_t v0[4];
                    tile_voff(row_byte(gi, sb_g, m_cur), v0);
#pragma unroll
                    for (int u = 0; u < C::SLOTS; u++) issue_slot(v0, u);
#endif
                }
                {   // step 0 of the wave's first tile in this object: converted here, every later tile's step 0 inside its predecessor's last step
                    if constexpr (!T2P_ROWS_DIRECT) wait_ring();
                    f32x4 x[2], b[2];
                    read_step(0, (uint32_t)C::BT_OFF + ((m_cur >> 8) & 127u) * (uint32_t)C::BT_STRIDE + (uint32_t)(h * 32), x, b);
                    uint32_t nh[4], nl[4];
#pragma unroll
                    for (int pr = 0; pr < 4; pr++) {
                        float v[2];
                        prep_a(pr, x, b, v);
                        prep_b(v, nh[pr], nl[pr]);
                    }
                    a_hi = __builtin_bit_cast(half8, u32x4{nh[0], nh[1], nh[2], nh[3]});
                    a_lo = __builtin_bit_cast(half8, u32x4{nl[0], nl[1], nl[2], nl[3]});
                }
            }
            while (have) {
                // look ahead: the next tile is prefetched while this one is multiplied, if its row list is in LDS already
                // (same object or the next one); otherwise the same addresses are fetched again to keep the DMA count of the
                // counted waits (the slot is dead by then)
                const bool chain = r0 + NW * 32 < n_g;                              // next tile in the same object
                const bool nxt_ok = chain || (gi + 1 < cnt && wave * 32 < n_g1);
                const int gi_n = chain ? gi : (nxt_ok ? gi + 1 : gi);
                const int r0_n = chain ? r0 + NW * 32 : (nxt_ok ? wave * 32 : r0);
                const int n_n = (chain || !nxt_ok) ? n_g : n_g1;
                const uint32_t sb_n = (chain || !nxt_ok) ? sb_g : sb_g1;
                // its metadata: the LDS read is issued here, decoded inside step 0
                uint32_t m_nxt = tile_meta(gi_n, r0_n, n_n);
                uint32_t vn[4];
#if T2P_ROWS_DIRECT
                uint32_t roff_n = roff_cur;
                (void)vn;
#endif
                // this tile: centroid of the lane's row -> table row, accumulator row
                const uint32_t dl = (m_cur >> 8) & 127u;
                const uint32_t brow = (uint32_t)C::BT_OFF + dl * (uint32_t)C::BT_STRIDE + (uint32_t)(h * 32);
                // (inline asm: hipcc puts s_waitcnt vmcnt(0) in front of an ordinary LDS access it cannot separate from the
                // outstanding LDS-DMA pieces, which would drain the ring once per tile)
                asm volatile("ds_write_b16 %0, %1" ::"v"(dstl_addr + (uint32_t)(rr * 2)), "v"(dl * (uint32_t)(N * 4)) : "memory");
                uint2 four[4];   // accumulator-row byte offsets of this lane's 16 result rows 8 q + 4 h + {0..3} (fetched in step 6)

                uint32_t brow_n = 0;

                // ---- steps 0 .. S16-2: [reads of step s+1] [4 MFMAs hi.hi] [8 x (half-chunk of the conversion, 1 MFMA)] --------
                // One wave per SIMD: the instruction order IS the schedule, so it is pinned with sched_barrier fences: the LDS
                // round trip of the reads hides behind the first four MFMAs, every later MFMA carries ~4 VALU operations.  Step 0
                // also decodes the next tile (row offsets -> DMA addresses), odd steps refill the slot the previous step emptied.
#pragma unroll
                for (int s = 0; s < C::S16 - 1; s++) {
                    f32x4 x[2], b[2];
                    if (!T2P_ROWS_DIRECT && ((s + 1) & 1) == 0) wait_ring();      // first step of the next slot
                    read_step(s + 1, brow, x, b);
                    if (s == C::S16 - 2) {
                        const uint32_t a4 = dstl_addr + (uint32_t)(h * 8);
                        asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:16\n\tds_read_b64 %2, %4 offset:32\n\t"
                                     "ds_read_b64 %3, %4 offset:48"
                                     : "=&v"(four[0]), "=&v"(four[1]), "=&v"(four[2]), "=&v"(four[3]) : "v"(a4) : "memory");
                    }
                    SB();
                    if (s == 0) {
#pragma unroll
                        for (int nt = 0; nt < C::NTW - 1; nt++) mfma_w(acc[nt], a_hi, w_hi[nt][0], w_in_agpr(false, 0), true);
                        if (pend) {   // the previous tile's last result block: its MFMAs are long done, its registers are free below
#pragma unroll
                            for (int e = 0; e < 16; e++)
                                asm volatile(DS_MAX_STR ::"v"(row_addr(fourp, e)), "a"(acc[C::NTW - 1][e]), "n"((C::NTW - 1) * 128) : "memory");
                        }
                        mfma_w(acc[C::NTW - 1], a_hi, w_hi[C::NTW - 1][0], w_in_agpr(false, 0), true);
                    } else {
#pragma unroll
                        for (int nt = 0; nt < C::NTW; nt++) mfma_w(acc[nt], a_hi, w_hi[nt][s], w_in_agpr(false, s), false);
                    }
                    SB();
                    uint32_t nh[4], nl[4];
                    float v[2];
                    uint32_t rowbyte_n = 0;
#pragma unroll
                    for (int c = 0; c < 8; c++) {
                        if ((c & 1) == 0) prep_a(c >> 1, x, b, v);
                        else prep_b(v, nh[c >> 1], nl[c >> 1]);
                        if (s == 0) {          // next tile: row offset of the lane's row, the rows of every DMA lane, table row
                            if (c == 2) rowbyte_n = row_byte(gi_n, sb_n, m_nxt);
#if T2P_ROWS_DIRECT
                            if (c == 4) roff_n = rowbyte_n + hoff;
#else
                            if (c == 4) tile_voff(rowbyte_n, vn);
#endif
                            if (c == 6) brow_n = (uint32_t)C::BT_OFF + ((m_nxt >> 8) & 127u) * (uint32_t)C::BT_STRIDE + (uint32_t)(h * 32);
                        }
#if T2P_ROWS_DIRECT
                        {   // this chunk's load, if the schedule has one (rows_load_slot): at most one per MFMA gap
                            const int ld = rows_load_slot(s, c);
                            if (ld >= 0) load_half(ld < 2 * C::S16 ? roff_cur : roff_n, (ld >> 1) % C::S16, ld & 1, xw[(ld >> 1) % C::S16]);
                        }
#else
                        if ((s & 1) && (c & 1)) dma_piece(vn, s >> 1, c >> 1);   // refill of slot (s - 1) / 2, emptied in step s - 1
#endif
                        if (c < 4) mfma_w(acc[c], a_hi, w_lo[c][s], w_in_agpr(true, s), false);
                        else mfma_w(acc[c - 4], a_lo, w_hi[c - 4][s], w_in_agpr(false, s), false);
                        SB();
                    }
                    a_hi = __builtin_bit_cast(half8, u32x4{nh[0], nh[1], nh[2], nh[3]});
                    a_lo = __builtin_bit_cast(half8, u32x4{nl[0], nl[1], nl[2], nl[3]});
                }
                // ---- last step, column block by column block: a finished block's atomics ride under the next block's MFMAs --------
                {
                    constexpr int s = C::S16 - 1;
                    f32x4 x[2], b[2];
                    if (chain) {
                        if constexpr (!T2P_ROWS_DIRECT) wait_ring();     // slot 0 of the next tile
                        read_step(0, brow_n, x, b);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 1" : "+v"(four[0]), "+v"(four[1]), "+v"(four[2]), "+v"(four[3])::"memory");
                    SB();
                    uint32_t ad[16];
                    uint32_t nh[4], nl[4];
                    float v[2];
#pragma unroll
                    for (int i = 0; i < 3 * C::NTW; i++) {
                        const int nt = i / 3, k = i % 3;
                        if (k == 0) mfma_w(acc[nt], a_hi, w_hi[nt][s], w_in_agpr(false, s), false);
                        else if (k == 1) mfma_w(acc[nt], a_hi, w_lo[nt][s], w_in_agpr(true, s), false);
                        else mfma_w(acc[nt], a_lo, w_hi[nt][s], w_in_agpr(false, s), false);
                        if (i < 4) {
#pragma unroll
                            for (int e = 4 * i; e < 4 * i + 4; e++) ad[e] = row_addr(four, e);
                        }
                        if (chain && i < 8) {
                            if ((i & 1) == 0) prep_a(i >> 1, x, b, v);
                            else prep_b(v, nh[i >> 1], nl[i >> 1]);
                        }
#if !T2P_ROWS_DIRECT
                        if (i < 8 && (i & 1)) dma_piece(vn, s >> 1, i >> 1);      // refill of the last slot
#endif
                        // block nb = (i - 4) / 3 is complete two MFMAs before chunk i = 3 nb + 4: 8 atomics here, 8 in the next chunk
                        if (i >= 4 && ((i - 4) % 3) < 2 && (i - 4) / 3 < C::NTW - 1) {
                            const int nb = (i - 4) / 3, e0 = ((i - 4) % 3) * 8;
#pragma unroll
                            for (int e = e0; e < e0 + 8; e++)
                                asm volatile(DS_MAX_STR ::"v"(ad[e]), "a"(acc[nb][e]), "n"(nb * 128) : "memory");
                        }
                        SB();
                    }
                    if (chain) {
                        a_hi = __builtin_bit_cast(half8, u32x4{nh[0], nh[1], nh[2], nh[3]});
                        a_lo = __builtin_bit_cast(half8, u32x4{nl[0], nl[1], nl[2], nl[3]});
#pragma unroll
                        for (int q = 0; q < 4; q++) fourp[q] = four[q];
                        pend = true;
                    } else {      // last tile of this wave in the object: the last block goes out now
                        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
                        for (int e = 0; e < 16; e++)
                            asm volatile(DS_MAX_STR ::"v"(ad[e]), "a"(acc[C::NTW - 1][e]), "n"((C::NTW - 1) * 128) : "memory");
                        pend = false;
                    }
                }
                have = chain;
                r0 = r0_n;
                cur_fetched = nxt_ok;
                m_cur = m_nxt;
#if T2P_ROWS_DIRECT
                roff_cur = roff_n;
#endif
            }
            if (!did_tile) cur_fetched = false;
            // ---- object gi is complete for this wave ------------------------------------------------------------------------
            // the DMA pieces this wave issued for later objects (row lists, positions) are older than any ring piece a
            // counted wait has since retired - unless the wave had no tile here
            if (!did_tile) wait_all_vm();
            if constexpr (!(T2P_RABL & 16)) lds_barrier_r();    // A: all atomics of object gi are in the accumulator
            if constexpr (!(T2P_RABL & 4)) flush(ga + gi);
            if (gi + 1 < cnt) {
                if constexpr (!(T2P_RABL & 4)) build_b(ga + gi + 1);
                if (gi + 3 < cnt) dma_rows(ga + gi + 3);
                if (gi + 2 < cnt) dma_cpos(ga + gi + 2);
            }
            if constexpr (!(T2P_RABL & 16)) lds_barrier_r();    // B: accumulator cleared, next table in place
        }
    }
    wait_all_vm();
    uint32_t gbits = 0;
    guard_track_bits(gbits, gtop);
    if (p.amax_out != nullptr && lane == 0 && gbits != 0u)
        atomicMax(p.amax_out, __float_as_uint(__uint_as_float(gbits) * p.out_scale));
}

}  // namespace

bool sa_rows_selected(int H, int Cout, const SaParams& p) {
    return H == 128 && Cout == 128 && p.W_x3 != nullptr && p.wp != nullptr;
}

// (tile rows, workgroups) for the range balancing: four waves share an object, a round of the workgroup covers 128 rows
int sa_rows_launch_shape(int64_t n_obj, int* tile_rows, int* n_wg) {
    int n = matrix_wgs();
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
    T2P_REPEAT(ps_) hipLaunchKernelGGL(kern, dim3(n_wg), dim3(C::NT), C::lds_bytes(), st, p);
    T2P_CHECK_LAUNCH("sa_rows");
    return 0;
}

}  // namespace t2p

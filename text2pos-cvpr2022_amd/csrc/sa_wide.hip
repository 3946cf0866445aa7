// Set-abstraction edge kernel, f16x3 path, SA level 3 (H = C = 256): K-SPLIT CONVERSION, COLUMN-SPLIT PRODUCT.
// (reference: gnn.PointConv(local_nn)(x, (pos, pos[idx]), edge_index), models/pointcloud/pointnet2.py:31-35).
//
// The 256 x 256 weight matrix (hi + lo fp16 planes = 256 KB) is half of a CU's register file: it can be resident exactly once,
// a quarter per wave at one wave per SIMD.  Wave w owns output columns [64 w, 64 w + 64) - 256 registers of weights, 96 MFMAs per
// 32-row tile - so every wave needs the converted activations of ALL rows.  ws_sa2.hip shares them through LDS planes as well,
// but with 8 waves x 32 columns (each LDS operand feeds half as many MFMAs), gathers through VGPRs and a conversion pass whose
// row slices every thread re-derives.  Here:
//   * wave w converts only ITS k-quarter [64 w, 64 w + 64) of every row of the tile: it fetches those 256 B of each row by
//     LDS-DMA into a private 8 KB buffer (8 rows x 128 B per instruction, XOR swizzle on the source address), reads them in
//     the MFMA A-operand layout, forms ReLU(A_j - B_i) and the fp16 hi / lo split in registers (4 of the 16 k-steps: no
//     conversion work is duplicated) and drops the operands - already in operand order - into the shared planes;
//   * after a barrier every wave reads all 16 steps (2 ds_read_b128 per 6 MFMAs) and multiplies them by its 64 columns;
//   * the conversion of tile T+1 runs inside the MFMA loop of tile T, its DMA was issued a tile earlier; float max into the
//     object's LDS accumulator, bias + ReLU at the drain (sa_rows.hip's form).
// All four waves work on every tile: no tile-granular imbalance inside an object.
//
// STATUS: opt-in (t2p_cell_config.tuning bit 4), NOT the default.  Bit-identical run to run and within fp32 rounding of
// ws_sa2.hip, but slower on the 12k-cell step: 32.7 - 40 ms against 27.2 ms (docs/notebook.md, round 3).  T2P_WPROF's cycle
// stamps (one wave, whole launch): 9.1 k cycles per 32-row tile against 6.2 k for ws_sa2.hip's equivalent; the first eight
// steps run at 290 cycles per 6 MFMAs, the eight that also convert the next tile at 540, and 0.8 k + 0.5 k + 0.8 k cycles sit
// in front of barrier X, behind the loop (atomics) and between objects.  One wave per SIMD has nobody to cover those.
// T2P_WABL (development only, results are wrong): 1 = no DMA, 2 = no atomics.  T2P_WPROF = w + 1: wave w of block 0 sums the
// s_memtime differences of the loop's segments over the launch (read with t2p_debug_wprof)
#ifndef T2P_WABL
#define T2P_WABL 0
#endif
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
#define SB() __builtin_amdgcn_sched_barrier(0)
#if T2P_WABL & 2
#define DS_MAXW_STR "; no atomic %0 %1 %2"
#else
#define DS_MAXW_STR "ds_max_f32 %0, %1 offset:%2"
#endif

constexpr int kSubW = 1024;   // objects of a workgroup's range cached at a time (row counts, self-loop bases)

template <int K, int N, int NC>
struct WideCfg {
    static constexpr int NW = 4, NT = 256;
    static constexpr int ND = 2 * NC;
    static constexpr int S16 = K / 16;
    static constexpr int SL = S16 / NW;                  // k-steps a wave converts
    static constexpr int NTW = N / NW / 32;              // 32-column blocks a wave multiplies
    static constexpr int KQ_BYTES = K / NW * 4;          // a wave's piece of a row (256 B)
    static constexpr int LINES = KQ_BYTES / 128;         // ... in 128-byte lines
    static constexpr int RAW_BYTES = LINES * 32 * 128;   // the wave's tile buffer
    static constexpr int MAXR = NC * 33;
    static constexpr int ROWS_CHUNKS = (MAXR * 2 + 1023) / 1024;
    static constexpr int ROWS_BUF = ROWS_CHUNKS * 1024;
    static constexpr int BT_STRIDE = K * 4 + 16;
    static constexpr int CPOS_BUF = 512;                 // [3][NC] floats, filled by two 64-lane 4-byte DMAs
    static constexpr int PLANE_BYTES = S16 * 1024;       // [step][64 lanes][16 B]: one MFMA A operand per lane and step
    static constexpr int ACC_OFF = 0;
    static constexpr int BT_OFF = ACC_OFF + NC * N * 4;
    static constexpr int PL_OFF = BT_OFF + NC * BT_STRIDE;
    static constexpr int RAW_OFF = PL_OFF + 2 * PLANE_BYTES;
    static constexpr int ROWS_OFF = RAW_OFF + NW * RAW_BYTES;
    static constexpr int CPOS_OFF = ROWS_OFF + 3 * ROWS_BUF;
    static constexpr int NR_OFF = CPOS_OFF + 2 * CPOS_BUF;
    static constexpr int SB_OFF = NR_OFF + kSubW * 2;
    static constexpr int DSTL_OFF = SB_OFF + kSubW * 4;
    static constexpr size_t lds_bytes() { return (size_t)DSTL_OFF + NW * 64; }
    static_assert(K == 256 && N == 256 && NC == 32, "built for SA level 3");
    static_assert(LINES == 2 && SL == 4 && NTW == 2, "shape");
};

template <int SEL>
__device__ __forceinline__ float sub_half_w(float v, fp16x2 h) {   // v - (float)h[SEL] in one VALU op (exact)
    float r;
    if constexpr (SEL == 0)
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(v));
    else
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(v));
    return r;
}

__device__ __forceinline__ void lds_barrier_w() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void wait_all_vm_w() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

#ifndef T2P_WPROF
#define T2P_WPROF 0
#endif
#if T2P_WPROF
// cycle accounting of one wave (block 0, wave T2P_WPROF - 1): sums of the tile-loop segments over the launch
__device__ unsigned long long t2p_wprof_sums[16];
#define WPROF_DECL unsigned long long wp_t = 0, wp_sum[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; const unsigned long long wp_begin = __builtin_amdgcn_s_memtime()
#define WPROF_START() wp_t = __builtin_amdgcn_s_memtime()
#define WPROF_MARK(i)                                                  \
    do {                                                               \
        const unsigned long long n_ = __builtin_amdgcn_s_memtime();    \
        wp_sum[i] += n_ - wp_t;                                        \
        wp_t = n_;                                                     \
    } while (0)
#else
#define WPROF_DECL
#define WPROF_START()
#define WPROF_MARK(i)
#endif

template <int K, int N, int NC>
__global__ __launch_bounds__(256, 1) void k_sa_wide(SaParams p) {
    using C = WideCfg<K, N, NC>;
    constexpr int NW = C::NW;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    int* acc_lds = (int*)(lds + C::ACC_OFF);
    uint16_t* nr = (uint16_t*)(lds + C::NR_OFF);
    int* sbase = (int*)(lds + C::SB_OFF);
    WPROF_DECL;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, rr = lane & 31;
    const uint32_t rawb = (uint32_t)(C::RAW_OFF + wave * C::RAW_BYTES);
    const uint32_t dstl_addr = (uint32_t)(C::DSTL_OFF + wave * 64);
    const uint32_t pl_lane = (uint32_t)(C::PL_OFF + lane * 16);     // this lane's 16 bytes of a step's operand (hi plane)

    // ---- stationary weights: columns [64 w, 64 w + 64), all K, hi / lo planes, natural k order -------------------------------
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
                const int idx = ((((wave * C::NTW + nt) * C::S16 + step_) * 2 + half_) * 32) + rr;
                w_hi[nt][s] = __builtin_bit_cast(half8, wp[idx]);
                w_lo[nt][s] = __builtin_bit_cast(half8, wp[PLANE_U4 + idx]);
            }
    }
    constexpr f32x16 kZero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x4 bias4;         // this thread's four output columns in the drain (column quad = tid % (N / 4))
    bias4 = *(const f32x4*)(p.bias + (tid % (N / 4)) * 4);
    // position rows of the layer-1 weights, this thread's column quad (centroid table build)
    constexpr int QPR = K / 4, CGS = C::NT / QPR, CPT = NC / CGS;
    const int cq = tid % QPR, cg = tid / QPR;
    const f32x4 wq0 = *(const f32x4*)(p.wp + cq * 4), wq1 = *(const f32x4*)(p.wp + K + cq * 4),
                wq2 = *(const f32x4*)(p.wp + 2 * K + cq * 4);

    // ---- per-lane constants --------------------------------------------------------------------------------------------------
    // tile buffer of the wave: [line u][row][128 B], chunk c of a line at position c ^ ((row >> 1) & 7); local step sl reads
    // line sl >> 1, chunks 4 (sl & 1) + 2 h + j
    uint32_t rd[C::SL][2];
#pragma unroll
    for (int sl = 0; sl < C::SL; sl++)
#pragma unroll
        for (int j = 0; j < 2; j++)
            rd[sl][j] = rawb + (uint32_t)((sl >> 1) * 4096 + rr * 128 + (((4 * (sl & 1) + 2 * h + j) ^ ((rr >> 1) & 7)) * 16));
    uint32_t dma_sel[4], dma_chunk[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int r = 8 * q + (lane >> 3);
        dma_sel[q] = (uint32_t)(r * 4);
        dma_chunk[q] = (uint32_t)((((lane & 7) ^ ((r >> 1) & 7))) * 16 + wave * C::KQ_BYTES);
    }

    for (int i = tid; i < NC * N; i += C::NT) acc_lds[i] = (int)0xFF800000;   // -inf
    int gtop = 0;         // fp16-range guard: this lane's maximum (bit pattern, before out_scale) of the drained outputs; reduced over
                          // the wave once, at the end (six ds_bpermute round trips per drain otherwise)
    const int g_begin = p.bounds_ws[blockIdx.x], g_end = p.bounds_ws[blockIdx.x + 1];

    auto dma16 = [&](const void* base, uint32_t voff, uint32_t lds_off) {
        __builtin_amdgcn_global_load_lds((gl_void*)((const char*)base + voff), (lds_void*)(lds + lds_off), 16, 0, 0);
    };
    auto dma_rows = [&](int g) {   // row list of object g -> rows buffer g % 3 (1 KB pieces over the waves)
        const char* src = (const char*)(p.rows + (int64_t)g * C::MAXR);
        const uint32_t dst = (uint32_t)(C::ROWS_OFF + (g % 3) * C::ROWS_BUF);
#pragma unroll
        for (int c = 0; c < C::ROWS_CHUNKS; c++) {
            if (c % NW != wave) continue;
            uint32_t off = (uint32_t)(c * 1024 + lane * 16);
            if (off + 16 > (uint32_t)(C::MAXR * 2)) off = 0;
            dma16(src, off, dst + c * 1024);
        }
    };
    auto dma_cpos = [&](int g) {   // centroid positions of object g -> cpos buffer g & 1, [3][NC] (+ unused tail lanes)
        if (wave != 3) return;
        const uint32_t dst = (uint32_t)(C::CPOS_OFF + (g & 1) * C::CPOS_BUF);
        // one 4-byte DMA for all 3 NC = 96 coordinates does not fit a wave: two instructions of 64 lanes, lane = flat index
#pragma unroll
        for (int b = 0; b < 2; b++) {
            int i = b * 64 + lane;
            i = i < 3 * NC ? i : 0;
            const float* src = p.out + ((int64_t)g * NC + i % NC) * (int64_t)p.ldo + N + i / NC;
            __builtin_amdgcn_global_load_lds((gl_void*)src, (lds_void*)(lds + dst + b * 256), 4, 0, 0);
        }
    };

    for (int ga = g_begin; ga < g_end; ga += kSubW) {
        const int cnt = (g_end - ga) < kSubW ? (g_end - ga) : kSubW;
        wait_all_vm_w();
        __syncthreads();
        for (int i = tid; i < cnt; i += C::NT) {
            const int g = ga + i;
            nr[i] = p.n_rows[g];
            const int first = p.first[g];
            sbase[i] = first * C::ND + (g - first) * NC;
        }
        dma_rows(ga);
        if (cnt > 1) dma_rows(ga + 1);
        dma_cpos(ga);
        wait_all_vm_w();
        __syncthreads();

        auto rows_of = [&](int gi) { return __builtin_amdgcn_readfirstlane((int)nr[gi]); };
        auto tile_meta = [&](int gi, int r0, int n) -> uint32_t {
            int idx = r0 + rr;
            idx = idx < n ? idx : n - 1;
            const uint16_t* rows_l = (const uint16_t*)(lds + C::ROWS_OFF + ((ga + gi) % 3) * C::ROWS_BUF);
            return (uint32_t)rows_l[idx];
        };
        auto row_byte = [&](int gi, uint32_t sb0, uint32_t m) -> uint32_t {
            const uint32_t src = m & 0xFFu, d = m >> 8;
            const uint32_t srow = (d & 0x80u) ? (sb0 + src) : ((uint32_t)(ga + gi) * (uint32_t)C::ND + src);
            return srow * (uint32_t)(K * 4);
        };
        // DMA of a tile's k-quarter: 2 lines x 4 instructions of 8 rows
        auto issue_tile = [&](uint32_t rowbyte) {
            if constexpr (T2P_WABL & 1) return;
            uint32_t voff[4];
#pragma unroll
            for (int q = 0; q < 4; q++)
                voff[q] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)dma_sel[q], (int)rowbyte) + dma_chunk[q];
#pragma unroll
            for (int u = 0; u < C::LINES; u++)
#pragma unroll
                for (int q = 0; q < 4; q++) dma16(p.A, voff[q] + (uint32_t)(u * 128), rawb + (uint32_t)(u * 4096 + q * 1024));
        };
        auto flush = [&](int g) {
            float* o = p.out + (int64_t)g * NC * (int64_t)p.ldo;
            int top = gtop;
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
                    const float r = fmaxf(raw[e] + bias4[e], 0.f);
                    const int bits = __float_as_int(r);
                    top = bits > top ? bits : top;
                    v[e] = r * p.out_scale;
                }
                *(f32x4*)(o + c * (int64_t)p.ldo + c4 * 4) = v;
                *(i32x4*)a = i32x4{(int)0xFF800000, (int)0xFF800000, (int)0xFF800000, (int)0xFF800000};
            }
            gtop = top;
        };
        auto build_b = [&](int g) {
            const float* cp = (const float*)(lds + C::CPOS_OFF + (g & 1) * C::CPOS_BUF);
#pragma unroll
            for (int i = 0; i < CPT; i++) {
                const int c = cg + CGS * i;
                const float px = cp[c], py = cp[NC + c], pz = cp[2 * NC + c];
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float a = px * wq0[e];
                    a = fmaf(py, wq1[e], a);
                    a = fmaf(pz, wq2[e], a);
                    v[e] = a;
                }
                *(f32x4*)(lds + C::BT_OFF + c * C::BT_STRIDE + cq * 16) = v;
            }
        };
        // conversion of the wave's k-quarter of the tile whose pieces sit in its buffer: local step sl -> operand registers
        auto convert = [&](const f32x4 (&x)[2], const f32x4 (&b)[2], half8& oh, half8& ol) {
            uint32_t nh[4], nl[4];
#pragma unroll
            for (int pr = 0; pr < 4; pr++) {
                const int j = pr >> 1, e0 = (pr & 1) * 2;
                const float v0 = fmaxf(x[j][e0] - b[j][e0], 0.f), v1 = fmaxf(x[j][e0 + 1] - b[j][e0 + 1], 0.f);
                const fp16x2 hh = __builtin_amdgcn_cvt_pkrtz(v0, v1);
                const fp16x2 ll = __builtin_amdgcn_cvt_pkrtz(sub_half_w<0>(v0, hh), sub_half_w<1>(v1, hh));
                nh[pr] = __builtin_bit_cast(uint32_t, hh);
                nl[pr] = __builtin_bit_cast(uint32_t, ll);
            }
            oh = __builtin_bit_cast(half8, u32x4{nh[0], nh[1], nh[2], nh[3]});
            ol = __builtin_bit_cast(half8, u32x4{nl[0], nl[1], nl[2], nl[3]});
        };
        // raw + table reads of local step sl (inline asm: see sa_rows.hip on hipcc's vmcnt(0) in front of ordinary LDS reads)
        auto load_step = [&](int sl, uint32_t brow, f32x4 (&x)[2], f32x4 (&b)[2]) {
            const uint32_t ba = brow + (uint32_t)(sl * 64);
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %6 offset:16\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(x[0]), "=&v"(x[1]), "=&v"(b[0]), "=&v"(b[1]) : "v"(rd[sl][0]), "v"(rd[sl][1]), "v"(ba) : "memory");
        };
        auto write_planes = [&](const half8 (&ph)[C::SL], const half8 (&pl)[C::SL]) {
#pragma unroll
            for (int sl = 0; sl < C::SL; sl++) {
                const uint32_t a = pl_lane + (uint32_t)((wave * C::SL + sl) * 1024);
                asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(ph[sl]) : "memory");
                asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(a), "v"(pl[sl]), "n"(C::PLANE_BYTES) : "memory");
            }
        };

        build_b(ga);
        if (cnt > 2) dma_rows(ga + 2);
        if (cnt > 1) dma_cpos(ga + 1);
        lds_barrier_w();

        bool cur_fetched = false;     // the pieces of the next tile to convert are in flight / landed in the wave's buffer
        uint32_t m_cur = 0;
        // look-ahead: row metadata and DMA addresses of tile T+1, prepared inside the loop of tile T-1
        uint32_t m_nxt = 0, vn[4] = {0u, 0u, 0u, 0u};
        bool vn_valid = false;
        bool u0_sent = false;         // the first line of tile T+1's pieces went out during tile T-1
        auto decode_voff = [&](uint32_t rowbyte, uint32_t (&voff)[4]) {
#pragma unroll
            for (int q = 0; q < 4; q++)
                voff[q] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)dma_sel[q], (int)rowbyte) + dma_chunk[q];
        };
        auto row_addr = [&](const uint2 (&f4)[4], int e) -> uint32_t {
            const uint32_t pair = (e & 2) ? f4[e >> 2].y : f4[e >> 2].x;
            return (uint32_t)(C::ACC_OFF + (wave * 64 + rr) * 4) + ((e & 1) ? (pair >> 16) : (pair & 0xFFFFu));
        };

        for (int gi = 0; gi < cnt; gi++) {
            const int n_g = rows_of(gi), n_g1 = gi + 1 < cnt ? rows_of(gi + 1) : 0, n_g2 = gi + 2 < cnt ? rows_of(gi + 2) : 0;
            const uint32_t sb_g = (uint32_t)__builtin_amdgcn_readfirstlane(sbase[gi]);
            const uint32_t sb_g1 = gi + 1 < cnt ? (uint32_t)__builtin_amdgcn_readfirstlane(sbase[gi + 1]) : 0u;
            const uint32_t sb_g2 = gi + 2 < cnt ? (uint32_t)__builtin_amdgcn_readfirstlane(sbase[gi + 2]) : 0u;
            half8 ph[C::SL], pl[C::SL];     // converted operands of the wave's k-quarter (tile to be multiplied next)
            if (n_g > 0) {
                // ---- first tile of the object: fetched (by the previous object's last tile) or fetched now; converted here ----
                if (!cur_fetched) {
                    m_cur = tile_meta(gi, 0, n_g);
                    issue_tile(row_byte(gi, sb_g, m_cur));
                    vn_valid = false;
                    u0_sent = false;
                }
                wait_all_vm_w();
                {
                    const uint32_t dl = (m_cur >> 8) & 127u;
                    const uint32_t brow = (uint32_t)C::BT_OFF + dl * (uint32_t)C::BT_STRIDE + (uint32_t)(wave * C::KQ_BYTES + h * 32);
#pragma unroll
                    for (int sl = 0; sl < C::SL; sl++) {
                        f32x4 x[2], b[2];
                        load_step(sl, brow, x, b);
                        convert(x, b, ph[sl], pl[sl]);
                    }
                }
                cur_fetched = false;
            }
            for (int r0 = 0; r0 < n_g; r0 += 32) {
                WPROF_START();
                // tile T = (gi, r0): its operands are in ph / pl; m_cur its row metadata
                const bool chain = r0 + 32 < n_g;
                const bool nxt_ok = chain || (gi + 1 < cnt && n_g1 > 0);
                const int gi_n = chain ? gi : gi + 1, r0_n = chain ? r0 + 32 : 0, n_n = chain ? n_g : n_g1;
                const uint32_t sb_n = chain ? sb_g : sb_g1;
                if (nxt_ok && !vn_valid) {     // no look-ahead yet (first tile of the range, or behind a tile that had none)
                    m_nxt = tile_meta(gi_n, r0_n, n_n);
                    decode_voff(row_byte(gi_n, sb_n, m_nxt), vn);
                }
                // tile T+2 (its metadata is decoded inside this tile's loop)
                const bool chain2 = nxt_ok && r0_n + 32 < n_n;
                const int n_after = gi_n == gi ? n_g1 : n_g2;            // rows of the object behind T+1's
                const bool n2_ok = nxt_ok && (chain2 || (gi_n + 1 < cnt && n_after > 0));
                const int gi_2 = chain2 ? gi_n : gi_n + 1, r0_2 = chain2 ? r0_n + 32 : 0, n_2 = chain2 ? n_n : n_after;
                const uint32_t sb_2 = chain2 ? sb_n : (gi_n == gi ? sb_g1 : sb_g2);
                uint32_t m_n2 = 0, vn2[4] = {0u, 0u, 0u, 0u};
                uint32_t m_n2_addr = 0;
                if (n2_ok) {
                    int idx = r0_2 + rr;
                    idx = idx < n_2 ? idx : n_2 - 1;
                    m_n2_addr = (uint32_t)(C::ROWS_OFF + ((ga + gi_2) % 3) * C::ROWS_BUF + idx * 2);
                }

                const uint32_t dl = (m_cur >> 8) & 127u;
                asm volatile("ds_write_b16 %0, %1" ::"v"(dstl_addr + (uint32_t)(rr * 2)), "v"(dl * (uint32_t)(N * 4)) : "memory");
                // planes of tile T: written behind barrier Y of the previous tile (or the object's opening barrier)
                write_planes(ph, pl);
                WPROF_MARK(0);
                lds_barrier_w();                                   // X: every wave's quarter of the operands is in place
                WPROF_MARK(1);
                uint2 four[4];
                {
                    const uint32_t a4 = dstl_addr + (uint32_t)(h * 8);
                    asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:16\n\tds_read_b64 %2, %4 offset:32\n\t"
                                 "ds_read_b64 %3, %4 offset:48"
                                 : "=&v"(four[0]), "=&v"(four[1]), "=&v"(four[2]), "=&v"(four[3]) : "v"(a4) : "memory");
                }
                f32x16 acc[C::NTW];
                half8 a_hi, a_lo, n_hi, n_lo;
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:%3\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(a_hi), "=&v"(a_lo) : "v"(pl_lane), "n"(C::PLANE_BYTES) : "memory");
                WPROF_MARK(2);
                const uint32_t dl_n = (m_nxt >> 8) & 127u;
                const uint32_t brow_n = (uint32_t)C::BT_OFF + dl_n * (uint32_t)C::BT_STRIDE + (uint32_t)(wave * C::KQ_BYTES + h * 32);
                f32x4 x[2], b[2];          // raw row pieces / table entries of the local step being converted
                uint32_t nh[4], nl[4];
                // ---- 16 steps: [operand reads of step s+1 (+ raw reads of a local step)] [2 MFMAs] [DMA piece / pending atomics /
                // look-ahead decode] [4 MFMAs with the conversion in four chunks]; the order is pinned (one wave per SIMD) -------------
#pragma unroll
                for (int s = 0; s < C::S16; s++) {
                    if (s + 1 < C::S16) {
                        const uint32_t a = pl_lane + (uint32_t)((s + 1) * 1024);
                        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:%3"
                                     : "=&v"(n_hi), "=&v"(n_lo) : "v"(a), "n"(C::PLANE_BYTES) : "memory");
                    }
                    const bool conv_read = chain && s >= 8 && (s & 1) == 0;      // local step (s - 8) / 2: reads now, VALU in step s + 1
                    const bool conv_math = chain && s >= 9 && (s & 1) == 1;
                    if (conv_read) {
                        if (s == 8) {     // first line in: behind it went the second line's four pieces, or (first line sent in
                                          // this tile, alternating with the second) only the second line's last piece
                            if (u0_sent) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                            else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                        }
                        if (s == 12) wait_all_vm_w();
                        const int sl = (s - 8) >> 1;
                        const uint32_t ba = brow_n + (uint32_t)(sl * 64);
                        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %6 offset:16"
                                     : "=&v"(x[0]), "=&v"(x[1]), "=&v"(b[0]), "=&v"(b[1]) : "v"(rd[sl][0]), "v"(rd[sl][1]), "v"(ba) : "memory");
                    }
                    if (s == 2 && n2_ok) asm volatile("ds_read_u16 %0, %1" : "=v"(m_n2) : "v"(m_n2_addr) : "memory");
                    SB();
#pragma unroll
                    for (int nt = 0; nt < C::NTW; nt++) acc[nt] = MFMA16(a_hi, w_hi[nt][s], s == 0 ? kZero16 : acc[nt]);
                    SB();
                    if constexpr (!(T2P_WABL & 1)) {
                        // pieces of the rows: the buffer's first line (k 0..31 of the quarter) is free once step 10's reads are
                        // in, the second after step 14's; a tile's first line goes out before its second (the counted waits)
                        if (s < 4 && nxt_ok) {
                            if (!u0_sent) dma16(p.A, vn[s], rawb + (uint32_t)(s * 1024));
                            dma16(p.A, vn[s] + 128u, rawb + (uint32_t)(4096 + s * 1024));
                        }
                        if (s >= 12 && chain && n2_ok) dma16(p.A, vn2[s - 12], rawb + (uint32_t)((s - 12) * 1024));
                    }
                    if (s == 4 && n2_ok) decode_voff(row_byte(gi_2, sb_2, m_n2), vn2);
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        if (conv_math) {     // pair c of the local step read in the previous step
                            const int j = c >> 1, e0 = (c & 1) * 2;
                            const float v0 = fmaxf(x[j][e0] - b[j][e0], 0.f), v1 = fmaxf(x[j][e0 + 1] - b[j][e0 + 1], 0.f);
                            const fp16x2 hh = __builtin_amdgcn_cvt_pkrtz(v0, v1);
                            const fp16x2 ll = __builtin_amdgcn_cvt_pkrtz(sub_half_w<0>(v0, hh), sub_half_w<1>(v1, hh));
                            nh[c] = __builtin_bit_cast(uint32_t, hh);
                            nl[c] = __builtin_bit_cast(uint32_t, ll);
                        }
                        if (s + 1 < C::S16) {
                            if (c < 2) acc[c] = MFMA16(a_hi, w_lo[c][s], acc[c]);
                            else acc[c - 2] = MFMA16(a_lo, w_hi[c - 2][s], acc[c - 2]);
                        } else {
                            // last step: column block 0 finishes first (its atomics go under block 1's MFMAs)
                            if (c == 0) acc[0] = MFMA16(a_hi, w_lo[0][s], acc[0]);
                            else if (c == 1) acc[0] = MFMA16(a_lo, w_hi[0][s], acc[0]);
                            else if (c == 2) acc[1] = MFMA16(a_hi, w_lo[1][s], acc[1]);
                            else acc[1] = MFMA16(a_lo, w_hi[1][s], acc[1]);
                        }
                        SB();
                        if (s + 1 == C::S16 && c == 1) {
                            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(four[0]), "+v"(four[1]), "+v"(four[2]), "+v"(four[3])::"memory");
                        }
                        if (s + 1 == C::S16 && c >= 2) {
                            if (c == 2) asm volatile("s_nop 15\n\ts_nop 15" : "+a"(acc[0])::"memory");
                            if constexpr (!(T2P_WABL & 2)) {
#pragma unroll
                                for (int i = (c - 2) * 8; i < (c - 2) * 8 + 8; i++)
                                    asm volatile(DS_MAXW_STR ::"v"(row_addr(four, i)), "a"(acc[0][i]), "n"(0) : "memory");
                            }
                            SB();
                        }
                    }
                    if (conv_math) {
                        ph[(s - 9) >> 1] = __builtin_bit_cast(half8, u32x4{nh[0], nh[1], nh[2], nh[3]});
                        pl[(s - 9) >> 1] = __builtin_bit_cast(half8, u32x4{nl[0], nl[1], nl[2], nl[3]});
                    }
                    // one wait for everything this step requested from LDS; the registers pass through it
                    if (conv_read)
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(n_hi), "+v"(n_lo), "+v"(x[0]), "+v"(x[1]), "+v"(b[0]), "+v"(b[1])::"memory");
                    else if (s == 2 && n2_ok)
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(n_hi), "+v"(n_lo), "+v"(m_n2)::"memory");
                    else if (s + 1 < C::S16)
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(n_hi), "+v"(n_lo)::"memory");
                    if (s + 1 < C::S16) {
                        a_hi = n_hi;
                        a_lo = n_lo;
                    }
                    if (s == 7) WPROF_MARK(3);
                    if (s == 11) WPROF_MARK(4);
                    if (s == 15) WPROF_MARK(5);
                }
                asm volatile("s_nop 15\n\ts_nop 15" : "+a"(acc[1])::"memory");
                if constexpr (!(T2P_WABL & 2)) {
#pragma unroll
                    for (int i = 0; i < 16; i++)
                        asm volatile(DS_MAXW_STR ::"v"(row_addr(four, i)), "a"(acc[1][i]), "n"(128) : "memory");
                }
                WPROF_MARK(6);
                lds_barrier_w();                                   // Y: the planes have been read by everybody
                WPROF_MARK(7);
#if T2P_WPROF
                wp_sum[8] += 1;
#endif
                cur_fetched = nxt_ok && !chain;
                m_cur = m_nxt;
                m_nxt = m_n2;
#pragma unroll
                for (int q = 0; q < 4; q++) vn[q] = vn2[q];
                vn_valid = n2_ok;
                u0_sent = chain && n2_ok;
            }
            // ---- object gi is complete -----------------------------------------------------------------------------------------
            // (barrier Y of the last tile = all atomics of the object are in the accumulator; an object without rows: none needed)
            flush(ga + gi);
            if (gi + 1 < cnt) {
                build_b(ga + gi + 1);
                // the lists / positions of the objects ahead: their DMA must not wait behind this wave's tile pieces for long,
                // and must have landed before another wave reads them: requested here, waited for (vmcnt(0)) by the next object's
                // opening wait or its first in-loop wait, both in front of barriers every reader passes later
                if (gi + 3 < cnt) dma_rows(ga + gi + 3);
                if (gi + 2 < cnt) dma_cpos(ga + gi + 2);
            }
            if (n_g1 == 0 || !cur_fetched) wait_all_vm_w();
            lds_barrier_w();                                        // B: accumulator cleared, next table in place
        }
    }
#if T2P_WPROF
    if (blockIdx.x == 0 && wave == T2P_WPROF - 1 && lane == 0) {
        for (int i = 0; i < 9; i++) atomicAdd(&t2p_wprof_sums[i], wp_sum[i]);
        atomicAdd(&t2p_wprof_sums[9], __builtin_amdgcn_s_memtime() - wp_begin);
    }
#endif
    wait_all_vm_w();
    uint32_t gbits = 0;
    guard_track_bits(gbits, gtop);
    if (p.amax_out != nullptr && lane == 0 && gbits != 0u)
        atomicMax(p.amax_out, __float_as_uint(__uint_as_float(gbits) * p.out_scale));
}

}  // namespace

bool sa_wide_selected(int H, int Cout, const SaParams& p) {
    return H == 256 && Cout == 256 && p.W_x3 != nullptr && p.wp != nullptr && (p.plan & 4);
}

int sa_wide_launch_shape(int64_t n_obj, int* tile_rows, int* n_wg) {
    int n = num_cus();
    if (n > 1024) n = 1024;
    if (n > n_obj) n = (int)n_obj;
    *tile_rows = 32;
    *n_wg = n;
    return 0;
}

int launch_sa_wide(int H, int Cout, const SaParams& p, hipStream_t st) {
    if (!(H == 256 && Cout == 256 && p.n_cent == 32 && p.n_dense == 64 && p.wp && p.W_x3)) {
        set_error("sa_wide: built for SA level 3 (H = C = 256, 32 centroids of 64 points, f16x3, LDS centroid table)");
        return T2P_E_UNSUPPORTED;
    }
    using C = WideCfg<256, 256, 32>;
    auto kern = k_sa_wide<256, 256, 32>;
    T2P_TRY(reserve_lds((const void*)kern, C::lds_bytes(), "sa_wide"));
    if (p.n_obj <= 0) return 0;
    T2P_CHECK_ARG(p.n_obj < (1 << 30) && p.n_obj * p.n_dense * (int64_t)H * 4 < 0xffffffffLL,
                  "sa_wide: chunk too large for 32-bit table offsets");
    T2P_CHECK_ARG((((uintptr_t)p.A | (uintptr_t)p.rows | (uintptr_t)p.out | (uintptr_t)p.W_x3) & 15) == 0 && p.ldo % 4 == 0,
                  "sa_wide: tables must be 16-byte aligned");
    int tr, n_wg;
    sa_wide_launch_shape(p.n_obj, &tr, &n_wg);
    if (!p.balanced) T2P_TRY(launch_sa_balance(p, tr, n_wg, st));
    ProfScope ps_("ws_edge_sa_k256_n256", st);
    hipLaunchKernelGGL(kern, dim3(n_wg), dim3(C::NT), C::lds_bytes(), st, p);
    T2P_CHECK_LAUNCH("sa_wide");
    return 0;
}

}  // namespace t2p

#if T2P_WPROF
extern "C" void t2p_debug_wprof(unsigned long long* out, int reset) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(t2p::t2p_wprof_sums), sizeof(unsigned long long) * 16);
    if (reset) {
        unsigned long long z[16] = {};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(t2p::t2p_wprof_sums), z, sizeof(z));
    }
}
#endif

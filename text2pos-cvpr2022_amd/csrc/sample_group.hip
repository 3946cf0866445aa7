// Fused farthest-point sampling + ball query for the three PointNet++ set-abstraction levels of one object.
//
// Replaces (reference call sites): gnn.fps  models/pointcloud/pointnet2.py:26  and gnn.radius
// models/pointcloud/pointnet2.py:28-30 (max_num_neighbors = 32) for sa1/sa2/sa3 (ratio 0.5, r = 0.2/0.3/0.4).
//
// Both operators depend on the xyz coordinates only, and level l+1 works on the FPS subset of level l, so all
// three levels are computed by ONE wavefront per object with the coordinates staged once in LDS
// (256 x 3 fp32 = 3 KiB).  Output is the compact uint8 group table (GroupTables in t2p_common.h): 7.4 KiB per
// object instead of ~90 KiB of int64 COO edges.
//
// Pinned semantics (SURVEY.md 0.6 / oracle/primitives.c): FPS starts at local point 0, arg-max ties -> lowest
// index; ball query keeps the first <= 32 in-range dense points in ascending index, strict d2 < r*r;
// d2 = (dx*dx + dy*dy) + dz*dz in fp32 with no FMA contraction.
#include "t2p_common.h"

namespace t2p {

#pragma clang fp contract(off)

namespace {

// T2P_FPS_PK: the distance pass of the production FPS / ball-query loop on packed fp32 operations (A/B switch; same bits either way)
#ifndef T2P_FPS_PK
#define T2P_FPS_PK 1
#endif

constexpr int kMaxPts = 256;
constexpr int kMaxNbr = 32;

__device__ __forceinline__ float dist2(float ax, float ay, float az, float bx, float by, float bz) {
#pragma clang fp contract(off)
    float dx = ax - bx, dy = ay - by, dz = az - bz;
    float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    float s = xx + yy;
    return s + zz;
}

// Cross-lane reductions on the VALU (DPP) instead of LDS-routed shuffles: the FPS loop is a chain of ~220 dependent
// arg-max steps per object, so the reduction latency is what bounds it.
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
// quad_perm [1,0,3,2] = 0xB1, quad_perm [2,3,0,1] = 0x4E, row_half_mirror = 0x141, row_mirror = 0x140:
// after the four steps every lane of a 16-lane row holds the row's result; rows are combined through SGPRs.
__device__ __forceinline__ float wave_max_f(float v) {
    v = fmaxf(v, __int_as_float(dpp_i<0xB1>(__float_as_int(v))));
    v = fmaxf(v, __int_as_float(dpp_i<0x4E>(__float_as_int(v))));
    v = fmaxf(v, __int_as_float(dpp_i<0x141>(__float_as_int(v))));
    v = fmaxf(v, __int_as_float(dpp_i<0x140>(__float_as_int(v))));
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(a, b), fmaxf(c, d));
}
// The same for NON-NEGATIVE floats through their bit patterns (they order like unsigned integers): fmaxf() quiets NaNs, which
// costs a canonicalising v_max x, x in front of every maximum - 3 instructions + an s_nop per DPP step instead of 1.
__device__ __forceinline__ uint32_t wave_max_u(uint32_t v) {
    v = max(v, (uint32_t)dpp_i<0xB1>((int)v));
    v = max(v, (uint32_t)dpp_i<0x4E>((int)v));
    v = max(v, (uint32_t)dpp_i<0x141>((int)v));
    v = max(v, (uint32_t)dpp_i<0x140>((int)v));
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    return max(max(a, b), max(c, d));
}
__device__ __forceinline__ int wave_min_i(int v) {
    v = min(v, dpp_i<0xB1>(v));
    v = min(v, dpp_i<0x4E>(v));
    v = min(v, dpp_i<0x141>(v));
    v = min(v, dpp_i<0x140>(v));
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return min(min(a, b), min(c, d));
}
// wave-wide arg-max of (d, idx): larger d wins, equal d -> smaller idx.  Result uniform across the wave.
__device__ __forceinline__ void wave_argmax(float& d, int& idx) {
    const float m = wave_max_f(d);
    idx = wave_min_i(d == m ? idx : 0x7fffffff);
    d = m;
}

// One level: FPS of n_c samples among the n_d points in (px,py,pz) [LDS], then ball query.
// sel (LDS, n_c bytes) receives the FPS indices; the sampled coordinates are written to (qx,qy,qz).
template <int PPL>  // points per lane: ceil(n_d / 64)
__device__ void level(const float* px, const float* py, const float* pz, int n_d, int n_c, float r2,
                      uint8_t* sel, float* qx, float* qy, float* qz, uint8_t* nbr_lds, uint8_t* cnt_lds,
                      uint16_t* __restrict__ rows_out, int self_loops, int* n_rows_out) {
    const int lane = threadIdx.x;
    // Lane l owns the CONTIGUOUS points [l*PPL, (l+1)*PPL): ascending index order is then lane-major, so "ties -> lowest
    // index" is "lowest lane, then lowest j" (one ballot + s_ff1 instead of a second wave reduction), and a hit's rank
    // in the ascending neighbour list is (hits in lower lanes) + (own earlier hits).
    float x[PPL], y[PPL], z[PPL], mind[PPL];
#pragma unroll
    for (int j = 0; j < PPL; j++) {
        int i = lane * PPL + j;
        bool v = i < n_d;
        x[j] = v ? px[i] : 0.f;
        y[j] = v ? py[i] : 0.f;
        z[j] = v ? pz[i] : 0.f;
        mind[j] = INFINITY;
    }
    // FPS and ball query share their distance pass: iteration c measures every dense point against centroid c (chosen
    // by the previous iteration), which is both the FPS update for the choice of centroid c+1 and the ball-query test
    // of centroid c -- same dist2() call, so the same bits as two separate passes.
    // The hits are emitted as the object's compact edge-row list (sorted by centroid), one u16 per row: low byte =
    // source (dense index; for a self-loop row: the centroid index), high byte = centroid | 0x80 if self loop.
    int cur = 0;
    int base = 0;
    for (int c = 0; c < n_c; c++) {
        const float cx = px[cur], cy = py[cur], cz = pz[cur];
        if (lane == 0) {
            sel[c] = (uint8_t)cur;
            qx[c] = cx;
            qy[c] = cy;
            qz[c] = cz;
        }
        float bd = -1.f;
        int bi = 0x7fffffff;
        bool hit[PPL];
        int lower = 0;   // hits of this centroid in lower lanes
        int count = 0;   // all hits (uniform)
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            const int i = lane * PPL + j;
            const bool in = i < n_d;
            const float d = dist2(x[j], y[j], z[j], cx, cy, cz);
            if (in) {
                mind[j] = d < mind[j] ? d : mind[j];
                if (mind[j] > bd) { bd = mind[j]; bi = i; }
            }
            hit[j] = in && (d < r2);
            const unsigned long long m = __ballot(hit[j]);
            lower = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, lower));
            count += __popcll(m);
        }
        int pos = lower;
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            if (hit[j] && pos < kMaxNbr) {
                const int i = lane * PPL + j;
                if (nbr_lds) nbr_lds[c * kMaxNbr + pos] = (uint8_t)i;
                if (rows_out) rows_out[base + pos] = (uint16_t)((c << 8) | i);
            }
            pos += hit[j] ? 1 : 0;
        }
        const int kept = count < kMaxNbr ? count : kMaxNbr;
        if (lane == 0) {
            if (nbr_lds) cnt_lds[c] = (uint8_t)kept;
            if (self_loops && rows_out) rows_out[base + kept] = (uint16_t)(((c | 0x80) << 8) | c);
        }
        base += kept + (self_loops ? 1 : 0);
        if (c + 1 < n_c) {  // uniform
            const float mx = wave_max_f(bd);
            const unsigned long long tie = __ballot(bd == mx);
            cur = __builtin_amdgcn_readlane(bi, (int)__builtin_ctzll(tie));
        }
    }
    // pad the list to a multiple of 4 rows with the "no row" marker, so that consumers may fetch 4 rows per load
    if (rows_out && lane < 4 && base + lane < n_c * (kMaxNbr + 1)) rows_out[base + lane] = 0xFFFF;
    *n_rows_out = base;
    __syncthreads();
}

// The production shape of level(): every lane owns PPL valid points (n_d == 64 * PPL), the row list is wanted, the
// neighbour table is not.  Same arithmetic and tie-breaking, ~1/3 fewer instructions per FPS step (the kernel is bound by
// instruction issue: ~230 instructions per step of the 256-point level in the generic form):
//   * no validity masks, no NULL checks of the optional outputs;
//   * the 32-neighbour cap is tested once per centroid (uniform) instead of once per point;
//   * the lane's farthest point is found with max() and located only afterwards (lowest j that equals the wave maximum:
//     the first point a strict '>' scan would have kept).
template <int PPL>
__device__ void level_fast(const float* px, const float* py, const float* pz, int n_c, float r2, uint8_t* sel, float* qx,
                           float* qy, float* qz, uint16_t* __restrict__ rows_out, int self_loops, int* n_rows_out) {
    const int lane = threadIdx.x;
    const int i0 = lane * PPL;
    float x[PPL], y[PPL], z[PPL];
    uint32_t mind[PPL];   // running minimum of the squared distances, as bit patterns (non-negative floats order like integers)
#pragma unroll
    for (int j = 0; j < PPL; j++) {
        x[j] = px[i0 + j];
        y[j] = py[i0 + j];
        z[j] = pz[i0 + j];
        mind[j] = 0x7f800000u;   // +inf
    }
    int cur = 0;
    int base = 0;
    for (int c = 0; c < n_c; c++) {
        const float cx = px[cur], cy = py[cur], cz = pz[cur];
        if (lane == 0) {
            sel[c] = (uint8_t)cur;
            qx[c] = cx;
            qy[c] = cy;
            qz[c] = cz;
        }
        float d[PPL];
        unsigned long long m[PPL];
        if constexpr (T2P_FPS_PK && PPL % 2 == 0) {
            // the same eight IEEE operations per point as dist2(), two points per instruction (v_pk_add_f32 / v_pk_mul_f32 are
            // plain single-precision adds / multiplies on a register pair, un-contracted like the scalar form: same bits as
            // oracle/primitives.c:35-40) - the scan is bound by VALU issue, and this is 16 of its ~85 instructions per step
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const f32x2 c2x = {cx, cx}, c2y = {cy, cy}, c2z = {cz, cz};
#pragma unroll
            for (int j = 0; j < PPL; j += 2) {
#pragma clang fp contract(off)
                const f32x2 dx = f32x2{x[j], x[j + 1]} - c2x, dy = f32x2{y[j], y[j + 1]} - c2y, dz = f32x2{z[j], z[j + 1]} - c2z;
                const f32x2 xx = dx * dx, yy = dy * dy, zz = dz * dz;
                const f32x2 sxy = xx + yy;
                const f32x2 dd = sxy + zz;
                d[j] = dd[0];
                d[j + 1] = dd[1];
            }
#pragma unroll
            for (int j = 0; j < PPL; j++) {
                mind[j] = min(mind[j], __float_as_uint(d[j]));
                m[j] = __ballot(d[j] < r2);
            }
        } else {
#pragma unroll
            for (int j = 0; j < PPL; j++) {
                d[j] = dist2(x[j], y[j], z[j], cx, cy, cz);
                mind[j] = min(mind[j], __float_as_uint(d[j]));   // one v_min_u32 (a float compare + select takes two and an s_nop)
                m[j] = __ballot(d[j] < r2);
            }
        }
        int lower = 0, count = 0;
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            lower = __builtin_amdgcn_mbcnt_hi((uint32_t)(m[j] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m[j], lower));
            count += __popcll(m[j]);
        }
        const uint32_t tag = (uint32_t)c << 8;
        // (unsigned 32-bit row positions from the object's uniform base: the stores then take their address as SGPR base +
        // 32-bit lane offset instead of a 64-bit address built on the VALU - 8 of this loop's ~85 VALU instructions)
        char* const rows_b = (char*)rows_out;   // byte offsets: SGPR base + 32-bit lane offset, no shift per store
        if (count <= kMaxNbr) {  // (uniform) nothing to cut off
            uint32_t off = 2u * (uint32_t)(base + lower);
#pragma unroll
            for (int j = 0; j < PPL; j++) {
                const bool hit = d[j] < r2;
                if (hit) *(uint16_t*)(rows_b + off) = (uint16_t)(tag | (uint32_t)(i0 + j));
                off += hit ? 2u : 0u;
            }
        } else {
            uint32_t pos = (uint32_t)lower;
#pragma unroll
            for (int j = 0; j < PPL; j++) {
                const bool hit = d[j] < r2;
                if (hit && pos < (uint32_t)kMaxNbr) *(uint16_t*)(rows_b + 2u * ((uint32_t)base + pos)) = (uint16_t)(tag | (uint32_t)(i0 + j));
                pos += hit ? 1u : 0u;
            }
        }
        const int kept = count < kMaxNbr ? count : kMaxNbr;
        if (self_loops && lane == 0) rows_out[(uint32_t)(base + kept)] = (uint16_t)(((c | 0x80) << 8) | c);
        base += kept + (self_loops ? 1 : 0);
        if (c + 1 < n_c) {  // uniform
            // (squared distances: non-negative, so the arg-max runs on bit patterns - see wave_max_u)
            uint32_t bd = mind[0];
#pragma unroll
            for (int j = 1; j < PPL; j++) bd = max(bd, mind[j]);
            const uint32_t mx = wave_max_u(bd);
            const unsigned long long tie = __ballot(bd == mx);
            int jb = PPL - 1;
#pragma unroll
            for (int j = PPL - 2; j >= 0; j--) jb = mind[j] == mx ? j : jb;
            cur = __builtin_amdgcn_readlane(i0 + jb, (int)__builtin_ctzll(tie));
        }
    }
    if (lane < 4 && base + lane < n_c * (kMaxNbr + 1)) rows_out[base + lane] = 0xFFFF;
    *n_rows_out = base;
    __syncthreads();
}

// Centroid table of one level (out; may be nullptr: the f16x3 SA kernels of levels 1 and 2 build theirs in LDS) + the
// tail behind the features of the SA output rows (was k_pos_table): the [xyz 0] quad alone (tail_quads = 1: levels 1 and 2,
// whose readers mask the 28 pad columns, WsParams::k_live) or [xyz | 0 x 29] (tail_quads = 8: level 3; GA layer 1 sits at
// its register limit and reads whole rows).  H/4 lanes per centroid row, the lane's 4 output columns of W1p in registers,
// 16-byte stores.  Same arithmetic order as the stand-alone kernel.
__device__ __attribute__((noinline)) void emit_centroid_table(const float* qx, const float* qy, const float* qz, int n_c,
                                                    const float* __restrict__ wp, int H, float* __restrict__ out,
                                                    float* __restrict__ tail, int ld_tail, int tail_col0, int tail_quads) {
    const int lane = threadIdx.x;
    if (out == nullptr) {   // tails only: tail_quads (1 or 8) lanes per row
        if (tail == nullptr) return;
        const int hq = lane % tail_quads;
        for (int c = lane / tail_quads; c < n_c; c += 64 / tail_quads)
            *(f32x4*)(tail + (size_t)c * ld_tail + tail_col0 + hq * 4) =
                hq == 0 ? f32x4{qx[c], qy[c], qz[c], 0.f} : f32x4{0.f, 0.f, 0.f, 0.f};
        return;
    }
    const int tpr = H >> 2, rpp = 64 / tpr, hq = lane % tpr;
    const f32x4 w0 = *(const f32x4*)(wp + hq * 4), w1 = *(const f32x4*)(wp + H + hq * 4), w2 = *(const f32x4*)(wp + 2 * H + hq * 4);
    for (int c = lane / tpr; c < n_c; c += rpp) {
        const float px = qx[c], py = qy[c], pz = qz[c];
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            float a = px * w0[e];
            a = fmaf(py, w1[e], a);
            a = fmaf(pz, w2[e], a);
            v[e] = a;
        }
        *(f32x4*)(out + (size_t)c * H + hq * 4) = v;
        if (tail != nullptr && hq < tail_quads)
            *(f32x4*)(tail + (size_t)c * ld_tail + tail_col0 + hq * 4) = hq == 0 ? f32x4{px, py, pz, 0.f} : f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// SA1 layer-1 point table of one object (was k_sa1_point_table): A_1[j] = W1 [rgb_j | xyz_j] + b1
__device__ __attribute__((noinline)) void emit_point_table(const float* px, const float* py, const float* pz,
                                                 const float* __restrict__ rgb, int n_pts, const float* __restrict__ w,
                                                 const float* __restrict__ bias, int H, float* __restrict__ out) {
    const int lane = threadIdx.x;
    const int tpr = H >> 2, rpp = 64 / tpr, hq = lane % tpr;
    f32x4 wk[6];
#pragma unroll
    for (int k = 0; k < 6; k++) wk[k] = *(const f32x4*)(w + k * H + hq * 4);
    const f32x4 bv = *(const f32x4*)(bias + hq * 4);
    for (int j = lane / tpr; j < n_pts; j += rpp) {
        const float in[6] = {rgb[j * 3], rgb[j * 3 + 1], rgb[j * 3 + 2], px[j], py[j], pz[j]};
        f32x4 v = bv;
#pragma unroll
        for (int k = 0; k < 6; k++)
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = fmaf(in[k], wk[k][e], v[e]);
        *(f32x4*)(out + (size_t)j * H + hq * 4) = v;
    }
}

// 6 waves per SIMD (80 registers, some spills) measured fastest: 4 -> 3.26, 5 -> 3.08, 6 -> 2.85, 7 -> 3.5, 8 -> 3.16 ms / 3k cells
// (FAST form, 12k cells: 5 -> 10.0, 6 -> 9.8, 7 -> 10.2, 8 -> 10.7 ms)
// FAST: n_pts == 256 (every level fills its lanes), row lists of all levels wanted, no neighbour tables: level_fast
#ifndef T2P_SG_WAVES
#define T2P_SG_WAVES 6
#endif
template <bool FAST>
__global__ __launch_bounds__(64, T2P_SG_WAVES) void k_sample_group(const float* __restrict__ xyz, int64_t n_obj, int n_pts,
                                                     float r0, float r1, float r2, GroupTables gt) {
    // dynamic LDS: coordinates of the 4 levels | FPS selection | (only when the neighbour table is wanted: nbr + cnt);
    // 5.9 KB in the production path, so the register count sets the occupancy
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* p0 = (float*)smem;                     // [3][256]
    float* p1 = p0 + 3 * kMaxPts;                 // [3][128]
    float* p2 = p1 + 3 * (kMaxPts / 2);           // [3][64]
    float* p3 = p2 + 3 * (kMaxPts / 4);           // [3][32]
    uint8_t* sel_lds = (uint8_t*)(p3 + 3 * (kMaxPts / 8));                 // [128]
    const bool want_nbr = !FAST && gt.nbr[0] != nullptr;
    uint8_t* nbr_lds = want_nbr ? sel_lds + kMaxPts / 2 : nullptr;  // [128*32]
    uint8_t* cnt_lds = want_nbr ? nbr_lds + (kMaxPts / 2) * kMaxNbr : nullptr;                       // [128]
    const int lane = threadIdx.x;
    for (int64_t o = blockIdx.x; o < n_obj; o += gridDim.x) {
        const float* src = xyz + o * (int64_t)n_pts * 3;
        // fp16-range guard: the layer-1 tables are bounded from the input magnitudes (k_guard_check).  The maximum runs over the BIT
        // PATTERNS of |v| (they order like unsigned integers): a NaN input is the largest pattern of all and reaches the guard
        // word - fmaxf() would drop it, and the float-max aggregation of the f16x3 SA kernels drops NaN operands too, so without
        // this a NaN point would come out as a finite embedding (the reference's ReLU / scatter-max propagate it)
        uint32_t in_bits = 0u;
        for (int i = lane; i < n_pts * 3; i += 64) {
            float v = src[i];
            const uint32_t b = __float_as_uint(v) & 0x7fffffffu;
            in_bits = b > in_bits ? b : in_bits;
            p0[(i % 3) * kMaxPts + i / 3] = v;
        }
        if (gt.guard != nullptr) {
            if (gt.rgb != nullptr) {
                const float* col = gt.rgb + o * (int64_t)n_pts * 3;
                for (int i = lane; i < n_pts * 3; i += 64) {
                    const uint32_t b = __float_as_uint(col[i]) & 0x7fffffffu;
                    in_bits = b > in_bits ? b : in_bits;
                }
            }
            guard_publish_bits_above(gt.guard + G_INPUT, in_bits, 0x3f800000u);   // normalised inputs stay within [-1, 1]: nothing is published
        }
        __syncthreads();
        if (gt.A1 != nullptr) {
            // the weight pointers are laundered per object: hipcc would otherwise hoist the (loop-invariant) weight loads
            // of all four tables out of the object loop and keep ~100 registers alive across the FPS loops
            const float *w1 = gt.w1, *b1 = gt.b1;
            asm volatile("" : "+s"(w1), "+s"(b1));
            emit_point_table(p0, p0 + kMaxPts, p0 + 2 * kMaxPts, gt.rgb + o * (int64_t)n_pts * 3, n_pts, w1, b1, gt.H1,
                             gt.A1 + o * (int64_t)n_pts * gt.H1);
        }
        float* pin[4][3] = {{p0, p0 + kMaxPts, p0 + 2 * kMaxPts},
                            {p1, p1 + kMaxPts / 2, p1 + kMaxPts},
                            {p2, p2 + kMaxPts / 4, p2 + kMaxPts / 2},
                            {p3, p3 + kMaxPts / 8, p3 + kMaxPts / 4}};
        const float rr[3] = {r0 * r0, r1 * r1, r2 * r2};
#pragma unroll
        for (int l = 0; l < 3; l++) {
            const int n_d = gt.n_dense[l], n_c = gt.n_cent[l];
            if (want_nbr) {  // unused neighbour slots are zero-filled so that the table is deterministic
                for (int i = lane; i < n_c * kMaxNbr / 4; i += 64) ((uint32_t*)nbr_lds)[i] = 0u;
            }
            __syncthreads();
            int n_rows = 0;
            // the compact row list goes straight to HBM (2-byte stores from the hit lanes): staging it in LDS cost 8.4 KB per
            // wave and capped the occupancy at 11 waves per CU, and this kernel is latency-bound (half the waves: +64 % time)
            uint16_t* g_rows16 = gt.rows[l] ? gt.rows[l] + o * (int64_t)(n_c * (kMaxNbr + 1)) : nullptr;
            if constexpr (FAST) {
                if (l == 0)
                    level_fast<4>(pin[l][0], pin[l][1], pin[l][2], n_c, rr[l], sel_lds, pin[l + 1][0], pin[l + 1][1],
                                  pin[l + 1][2], g_rows16, gt.self_loops, &n_rows);
                else if (l == 1)
                    level_fast<2>(pin[l][0], pin[l][1], pin[l][2], n_c, rr[l], sel_lds, pin[l + 1][0], pin[l + 1][1],
                                  pin[l + 1][2], g_rows16, gt.self_loops, &n_rows);
                else
                    level_fast<1>(pin[l][0], pin[l][1], pin[l][2], n_c, rr[l], sel_lds, pin[l + 1][0], pin[l + 1][1],
                                  pin[l + 1][2], g_rows16, gt.self_loops, &n_rows);
            } else if (n_d > 128)
                level<4>(pin[l][0], pin[l][1], pin[l][2], n_d, n_c, rr[l], sel_lds, pin[l + 1][0], pin[l + 1][1],
                         pin[l + 1][2], nbr_lds, cnt_lds, g_rows16, gt.self_loops, &n_rows);
            else if (n_d > 64)
                level<2>(pin[l][0], pin[l][1], pin[l][2], n_d, n_c, rr[l], sel_lds, pin[l + 1][0], pin[l + 1][1],
                         pin[l + 1][2], nbr_lds, cnt_lds, g_rows16, gt.self_loops, &n_rows);
            else
                level<1>(pin[l][0], pin[l][1], pin[l][2], n_d, n_c, rr[l], sel_lds, pin[l + 1][0], pin[l + 1][1],
                         pin[l + 1][2], nbr_lds, cnt_lds, g_rows16, gt.self_loops, &n_rows);
            if (g_rows16 != nullptr && lane == 0) gt.n_rows[l][o] = (uint16_t)n_rows;
            if (gt.B[l] != nullptr || gt.tail[l] != nullptr) {
                const float* wpl = gt.wp[l];
                asm volatile("" : "+s"(wpl));
                emit_centroid_table(pin[l + 1][0], pin[l + 1][1], pin[l + 1][2], n_c, wpl, gt.H[l],
                                    gt.B[l] ? gt.B[l] + o * (int64_t)n_c * gt.H[l] : nullptr,
                                    gt.tail[l] ? gt.tail[l] + o * (int64_t)n_c * gt.ld_tail[l] : nullptr, gt.ld_tail[l],
                                    gt.tail_col0[l], l == 2 ? 8 : 1);
            }
            uint8_t* g_sel = gt.fps_idx[l] + o * (int64_t)n_c;
            for (int i = lane; i < n_c; i += 64) g_sel[i] = sel_lds[i];
            if (want_nbr) {
                uint8_t* g_nbr = gt.nbr[l] + o * (int64_t)n_c * kMaxNbr;
                uint8_t* g_cnt = gt.cnt[l] + o * (int64_t)n_c;
                for (int i = lane; i < n_c * kMaxNbr / 4; i += 64) ((uint32_t*)g_nbr)[i] = ((const uint32_t*)nbr_lds)[i];
                for (int i = lane; i < n_c; i += 64) g_cnt[i] = cnt_lds[i];
            }
            __syncthreads();
        }
    }
}

// ---- level-0 row lists without the edges of repeated points ----------------------------------------------------------
// T.FixedPoints draws an object's 256 points WITH replacement (dataloading/kitti360pose/utils.py:99-109), so 35-40 % of a
// typical object's points repeat an earlier point bit for bit (coordinates and colour).  A repeat's message
// W2 relu(W1 [x_j | pos_j - pos_i]) equals its original's for every centroid, and the original - lower index, same
// position - lies in the same ball and inside the 32-neighbour cap whenever the repeat does.  Under max-aggregation the
// repeat's edge row can therefore be dropped from the compact list without changing a bit of the result; the cap itself
// was applied to all hits in index order by k_sample_group (the reference's neighbourhood).  -36 % SA1 rows on the
// synthetic cells.  One wavefront per object: a 1024-slot LDS table keeps the lowest index per coordinate hash; a point whose
// slot holds a lower index with identical coordinates and colour is a repeat (a slot taken by a different point of the
// same hash only leaves a repeat undetected: its rows stay, which is always valid).  The list is compacted in place
// (writes never pass the read position), re-terminated and its length updated.
__global__ __launch_bounds__(64) void k_dedup_rows(const float* __restrict__ xyz, const float* __restrict__ rgb, int64_t n_obj,
                                                   uint16_t* rows, uint16_t* n_rows, int max_rows) {
    __shared__ uint32_t slot[1024];
    __shared__ __attribute__((aligned(16))) float pxyz[kMaxPts * 3];
    __shared__ __attribute__((aligned(16))) float prgb[kMaxPts * 3];
    __shared__ uint8_t rep[kMaxPts];
    const int lane = threadIdx.x;
    for (int64_t o = blockIdx.x; o < n_obj; o += gridDim.x) {
        const f32x4* sx = (const f32x4*)(xyz + o * kMaxPts * 3);
        const f32x4* sc = (const f32x4*)(rgb + o * kMaxPts * 3);
        for (int i = lane; i < kMaxPts * 3 / 4; i += 64) {   // straight 16-byte copies: point j at [3 j .. 3 j + 2]
            ((f32x4*)pxyz)[i] = sx[i];
            ((f32x4*)prgb)[i] = sc[i];
        }
        for (int i = lane; i < 1024; i += 64) slot[i] = 0xFFFFFFFFu;
        __syncthreads();
        uint32_t h[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int i = lane * 4 + j;
            uint32_t k = __float_as_uint(pxyz[i * 3]) * 0x9E3779B1u;
            k = (k ^ (k >> 15)) + __float_as_uint(pxyz[i * 3 + 1]) * 0x85EBCA77u;
            k = (k ^ (k >> 13)) + __float_as_uint(pxyz[i * 3 + 2]) * 0xC2B2AE3Du;
            h[j] = (k ^ (k >> 16)) & 1023u;
            atomicMin(&slot[h[j]], (uint32_t)i);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int i = lane * 4 + j;
            const int f = (int)slot[h[j]];   // f <= i: the slot keeps the lowest index of its hash
            bool r = f != i;
#pragma unroll
            for (int e = 0; e < 3; e++)
                r = r && __float_as_uint(pxyz[f * 3 + e]) == __float_as_uint(pxyz[i * 3 + e]) &&
                    __float_as_uint(prgb[f * 3 + e]) == __float_as_uint(prgb[i * 3 + e]);
            rep[i] = r ? 1 : 0;
        }
        __syncthreads();
        // 512 rows per step: lane l owns the 8 consecutive rows [512 s + 8 l, +8) (one 16-byte load; the lists are 16-byte
        // aligned and 0xFFFF-terminated inside their allocation), the next step's load is in flight while this one is written
        uint16_t* list = rows + o * (int64_t)max_rows;
        const int n = n_rows[o];
        const uint4* list4 = (const uint4*)list;
        int out = 0;
        // (a lane's 8 rows are fetched only when they lie inside the object's max_rows slice: the last step of a long list
        // would otherwise read up to 766 bytes past it - past the caller's tensor for the last object)
        uint4 cur = (n > 0 && lane * 8 + 8 <= max_rows) ? list4[lane] : uint4{0, 0, 0, 0};
        for (int b = 0; b < n; b += 512) {
            uint4 nxt = uint4{0, 0, 0, 0};
            if (b + 512 < n && b + 512 + lane * 8 + 8 <= max_rows) nxt = list4[(b + 512) / 8 + lane];
            const uint32_t w[4] = {cur.x, cur.y, cur.z, cur.w};
            uint32_t v[8];
            bool keep[8];
            int below = 0, mine = 0, total = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                v[k] = (k & 1) ? (w[k >> 1] >> 16) : (w[k >> 1] & 0xFFFFu);
                const int r = b + lane * 8 + k;
                // a self-loop row (flag 0x80 in the centroid byte) names a centroid, not a point: always kept
                keep[k] = r < n && ((v[k] & 0x8000u) || !rep[v[k] & 0xFFu]);
                const unsigned long long m = __ballot(keep[k]);
                below = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, below));
                total += __popcll(m);
            }
            int pos = out + below;   // kept rows of lower lanes come first (row order = lane-major)
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (keep[k]) list[pos + mine] = (uint16_t)v[k];
                mine += keep[k] ? 1 : 0;
            }
            out += total;
            cur = nxt;
        }
        if (lane < 4 && out + lane < max_rows) list[out + lane] = 0xFFFF;   // the 4-row terminator consumers rely on
        if (lane == 0) n_rows[o] = (uint16_t)out;
        __syncthreads();
    }
}

}  // namespace

int launch_dedup_rows(const float* xyz, const float* rgb, int64_t n_obj, int n_pts, uint16_t* rows, uint16_t* n_rows,
                      int n_cent, hipStream_t st) {
    T2P_CHECK_ARG(n_pts == kMaxPts && xyz && rgb && rows && n_rows, "dedup_rows: built for %d points per object", kMaxPts);
    T2P_CHECK_ARG(n_cent > 0 && (n_cent * (kMaxNbr + 1)) % 8 == 0 && ((uintptr_t)rows & 15) == 0,
                  "dedup_rows: the row lists are read 16 bytes at a time: n_cent * 33 must be a multiple of 8 (n_cent = %d)", n_cent);
    if (n_obj == 0) return 0;
    const int64_t grid = n_obj < (int64_t)num_cus() * 32 ? n_obj : (int64_t)num_cus() * 32;
    ProfScope ps_("dedup_rows", st);
    T2P_REPEAT(ps_) hipLaunchKernelGGL(k_dedup_rows, dim3((unsigned)grid), dim3(64), 0, st, xyz, rgb, n_obj, rows, n_rows,
                       n_cent * (kMaxNbr + 1));
    T2P_CHECK_LAUNCH("dedup_rows");
    return 0;
}

int launch_sample_group(const float* xyz, int64_t n_obj, int n_pts, const float radius[3], GroupTables gt,
                        hipStream_t st) {
    T2P_CHECK_ARG(n_pts >= 8 && n_pts <= kMaxPts, "sample_group: n_pts=%d outside [8,%d]", n_pts, kMaxPts);
    if (n_obj == 0) return 0;
    int64_t grid = n_obj < (int64_t)num_cus() * 64 ? n_obj : (int64_t)num_cus() * 64;
    ProfScope ps_("sample_group", st);
    const bool want_nbr = gt.nbr[0] != nullptr;
    for (int l = 0; l < 3; l++)
        if (gt.B[l] != nullptr)
            T2P_CHECK_ARG(gt.tail[l] == nullptr || gt.H[l] >= 32, "sample_group: tails need H >= 32");
    for (int l = 0; l < 3; l++)
        if (gt.B[l] != nullptr)
            T2P_CHECK_ARG(gt.H[l] % 4 == 0 && gt.H[l] >= 32 && gt.H[l] <= 256 && 64 % (gt.H[l] / 4) == 0 && gt.wp[l],
                          "sample_group: centroid table of level %d: H=%d", l, gt.H[l]);
    if (gt.A1 != nullptr)
        T2P_CHECK_ARG(gt.H1 % 4 == 0 && gt.H1 >= 4 && gt.H1 <= 256 && 64 % (gt.H1 / 4) == 0 && gt.w1 && gt.b1 && gt.rgb,
                      "sample_group: point table: H=%d", gt.H1);
    T2P_CHECK_ARG(!want_nbr || (gt.nbr[1] && gt.nbr[2] && gt.cnt[0] && gt.cnt[1] && gt.cnt[2]),
                  "sample_group: neighbour tables must be given for all levels or none");
    size_t lds = sizeof(float) * 3 * (kMaxPts + kMaxPts / 2 + kMaxPts / 4 + kMaxPts / 8) + kMaxPts / 2;
    if (want_nbr) lds += (kMaxPts / 2) * kMaxNbr + kMaxPts / 2;
    const bool fast = n_pts == kMaxPts && !want_nbr && gt.rows[0] && gt.rows[1] && gt.rows[2] &&
                      gt.n_dense[0] == kMaxPts && gt.n_dense[1] == kMaxPts / 2 && gt.n_dense[2] == kMaxPts / 4;
    if (fast)
        T2P_REPEAT(ps_) hipLaunchKernelGGL(k_sample_group<true>, dim3((unsigned)grid), dim3(64), lds, st, xyz, n_obj, n_pts, radius[0],
                           radius[1], radius[2], gt);
    else
        hipLaunchKernelGGL(k_sample_group<false>, dim3((unsigned)grid), dim3(64), lds, st, xyz, n_obj, n_pts, radius[0],
                           radius[1], radius[2], gt);
    T2P_CHECK_LAUNCH("sample_group");
    return 0;
}

}  // namespace t2p

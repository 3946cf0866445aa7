// Internal helpers shared by the gfx950 kernels of libt2p_hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

namespace t2p {

void set_error(const char* fmt, ...);

// All launch wrappers return 0 on success, a hipError_t (>0) on a HIP failure or a negative
// T2P_E_* code on an argument error; the message is kept for t2p_last_error().
#define T2P_E_ARG (-1)
#define T2P_E_WORKSPACE (-2)
#define T2P_E_UNSUPPORTED (-3)

#define T2P_CHECK_ARG(cond, ...)                  \
    do {                                          \
        if (!(cond)) {                            \
            ::t2p::set_error(__VA_ARGS__);        \
            return T2P_E_ARG;                     \
        }                                         \
    } while (0)

#define T2P_CHECK_LAUNCH(name)                                                   \
    do {                                                                         \
        hipError_t e__ = hipGetLastError();                                      \
        if (e__ != hipSuccess) {                                                 \
            ::t2p::set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return (int)e__;                                                     \
        }                                                                        \
    } while (0)

#define T2P_TRY(expr)             \
    do {                          \
        int rc__ = (expr);        \
        if (rc__ != 0) return rc__; \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// fp32 pair -> packed fp16 pair, round to NEAREST even (gfx950's v_cvt_pk_f16_f32; the f16x3 split x = hi + lo then carries
// |error| <= 2^-22 |x| - v_cvt_pkrtz_f16_f32, round toward zero, leaves 2^-20 and a bias that adds up along k).
// T2P_SPLIT_RTZ=1 rebuilds round 3's split for A/B measurements.
#ifndef T2P_SPLIT_RTZ
#define T2P_SPLIT_RTZ 0
#endif
typedef __fp16 t2p_fp16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ t2p_fp16x2 cvt_pk_f16(float a, float b) {
#if T2P_SPLIT_RTZ
    return __builtin_amdgcn_cvt_pkrtz(a, b);
#else
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef _Float16 h16x2_ __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(t2p_fp16x2, __builtin_convertvector((f32x2_){a, b}, h16x2_));
#endif
}
// Low piece of the f16x3 split of a pair: fp16(v0 - hi[0]) | fp16(v1 - hi[1]) << 16, the two differences formed exactly in fp32
// by v_fma_mix_f32 (hi half extended, times -1, plus v).  ONE asm statement: behind a stand-alone inline-asm v_fma_mix hipcc puts an
// s_nop in front of the dependent v_cvt_pk (it cannot see that the asm writes whole dwords) - one wait state per pair, in the
// staging arithmetic that rides beside the MFMA stream of the SA kernels.
// CAUTION (round 6, found the hard way): gfx950 needs two wait states between a VALU write of a VGPR and an MFMA that reads it as its
// A / B operand.  hipcc guarantees them between instructions it KNOWS; an asm statement is one opaque non-VALU instruction to its hazard
// recogniser, so nothing separates this statement's final v_cvt_pk from an MFMA that consumes `lo` next.  In k_sa_points the merged
// statement sat one instruction in front of the MFMA that read the four lo registers: SA1 outputs ~1e4 off (libt2p_hip_s1a; a
// different register assignment, s1b, happened to put two instructions between and passed).  Use it only where the consumer is an LDS /
// global store (k_sa3) or is provably far away (k_sa_rows: the pieces of step s + 1 are converted during step s and first multiplied in
// step s + 1, behind an explicit s_nop - see mfma_w there); k_sa_points and k_gemm_x3 keep the separate statements, behind which
// hipcc's own v_cvt_pk -> MFMA spacing applies.  (`lo` is an early-clobber output as well: it then never shares a register with an input.)
__device__ __forceinline__ uint32_t split_lo_pk(t2p_fp16x2 hi, float v0, float v1) {
    uint32_t lo;
    float t0, t1;
    asm("v_fma_mix_f32 %1, %3, -1.0, %4 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %2, %3, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_cvt_pk_f16_f32 %0, %1, %2"
        : "=&v"(lo), "=&v"(t0), "=&v"(t1) : "v"(hi), "v"(v0), "v"(v1));
    return lo;
}
typedef double f64x4 __attribute__((ext_vector_type(4)));

int num_cus();  // cached multiProcessorCount of the current device
int matrix_wgs(); // workgroups of the persistent matrix kernels (= num_cus() unless T2P_MATRIX_WGS says otherwise)
// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of a kernel: set once per (device, kernel), under a
// lock; returns 0 or the hipError_t (message kept for t2p_last_error)
int reserve_lds(const void* kernel, size_t bytes, const char* what);

// ---- fp16-range guard of the f16x3 path -------------------------------------------------------------------------
// The split-precision path converts fp32 activations to fp16 (hi = fp16(v) to nearest: a value past 65504 would
// become inf).  Every kernel that performs such a conversion reports the largest magnitude it converted into one
// of these per-chunk device words (float bits, atomicMax on the unsigned pattern); k_guard_check turns them into the
// caller's sticky overflow flag.  The SA edge kernels convert relu(A_j - B_i) and are covered by the bound
// max|A_l| + max|B_l| taken where the tables are produced, so their inner loop carries no check.
enum GuardSlot {
    G_INPUT = 0,               // max(|xyz|, |rgb|) of the call's inputs when above 1 (normalised inputs: never published).
                               // Bounds the geometry-only layer-1 tables: |B_l| <= ||W1p_l||_1 max|xyz|,
                               // |A_1| <= ||W1_1||_1 max(|rgb|, |xyz|) + max|b1|
    G_A2 = 2, G_A3 = 4,        // layer-1 point tables of SA levels 2 and 3, reported by the dense kernels that write them
    G_F1 = 6, G_F2 = 7, G_F3 = 8,  // SA outputs F_l = the rows the dense weight-stationary kernels split on the fly; exact
                                   // maxima, reported by the SA kernels (their xyz tail is bounded by 1).  GA layer 1's
                                   // output (handed on as fp16 planes) is bounded by ||W||_1 max(F_3, 1) + max|b|
    G_GEMM_IN = 9,    // rows split on the fly by the LDS-tiled f16x3 GEMM (f0, f1, cat, emb)
    G_SLOTS = 16
};
struct GuardBounds {   // host-side norms of the folded weights (packing.py), passed by value to k_guard_check
    float wp_l1[3];     // per SA level: max over columns of |W1p[0][c]| + |W1p[1][c]| + |W1p[2][c]|
    float a1_l1;        // SA1 layer 1 over all 6 inputs: max over columns of sum_k |W1[k][c]|
    float a1_bmax;      // max |b1| of SA1
    float ga1_l1;       // GA layer 1: max over columns of sum_k |W[k][c]|
    float ga1_bmax;
};
// Magnitudes below kGuardFloor are never published (no atomic traffic in the normal case: activations of a trained,
// batch-normalised network are O(1..100)); k_guard_check counts an unpublished word as kGuardFloor.
constexpr float kGuardFloor = 16384.f;
// Low side: fp16 pieces hi = fp16(v), lo = fp16(v - hi) carry an ABSOLUTE error of ~2^-25 (fp16's subnormal spacing) once
// |v| < 2^-3; harmless while a layer's largest activations are O(1), but a layer whose LARGEST magnitude is below 2^-7 keeps
// fewer than ~18 bits of its own scale (and below 2^-14 the hi piece itself goes subnormal).  k_guard_check sets bit 7 then.
constexpr float kGuardTiny = 0.0078125f;
#ifdef __HIPCC__
// m >= 0 (a magnitude; NaN compares false everywhere and is caught by the table producers' own inputs); at most one
// atomic per wavefront, and only when some lane saw a magnitude at or above the floor
__device__ __forceinline__ void guard_publish(uint32_t* slot, float m) {
    if (slot == nullptr) return;
    if (!__any(m >= kGuardFloor)) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(slot, __float_as_uint(m));
}
__device__ __forceinline__ void guard_publish_above(uint32_t* slot, float m, float floor_) {
    if (slot == nullptr) return;
    if (!__any(m > floor_)) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(slot, __float_as_uint(m));
}
// The same on the bit pattern of a magnitude (|v| as an unsigned integer; NaN patterns are the largest and DO get published)
__device__ __forceinline__ void guard_publish_bits_above(uint32_t* slot, uint32_t bits, uint32_t floor_bits) {
    if (slot == nullptr) return;
    if (!__any(bits > floor_bits)) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t other = (uint32_t)__shfl_xor((int)bits, o, 64);
        bits = other > bits ? other : bits;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(slot, bits);
}
// Exact running maximum without a floor, for sites that report once per wave (or rarely): the wave's maximum goes to the slot
// only when it exceeds what the slot already holds (a relaxed load first: after the first few waves almost no atomic is left).
// The exact maxima also serve the LOW side of the guard (k_guard_check: a level whose largest magnitude is below kGuardTiny
// would put the hi pieces of its activations into fp16's subnormals and lose the lo pieces altogether).
__device__ __forceinline__ void guard_publish_exact(uint32_t* slot, float m) {
    if (slot == nullptr) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) {
        const uint32_t b = __float_as_uint(m);
        if (b > __atomic_load_n(slot, __ATOMIC_RELAXED)) atomicMax(slot, b);
    }
}
// Exact form for the persistent SA kernels: `acc` is a wave-uniform bit pattern (an SGPR) that lives across the whole
// launch; guard_flush publishes it once per wave at the end (no floor, no contention: ~2 k atomics per launch).
__device__ __forceinline__ void guard_track_bits(uint32_t& acc, int lane_bits /* pattern of a non-negative float */) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int other = __shfl_xor(lane_bits, o, 64);
        lane_bits = other > lane_bits ? other : lane_bits;
    }
    const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane(lane_bits);
    acc = b > acc ? b : acc;
}
#endif
int launch_guard_check(const uint32_t* guard, int32_t* overflow_flag, const GuardBounds& b, hipStream_t st);  // small_kernels.hip

// Opt-in per-kernel timing (t2p_profile_enable): brackets a launch with hipEvents on the launch stream.
struct ProfScope {
    ProfScope(const char* name, hipStream_t st);
    ~ProfScope();
    int slot;
    hipStream_t st;
    void* ev_end;  // hipEvent_t recorded by the destructor
    int reps;      // 1, or t2p_profile_repeat's count for this scope: the launch sites of the large kernels issue their launch
                   // `reps` times back to back (same arguments: every one of them rewrites its outputs from its inputs), which
                   // is how profiles/energy_table.py holds ONE kernel on the chip for seconds under the power sampler
};
#define T2P_REPEAT(ps) for (int rep_ = (ps).reps; rep_ > 0; --rep_)

// ---- sample_group.hip -------------------------------------------------------------------------------------
// Compact per-object group tables produced by the fused FPS + ball-query kernel.
// Level l (0..2): n_dense[l] dense points -> n_cent[l] = ceil(n_dense[l]/2) centroids; indices are LOCAL to the
// object and refer to the level's dense ordering (level 0: input order; level l>0: FPS order of level l-1).
struct GroupTables {
    uint8_t* fps_idx[3];  // [n_obj, n_cent[l]]
    uint8_t* nbr[3];      // [n_obj, n_cent[l], 32]
    uint8_t* cnt[3];      // [n_obj, n_cent[l]]
    // compact edge-row lists consumed by the SA edge kernel (nullptr = not wanted):
    uint16_t* rows[3];    // [n_obj, n_cent[l]*33]  (centroid | self-loop flag 0x80) << 8 | source index, sorted by centroid
    uint16_t* n_rows[3];  // [n_obj] rows in the list (ball-query hits + one self loop per centroid when self_loops)
    int self_loops;
    int n_dense[3];
    int n_cent[3];
    // Optional layer-1 tables written by the same kernel (it is VALU-bound and leaves the memory pipes idle; as separate
    // kernels these tables were pure HBM-write time).  B[l] = nullptr / A1 = nullptr: not wanted.
    float* B[3];            // centroid tables  B_l[o*n_cent + c][H_l] = W1p_l pos_c
    const float* wp[3];     // [3][H_l] position rows of the level's layer-1 weights
    int H[3];
    float* tail[3];         // SA output rows F_l: the [xyz | 0] quad at column tail_col0[l]; the 28 pad columns behind it are never written, their readers mask them (WsParams::k_live) (row stride ld_tail[l])
    int ld_tail[3], tail_col0[3];
    float* A1;              // SA1 point table A_1[o*n_pts + j][H1] = W1 [rgb_j | xyz_j] + b1
    const float* w1;        // [6][H1]
    const float* b1;
    const float* rgb;       // [n_obj][n_pts][3]
    int H1;
    uint32_t* guard;        // nullptr, or the chunk's GuardSlot words (the input magnitude is reported here)
};
int launch_sample_group(const float* xyz, int64_t n_obj, int n_pts, const float radius[3], GroupTables gt,
                        hipStream_t st);
// Drops from level 0's compact row lists the edges of points that repeat an earlier point of their object bit for bit
// (identical message, so the max-aggregate is unchanged); updates n_rows.  rows [n_obj][n_cent * 33], n_pts = 256.
int launch_dedup_rows(const float* xyz, const float* rgb, int64_t n_obj, int n_pts, uint16_t* rows, uint16_t* n_rows,
                      int n_cent, hipStream_t st);

// ---- tables.hip / small kernels ------------------------------------------------------------------------------
// gather level-l centroid positions: out[(o*n_cent + c), 0..2]
int launch_rownorm(const float* in, int ld_in, int64_t n_rows, int dim, float* out, int ld_out, int col0,
                   hipStream_t st);
int launch_gather_rownorm(const float* table, const int32_t* idx, int64_t n_rows, int dim, float* out, int ld_out,
                          int col0, hipStream_t st);
int launch_segmax(const float* in, int dim, const int32_t* seg_ptr, int n_seg, float* out, int mean,
                  hipStream_t st);
int launch_knn(const float* x, int dim, const int32_t* seg_ptr, int n_seg, int max_seg_rows, int k,
               int32_t* out_idx, hipStream_t st);
// 3 -> 64 -> D MLP (BN folded, ReLU after both layers) + F.normalize, written into out[row*ld_out + col0 ...]
int launch_mlp3_norm(const float* in3, int64_t n_rows, const float* w1, const float* b1, const float* w2,
                     const float* b2, int D, float* out, int ld_out, int col0, hipStream_t st);
// chunk-local cell index arrays: seg_ptr_local[c] = cell_ptr[c] - o_lo (c = 0..n_cells), first[o] = start of o's cell
int launch_pack_objects(const float* raw_xyz, const float* raw_rgb, const int32_t* obj_ptr, const int32_t* sample_idx,
                        const float* rot, int64_t n_obj, int n_pts, float* xyz, float* rgb, float* center, float* mean_rgb,
                        hipStream_t st);
int launch_pack_scene(const float* raw_xyz, const float* raw_rgb, const int32_t* obj_ptr, const int32_t* obj_id, const uint64_t* key,
                      const float* scene_center, const float* scene_color, int64_t n_out, int n_pts, float* xyz, float* rgb,
                      float* center, float* mean_rgb, int32_t* idx_out, hipStream_t st);
int launch_pairwise_ranking(const float* scores, int batch, float margin, float* row_loss, float* d_scores, float* row_cnt,
                            hipStream_t st);
int launch_hardest_ranking(const float* scores, int batch, float margin, float* best, int32_t* where, float* d_scores,
                           hipStream_t st);
int launch_cell_index(const int32_t* cell_ptr, int n_cells, int32_t o_lo, int32_t* seg_ptr_local, int32_t* first,
                      hipStream_t st, uint32_t* guard_to_clear = nullptr);

// ---- tg_gemm.hip: generic tiled fp32-MFMA GEMM ------------------------------------------------------------
// C[M, N] (ldc, column offset c0) = act(A[M, K] (lda) * W[K, N] (row-major, ldw = N) + bias[N])
int launch_gemm(const float* A, int lda, const float* W, const float* bias, float* C, int ldc, int c0, int64_t M,
                int K, int N, int relu, hipStream_t st, const float* resid = nullptr, int ldr = 0);
// few rows x long K, no epilogue (tg_gemm.hip): the per-step products of the training-mode LSTM; K a multiple of 4, any N
int launch_gemm_skinny(const float* A, int lda, const float* W, float* C, int ldc, int64_t M, int K, int N, hipStream_t st);
// tg_gemm_x3.hip: the same contract on the f16x3 matrix path (W as the image of packing.py::pack_gemm_x3)
int launch_gemm_x3(const float* A, int lda, const void* Wx, float scale, const float* bias, float* C, int ldc, int c0,
                   int64_t M, int K, int N, int relu, hipStream_t st, const float* resid = nullptr, int ldr = 0,
                   uint32_t* amax_in = nullptr /* f16x3 guard word for the rows of A */);

// tg_gemm_tn.hip: C[K1, N] = A[M, K1]^T B[M, N], rows split over the grid + fixed-order reduction (training-mode gradients)
size_t gemm_tn_workspace_bytes(int64_t M, int K1, int N);
// train_ops.hip: PointConv edge lists (source row, target row) of a level from k_sample_group's compact row lists
int launch_edge_counts(const uint16_t* rows, const uint16_t* n_rows, const int32_t* first_obj, int64_t n_obj, int n_dense, int n_cent,
                       int self_loops, int32_t* counts, hipStream_t st);
int launch_edge_expand(const uint16_t* rows, const uint16_t* n_rows, const int32_t* first_obj, const int32_t* cent_ptr, int64_t n_obj,
                       int n_dense, int n_cent, int self_loops, int32_t* src, int32_t* dst, hipStream_t st);
// train_gemm.hip: weight + bias gradient of the training-mode Linear layers
size_t linear_wgrad_workspace_bytes(int64_t M, int K1, int N);
int launch_linear_wgrad_f32(const float* dY, int lda, const float* X, int ldb, float* dW, int ldc, float* colsum, int64_t M, int K1, int N,
                            void* ws, size_t ws_bytes, hipStream_t st);
int launch_gemm_tn(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int64_t M, int K1, int N, void* ws,
                   size_t ws_bytes, hipStream_t st);

// ---- ws_gemm.hip: weight-stationary streaming fp32-MFMA kernels ---------------------------------------------
enum WsMode { WS_DENSE_STORE = 0, WS_DENSE_GROUPMAX = 1, WS_EDGE_KNN = 3 };
struct WsParams {
    // operand tables
    const float* A;    // source rows [n_src, lda]   (dense: the GEMM's A; edge: layer-1 point table)
    int lda;
    int k_live;        // dense fp32 rows: columns >= k_live (a multiple of 4; 0 = all K) are taken as zero and never read
    const float* Bc;   // edge modes: per-destination term [n_dst, H]
    const float* W;    // [K][ldw] k-major weights (BN folded)
    int ldw;
    const void* W_x3;  // nullptr, or the packed f16x3 register image of the whole [K][ldw] matrix (dense modes)
    // f16x3 only: activations may travel between two dense kernels already split into fp16 hi / lo planes
    const void* A_hi;  // non-null: A is given as two [M][lda] fp16 planes (no conversion while staging)
    const void* A_lo;
    void* out_hi;      // non-null (DENSE_STORE): write the result as two [M][ldo] fp16 planes instead of fp32
    void* out_lo;
    const float* bias; // [N]
    float* out;        // dense_store: [M, ldo]; groupmax: [n_groups, ldo]; edge: [n_dst, ldo]
    int ldo;
    int relu;
    int64_t n_groups;  // dense: ceil(M / rows_per_group); edge SA: objects; edge kNN: ceil(n_dst/32)
    int64_t M;         // dense: total rows
    // edge kNN
    uint32_t* amax_out;     // f16x3 guard (nullable): largest output magnitude (a table the SA kernels split, or fp16 planes)
    const int32_t* knn_idx; // [n_dst, knn_k] (-1 = absent)
    int knn_k;
    int64_t n_dst;
    int mean;               // 0 = max aggregation, 1 = mean (needs knn_k == 8)
    int knn_group;          // destinations per group: 32 (0 = 32), or 8 for calls too small to fill the CUs with 32-destination groups
};
int launch_ws(int mode, int K, int N, const WsParams& p, hipStream_t st);

// ---- ws_sa.hip: the set-abstraction edge kernel (flattened, fully pipelined batch stream) ----------------------------
struct SaParams {
    const float* A;   // layer-1 point table [n_obj*n_dense][H]
    const float* Bc;  // centroid table [n_obj*n_cent][H] (unused when wp is given)
    const float* wp;  // nullptr, or the [3][H] position rows of the layer-1 weights: the f16x3 kernel then builds each object's
                      // centroid table B_i = W1p pos_i in LDS itself (pos_i from the [xyz | 0] tail of the `out` rows)
    const float* W;   // [H][C] k-major (fp32 MFMA path)
    const void* W_x3; // nullptr, or the host-packed f16x3 register image of W (selects the split-precision path)
    const float* bias;  // [C]; the f16x3 path takes it pre-multiplied by the weight image's scale
    float out_scale;    // f16x3: 1 / scale of the weight image (results are drained as acc * out_scale); fp32: 1
    float* out;       // [n_obj*n_cent][ldo] rows = [features C | centroid xyz 0 | pad] (this kernel writes the features)
    int ldo;
    const uint16_t* rows;    // GroupTables::rows[l]
    const uint16_t* n_rows;  // GroupTables::n_rows[l]
    const int32_t* first;    // [n_obj] first object of the object's cell (self-loop aliasing)
    const uint8_t* fps_idx;  // [n_obj][n_cent]
    const float* pos_src;    // rows holding the dense positions of this level
    int ld_pos, pos_col0;
    const float* feat_src;   // sa_points.hip (level 1 only): the points' features [n_obj*n_dense][3] (rgb), the whole layer-1
    const float* w1;         // weight matrix [6][H] (feature rows, then position rows) and its bias [H]; nullptr: the kernels
    const float* b1;         // that gather the point table A
    int n_dense, n_cent;
    int64_t n_obj;
    int32_t* prefix_ws;      // [n_obj+1] scratch (tile prefix sums)
    int32_t* bounds_ws;      // [n_workgroups+1] scratch (balanced contiguous object ranges)
    int balanced;            // 1: bounds_ws was filled by launch_sa_balance_levels for this level's launch shape
    uint32_t* amax_out;      // f16x3 guard (nullable): largest output magnitude (the next dense kernel splits these rows)
};
int launch_ws_sa(int H, int C, const SaParams& p, hipStream_t st);
// sa_rows.hip: row-owning f16x3 kernel of SA level 2 (H = C = 128, LDS centroid table): true when launch_ws_sa routes p there
bool sa_rows_selected(int H, int C, const SaParams& p);
int launch_sa_rows(int H, int C, const SaParams& p, hipStream_t st);
int sa_rows_launch_shape(int64_t n_obj, int* tile_rows, int* n_wg);
// sa_points.hip: both layers per edge from the object's points in LDS, f16x3 kernel of SA level 1 (needs wp, w1, b1, feat_src)
bool sa_points_selected(int H, int C, const SaParams& p);
int launch_sa_points(int H, int C, const SaParams& p, hipStream_t st);
int sa_points_launch_shape(int64_t n_obj, int* tile_rows, int* n_wg);
// sa3.hip: SA level 3 (H = C = 256, LDS centroid table), column-slice waves with scalar per-row control
bool sa3_selected(int H, int C, const SaParams& p);
int launch_sa3(const SaParams& p, hipStream_t st);
int sa3_launch_shape(int64_t n_obj, int* tile_rows, int* n_wg);
// One launch that balances all three levels (their row counts are known once k_sample_group has run); fills
// prefix_ws / bounds_ws of every p[l] for the launch shape launch_ws_sa(H[l], C[l], p[l]) will use.
int launch_sa_balance_levels(const SaParams p[3], const int H[3], const int C[3], hipStream_t st);

// ---- lstm.hip ---------------------------------------------------------------------------------------------
// gate_table: [2][V][4D] with V = vocabulary + 1: the LAST row of each direction is all zeros (what rows past their length
// gather); the f16x3 path scales the table in place by whh_scale.
int launch_bilstm_impl(const float* gate_table /*[2][V][4D]*/, const float* whh /*[2][D][4D] k-major*/,
                       const void* whh_x3 /*nullable: [2 dirs] scaled f16x3 images of whh (selects the f16x3 recurrence)*/,
                       float whh_scale, const int32_t* tokens /*[B, T]*/, const int32_t* lengths, int B, int T, int V, int D,
                       float* hdir_ws /*[2][B][D]*/, float* out /*[B, D] mean of the two final hidden states*/,
                       hipStream_t st);

int launch_lstm_cell_fwd(const float* pre, const float* table, const int32_t* tokens, const int32_t* lengths, int64_t B, int T,
                         int D, int step, int reverse, const float* c_prev, const float* h_prev, float* gates, float* c,
                         float* h, hipStream_t st);
int launch_lstm_cell_bwd(const float* dh_gemm, const float* dh_carry_in, const float* dc_in, const float* gates,
                         const float* c_prev, const float* c, const int32_t* lengths, int64_t B, int D, int step, float* d_pre,
                         float* dc_out, float* dh_carry_out, hipStream_t st);

// ---- train_ops.hip --------------------------------------------------------------------------------------------
size_t bn_train_workspace_bytes(int64_t rows, int n_seg, int C);
int launch_bn_relu_train_forward(const float* x, const int32_t* seg_ptr, int n_seg, int64_t rows, int C, const float* gamma,
                                 const float* beta, float eps, int relu, float* y, float* mean, float* invstd,
                                 float* var_unbiased, double* part, hipStream_t st);
int launch_bn_relu_train_backward(const float* dy, const float* x, const float* beta, const int32_t* seg_ptr, int n_seg,
                                  int64_t rows, int C, const float* mean, const float* invstd, const float* gamma, int relu,
                                  float* dx, float* dgamma_seg, float* dbeta_seg, double* part, hipStream_t st);
int launch_edge_feat_fwd(const float* x, const float* pos, const float* pos_c, const int32_t* src, const int32_t* dst, int64_t E,
                         int C, int W, float* out, hipStream_t st);
int launch_edge_feat_bwd(const float* dout, const int32_t* src, int64_t E, int C, int W, float* dx, hipStream_t st);
int launch_pair_feat_fwd(const float* x, const int32_t* tgt, const int32_t* src, int64_t E, int D, float* out, hipStream_t st);
int launch_pair_feat_bwd(const float* dout, const int32_t* tgt, const int32_t* src, int64_t E, int D, float* dx, hipStream_t st);
int launch_segment_mean(const float* x, const int32_t* seg_ptr, int n_seg, int C, float* out, hipStream_t st);
int launch_segment_mean_backward(const float* dout, const int32_t* seg_ptr, int n_seg, int C, float* dx, hipStream_t st);
int launch_rownorm_bwd(const float* x, const float* dy, int64_t n_rows, int dim, float* dx, hipStream_t st);
int launch_segment_max(const float* x, const int32_t* seg_ptr, int n_seg, int C, float* out, int32_t* arg, hipStream_t st);
int launch_segment_max_backward(const float* dout, const int32_t* arg, const int32_t* seg_ptr, int n_seg, int C, float* dx,
                                hipStream_t st);

// ---- sim_topk.hip -----------------------------------------------------------------------------------------
size_t sim_topk_workspace_bytes(int64_t nq, int64_t nc, int k);
int launch_sim_topk(const float* Q, const float* C, int64_t nq, int64_t nc, int dim, int k, int64_t c_index_offset,
                    int64_t* out_idx, double* out_score, void* ws, size_t ws_bytes, hipStream_t st);

}  // namespace t2p

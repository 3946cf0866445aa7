// extern "C" entry points of libt2p_hip.so (see include/t2p.h) and the launch orchestration of the cell branch.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <set>
#include <utility>

#include "../../include/t2p.h"
#include "t2p_common.h"

namespace t2p {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int num_cus() {
    static int cached = 0;
    if (cached == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            cached = prop.multiProcessorCount;
        if (cached <= 0) cached = 256;
    }
    return cached;
}

// Workgroups of the three persistent matrix kernels (k_sa3, k_sa_rows, k_ga2: one workgroup fills a CU).  Development switch
// T2P_MATRIX_WGS=<n>: fewer than one per CU leaves whole CUs to the kernels of the other HIP stream (notebook, round 4).
int matrix_wgs() {
    static int cached = 0;
    if (cached == 0) {
        cached = num_cus();
        const char* e = getenv("T2P_MATRIX_WGS");
        if (e != nullptr && atoi(e) >= 8 && atoi(e) <= cached) cached = atoi(e) / 8 * 8;
    }
    return cached;
}

int reserve_lds(const void* kernel, size_t bytes, const char* what) {
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({dev, kernel})) return 0;
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
        set_error("%s: cannot reserve %zu B of LDS on device %d: %s", what, bytes, dev, hipGetErrorString(e));
        return (int)e;
    }
    done.insert({dev, kernel});
    return 0;
}

// ---- opt-in kernel timing ---------------------------------------------------------------------------------------
// The only process-global state of the library besides the caches above: a ring of event pairs, guarded by one mutex
// (launches from several host threads may be recorded; the lock is taken only while profiling is on).
struct ProfRec {
    const char* name;
    hipEvent_t a, b;
};
static std::mutex g_prof_mu;
static volatile bool g_prof_on = false;
static ProfRec g_prof[8192];
static int g_prof_n = 0;

static char g_repeat_name[64] = {0};
static volatile int g_repeat_n = 1;

ProfScope::ProfScope(const char* name, hipStream_t s) : slot(-1), st(s), reps(1) {
    if (g_repeat_n > 1 && strncmp(name, g_repeat_name, sizeof(g_repeat_name)) == 0) reps = g_repeat_n;
    if (!g_prof_on) return;
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess) return;
    if (hipEventCreate(&b) != hipSuccess) {
        (void)hipEventDestroy(a);
        return;
    }
    {
        std::lock_guard<std::mutex> lock(g_prof_mu);
        if (g_prof_n < 8192) {
            slot = g_prof_n++;
            g_prof[slot].name = name;
            g_prof[slot].a = a;
            g_prof[slot].b = b;
        }
    }
    if (slot < 0) {
        (void)hipEventDestroy(a);
        (void)hipEventDestroy(b);
        return;
    }
    ev_end = b;
    (void)hipEventRecord(a, st);
}
ProfScope::~ProfScope() {
    if (slot >= 0) (void)hipEventRecord((hipEvent_t)ev_end, st);
}

namespace {

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Bump {
    char* base;
    size_t off, cap;
    template <typename T>
    T* take(size_t n) {
        off = align_up(off, 256);
        T* p = (T*)(base ? base + off : nullptr);
        off += n * sizeof(T);
        return p;
    }
};

// Level geometry for n_pts points: dense/centroid counts, feature widths, hidden widths.
struct Geo {
    int nd[3], nc[3];
    static constexpr int H[3] = {32, 128, 256};
    static constexpr int C[3] = {64, 128, 256};
    static constexpr int LD[3] = {96, 160, 288};  // SA output row = [C | xyz 0 | 28 pad columns: zero at level 3, never written at levels 1-2 (their readers mask them: k_live)]: K % 32 == 0 for the next GEMM
    explicit Geo(int n_pts) {
        nd[0] = n_pts;
        for (int l = 0; l < 3; l++) {
            nc[l] = (nd[l] + 1) / 2;
            if (l < 2) nd[l + 1] = nc[l];
        }
    }
};
constexpr int Geo::H[3];
constexpr int Geo::C[3];
constexpr int Geo::LD[3];

struct CellWs {
    GroupTables gt;
    float *A[3], *B[3], *F[3];
    float *gh, *f0, *f1, *f2, *cat, *emb, *embn, *P, *Q, *x1, *pool, *l1, *l2;
    int32_t *knn, *seg_ptr, *first, *prefix[3], *bounds[3];
    uint32_t* guard;  // [G_SLOTS] fp16-range guard words of the chunk (t2p_common.h)
};

// Carve the per-chunk workspace (n objects, nb cells).  With base == nullptr this only measures.
size_t carve(Bump& b, int64_t n, int64_t nb, const t2p_cell_config& cfg, CellWs* ws) {
    Geo g(cfg.n_pts);
    const int D = cfg.embed_dim;
    const int nfeat = (cfg.use_class ? 1 : 0) + (cfg.use_color ? 1 : 0) + (cfg.use_position ? 1 : 0);
    CellWs w{};
    for (int l = 0; l < 3; l++) {
        w.gt.n_dense[l] = g.nd[l];
        w.gt.n_cent[l] = g.nc[l];
        w.gt.fps_idx[l] = b.take<uint8_t>(n * g.nc[l]);
        w.gt.nbr[l] = b.take<uint8_t>(n * g.nc[l] * 32);   // only filled when a trace asks for it
        w.gt.cnt[l] = b.take<uint8_t>(n * g.nc[l]);
        w.gt.rows[l] = b.take<uint16_t>(n * g.nc[l] * 33);
        w.gt.n_rows[l] = b.take<uint16_t>(n);
        w.A[l] = b.take<float>(n * g.nd[l] * Geo::H[l]);
        w.B[l] = b.take<float>(n * g.nc[l] * Geo::H[l]);
        w.F[l] = b.take<float>(n * g.nc[l] * Geo::LD[l]);
    }
    w.gh = b.take<float>(n * g.nc[2] * 512);
    w.f0 = b.take<float>(n * 1024);
    w.f1 = b.take<float>(n * 512);
    w.f2 = b.take<float>(n * 256);
    w.cat = b.take<float>(n * (size_t)(nfeat > 0 ? nfeat : 1) * D);
    w.emb = b.take<float>(n * D);
    w.embn = b.take<float>(n * D);
    w.P = b.take<float>(n * D);
    w.Q = b.take<float>(n * D);
    w.x1 = b.take<float>(n * D);
    w.knn = b.take<int32_t>(n * (size_t)cfg.knn_k);
    w.first = b.take<int32_t>(n);
    for (int l = 0; l < 3; l++) {
        w.prefix[l] = b.take<int32_t>(n + 1);
        w.bounds[l] = b.take<int32_t>(1024 + 1);
    }
    w.seg_ptr = b.take<int32_t>(nb + 1);
    w.guard = b.take<uint32_t>(G_SLOTS);
    w.pool = b.take<float>(nb * D);
    w.l1 = b.take<float>(nb * D);
    w.l2 = b.take<float>(nb * D);
    if (ws) *ws = w;
    return align_up(b.off, 256);
}

// Default chunk: as many objects as the 32-bit table offsets of the SA kernels allow (< 65,536); measured 32,768 / 49,152 /
// 65,000 objects per chunk: 105.1 / 104.4 / 104.1 ms per 12k-cell step (fewer launches, fewer kernel tails), ~0.6 MB each
int default_chunk(const t2p_cell_config& cfg) { return cfg.chunk_objects > 0 ? cfg.chunk_objects : T2P_DEFAULT_CHUNK_OBJECTS; }

int check_cfg(const t2p_cell_config* cfg) {
    T2P_CHECK_ARG(cfg != nullptr, "encode_cells: cfg is NULL");
    if (cfg->n_pts != 256) {
        set_error("encode_cells: n_pts=%d not built (256; the GA max-pool tile assumes 32 points per object)", cfg->n_pts);
        return T2P_E_UNSUPPORTED;
    }
    if (cfg->objects_only) {
        if (cfg->embed_dim < 64 || cfg->embed_dim > 512 || cfg->embed_dim % 64 != 0) {
            set_error("encode_cells: objects_only supports embed_dim in {64, 128, ..., 512}, got %d", cfg->embed_dim);
            return T2P_E_UNSUPPORTED;
        }
    } else if (cfg->embed_dim != 256 && cfg->embed_dim != 128 && cfg->embed_dim != 384) {
        set_error("encode_cells: embed_dim=%d not built for the cell head (128, 256, 384; the host pads e.g. 300 to 384)", cfg->embed_dim);
        return T2P_E_UNSUPPORTED;
    }
    T2P_CHECK_ARG(cfg->variation == 0 || cfg->variation == 1, "encode_cells: variation=%d (0 = max, 1 = mean)",
                  cfg->variation);
    if (cfg->variation == 1 && cfg->knn_k != 8) {
        set_error("encode_cells: variation=1 (mean aggregation) is built for knn_k = 8 only, got %d", cfg->knn_k);
        return T2P_E_UNSUPPORTED;
    }
    T2P_CHECK_ARG(cfg->pointnet_features >= 0 && cfg->pointnet_features <= 2, "encode_cells: pointnet_features=%d",
                  cfg->pointnet_features);
    T2P_CHECK_ARG(cfg->use_class || cfg->use_color || cfg->use_position, "encode_cells: use_features is empty");
    T2P_CHECK_ARG(cfg->knn_k >= 1 && cfg->knn_k <= 32, "encode_cells: knn_k=%d outside [1,32]", cfg->knn_k);
    T2P_CHECK_ARG(cfg->precision == 0 || cfg->precision == 1, "encode_cells: precision=%d (0 = fp32, 1 = f16x3)",
                  cfg->precision);
    T2P_CHECK_ARG((cfg->tuning & ~T2P_TUNING_MASK) == 0, "encode_cells: tuning=%#x has bits outside %#x", cfg->tuning, T2P_TUNING_MASK);
    // 32-bit byte offsets into the per-chunk tables (n_obj * 128 points * 128 floats * 4 B < 2^32) and 16-bit object-local
    // indices cap a chunk at 65,535 objects; checked here, not deep inside a kernel launcher
    T2P_CHECK_ARG(cfg->chunk_objects >= 0 && cfg->chunk_objects <= T2P_MAX_CHUNK_OBJECTS,
                  "encode_cells: chunk_objects=%d outside [0, %d] (0 = default %d); the workspace takes ~0.6 MB per object of a chunk",
                  cfg->chunk_objects, T2P_MAX_CHUNK_OBJECTS, T2P_DEFAULT_CHUNK_OBJECTS);
    T2P_CHECK_ARG(!cfg->class_embed || cfg->class_idx != nullptr, "encode_cells: class_embed needs class_idx");
    T2P_CHECK_ARG(!cfg->color_embed || cfg->color_idx != nullptr, "encode_cells: color_embed needs color_idx");
    return 0;
}

template <typename T>
int copy_trace(T* dst, const T* src, size_t n, hipStream_t st) {
    if (dst == nullptr || n == 0) return 0;
    hipError_t e = hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) {
        set_error("encode_cells: trace copy failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

int encode_chunk(const float* xyz, const float* rgb, const float* center, const float* mean_rgb,
                 const int32_t* cell_ptr_dev /* at first cell of chunk */, int32_t o_lo, int64_t n, int64_t nb,
                 int max_cell, const t2p_cell_weights& W, const t2p_cell_config& cfg, float* out,
                 const t2p_cell_trace* tr, int64_t trace_obj0, CellWs& ws, hipStream_t st) {
    Geo g(cfg.n_pts);
    const int D = cfg.embed_dim;
    // fp16-range guard (f16x3 only): the chunk's words are cleared by the first kernel and judged by the last
    uint32_t* guard = (cfg.precision == 1 && cfg.overflow_flag != nullptr) ? ws.guard : nullptr;
    auto gslot = [&](int i) -> uint32_t* { return guard ? guard + i : nullptr; };
    GuardBounds gbounds{};
    for (int l = 0; l < 3; l++) gbounds.wp_l1[l] = W.sa_wp_l1[l];
    gbounds.a1_l1 = W.sa_a1_l1;
    gbounds.a1_bmax = W.sa_b1_absmax;
    gbounds.ga1_l1 = W.ga_w1_l1;
    gbounds.ga1_bmax = W.ga_b1_absmax;
    // f16x3: the SA kernels build their centroid tables B_i = W1p pos_i in LDS (sa_points.hip, sa_rows.hip, sa3.hip); the HBM
    // tables B_l are then neither written nor read.  The fp32 kernels (ws_sa.hip) gather all three from HBM.
    const bool lds_btab = cfg.precision == 1;
    const bool lds_btab0 = lds_btab;
    // level 0 runs on sa_points.hip, which computes layer 1 per edge from the points themselves: no point table A_1 either
    const bool sa1_points = lds_btab0 && cfg.n_pts == 256;
    T2P_TRY(launch_cell_index(cell_ptr_dev, (int)nb, o_lo, ws.seg_ptr, ws.first, st, guard));
    // models/object_encoder.py:86: the PointNet++ only runs when the "class" feature does not come from class_embedding
    const bool run_pointnet = cfg.use_class && !cfg.class_embed;
    if (run_pointnet) {
    ws.gt.self_loops = cfg.self_loops;
    {
        GroupTables gt = ws.gt;
        const bool want_nbr = tr != nullptr && (tr->nbr[0] || tr->nbr[1] || tr->nbr[2] || tr->cnt[0] || tr->cnt[1] || tr->cnt[2]);
        if (!want_nbr)
            for (int l = 0; l < 3; l++) gt.nbr[l] = gt.cnt[l] = nullptr;
        // the layer-1 tables that depend on geometry only are written by the same kernel: B_l = W1p_l pos_i (+ the
        // [xyz | 0] tail of the F_l rows) for every level, and the K = 6 point table A_1 of level 0
        for (int l = 0; l < 3; l++) {
            const int cf = l == 0 ? 3 : Geo::C[l - 1];
            gt.B[l] = (l > 0 ? lds_btab : lds_btab0) ? nullptr : ws.B[l];
            gt.wp[l] = W.sa_w1[l] + (size_t)cf * Geo::H[l];
            gt.H[l] = Geo::H[l];
            gt.tail[l] = ws.F[l];
            gt.ld_tail[l] = Geo::LD[l];
            gt.tail_col0[l] = Geo::C[l];
        }
        gt.A1 = sa1_points ? nullptr : ws.A[0];
        gt.w1 = W.sa_w1[0];
        gt.b1 = W.sa_b1[0];
        gt.rgb = rgb;
        gt.H1 = Geo::H[0];
        gt.guard = guard;

        T2P_TRY(launch_sample_group(xyz, n, cfg.n_pts, cfg.radius, gt, st));
        // level 0: edges of repeated points leave the row list (same max-aggregate, -36 % SA1 rows on the synthetic cells)
        if (!(cfg.tuning & 1) && cfg.n_pts == 256)
            T2P_TRY(launch_dedup_rows(xyz, rgb, n, cfg.n_pts, gt.rows[0], gt.n_rows[0], g.nc[0], st));
        // the per-centroid row counts of all three levels exist now: cut every level's balanced object ranges at once
        SaParams bp[3] = {};
        for (int l = 0; l < 3; l++) {
            bp[l].n_rows = gt.n_rows[l];
            bp[l].n_obj = n;
            bp[l].prefix_ws = ws.prefix[l];
            bp[l].bounds_ws = ws.bounds[l];
            bp[l].W_x3 = cfg.precision == 1 ? W.sa_w2_x3[l] : nullptr;
            bp[l].wp = (l > 0 ? lds_btab : lds_btab0) ? W.sa_w1[l] : nullptr;   // (non-null = LDS centroid table: selects the launch shape)
        }
        bp[0].w1 = sa1_points ? W.sa_w1[0] : nullptr;
        T2P_TRY(launch_sa_balance_levels(bp, Geo::H, Geo::C, st));
    }

    // ---- three set-abstraction levels -----------------------------------------------------------------------
    for (int l = 0; l < 3; l++) {
        const int H = Geo::H[l], C = Geo::C[l];
        const int cf = l == 0 ? 3 : Geo::C[l - 1];  // feature columns in front of the xyz columns
        const float* pos_src = l == 0 ? xyz : ws.F[l - 1];
        const int ld_pos = l == 0 ? 3 : Geo::LD[l - 1];
        const int pos_col0 = l == 0 ? 0 : Geo::C[l - 1];
        // layer-1 point table A_j = W1 [x_j | pos_j] + b1
        if (l != 0) {  // (level 0: written by k_sample_group)
            WsParams p{};
            p.A = ws.F[l - 1];
            p.lda = Geo::LD[l - 1];
            p.k_live = Geo::C[l - 1] + 4;   // [features | xyz 0]; the pad columns behind are never written
            p.W = W.sa_w1[l];
            p.W_x3 = cfg.precision == 1 ? W.sa_w1_x3[l] : nullptr;
            p.ldw = H;
            p.bias = W.sa_b1[l];
            p.out = ws.A[l];
            p.ldo = H;
            p.relu = 0;
            p.M = n * g.nd[l];
            p.amax_out = gslot(l == 1 ? G_A2 : G_A3);
            T2P_TRY(launch_ws(WS_DENSE_STORE, Geo::LD[l - 1] - (cfg.precision == 1 ? 16 : 0), H, p, st));  // (f16x3: K stops at C + 16)
        }
        // per-edge ReLU(A_j - B_i) -> layer 2 -> max per centroid
        SaParams p{};
        p.A = ws.A[l];
        p.Bc = ws.B[l];
        p.wp = (l > 0 ? lds_btab : lds_btab0) ? W.sa_w1[l] + (size_t)cf * Geo::H[l] : nullptr;
        p.W = W.sa_w2[l];
        p.W_x3 = cfg.precision == 1 ? W.sa_w2_x3[l] : nullptr;
        p.bias = cfg.precision == 1 ? W.sa_b2_x3[l] : W.sa_b2[l];
        p.out_scale = cfg.precision == 1 ? 1.0f / W.sa_w2_scale[l] : 1.0f;
        p.out = ws.F[l];
        p.ldo = Geo::LD[l];
        p.rows = ws.gt.rows[l];
        p.n_rows = ws.gt.n_rows[l];
        p.first = ws.first;
        p.fps_idx = ws.gt.fps_idx[l];
        p.pos_src = pos_src;
        p.ld_pos = ld_pos;
        p.pos_col0 = pos_col0;
        if (l == 0 && sa1_points) {
            p.feat_src = rgb;
            p.w1 = W.sa_w1[0];
            p.b1 = W.sa_b1[0];
        }
        p.n_dense = g.nd[l];
        p.n_cent = g.nc[l];
        p.n_obj = n;
        p.prefix_ws = ws.prefix[l];
        p.bounds_ws = ws.bounds[l];
        p.balanced = 1;
        p.amax_out = gslot(G_F1 + l);
        T2P_TRY(launch_ws_sa(H, C, p, st));
    }
    // ---- global abstraction: [x | pos] -> 512 -> 1024, max over the object's 32 points ------------------------
    {
        WsParams p{};
        p.A = ws.F[2];
        p.lda = Geo::LD[2];
        p.W = W.ga_w1;
        p.W_x3 = cfg.precision == 1 ? W.ga_w1_x3 : nullptr;
        p.ldw = 512;
        p.bias = W.ga_b1;
        p.out = ws.gh;
        // f16x3: GA1 hands its ReLU output to GA2 already split into fp16 hi / lo planes (same bytes as fp32), so the
        // eight column-slice workgroups of GA2 stage it with plain 16-byte copies instead of re-splitting it 8 times
        _Float16* gh_hi = (_Float16*)ws.gh;
        _Float16* gh_lo = gh_hi + (size_t)n * g.nc[2] * 512;
        if (cfg.precision == 1) {
            p.out_hi = gh_hi;
            p.out_lo = gh_lo;
        }
        p.ldo = 512;
        p.relu = 1;
        p.M = n * g.nc[2];
        T2P_TRY(launch_ws(WS_DENSE_STORE, Geo::LD[2] - (cfg.precision == 1 ? 16 : 0), 512, p, st));
        WsParams q{};
        q.A = ws.gh;
        if (cfg.precision == 1) {
            q.A_hi = gh_hi;
            q.A_lo = gh_lo;
        }
        q.lda = 512;
        q.W = W.ga_w2;
        q.W_x3 = cfg.precision == 1 ? W.ga_w2_x3 : nullptr;
        q.ldw = 1024;
        q.bias = W.ga_b2;
        q.out = ws.f0;
        q.ldo = 1024;
        q.relu = 1;
        q.M = n * g.nc[2];
        T2P_TRY(launch_ws(WS_DENSE_GROUPMAX, 512, 1024, q, st));
    }
    // ---- PointNet2 heads + ObjectEncoder ------------------------------------------------------------------------
    if (cfg.precision == 1 && W.lin1_x3 && W.lin2_x3) {
        T2P_TRY(launch_gemm_x3(ws.f0, 1024, W.lin1_x3, W.lin1_scale, W.lin1_b, ws.f1, 512, 0, n, 1024, 512, 1, st, nullptr, 0, gslot(G_GEMM_IN)));
        T2P_TRY(launch_gemm_x3(ws.f1, 512, W.lin2_x3, W.lin2_scale, W.lin2_b, ws.f2, 256, 0, n, 512, 256, 1, st, nullptr, 0, gslot(G_GEMM_IN)));
    } else {
        T2P_TRY(launch_gemm(ws.f0, 1024, W.lin1_w, W.lin1_b, ws.f1, 512, 0, n, 1024, 512, 1, st));
        T2P_TRY(launch_gemm(ws.f1, 512, W.lin2_w, W.lin2_b, ws.f2, 256, 0, n, 512, 256, 1, st));
    }
    }  // run_pointnet
    const int nfeat = (cfg.use_class ? 1 : 0) + (cfg.use_color ? 1 : 0) + (cfg.use_position ? 1 : 0);
    const int ldcat = nfeat * D;
    int slot = 0;
    if (cfg.use_class && cfg.class_embed) {
        T2P_TRY(launch_gather_rownorm(W.class_embedding, cfg.class_idx + o_lo, n, D, ws.cat, ldcat, slot * D, st));
        slot++;
    } else if (cfg.use_class) {
        const float* fin = cfg.pointnet_features == 0 ? ws.f0 : (cfg.pointnet_features == 1 ? ws.f1 : ws.f2);
        const int kin = cfg.pointnet_features == 0 ? 1024 : (cfg.pointnet_features == 1 ? 512 : 256);
        // mlp_pointnet into P (scratch), then F.normalize into the concat slot
        if (cfg.precision == 1 && W.pn_x3)
            T2P_TRY(launch_gemm_x3(fin, kin, W.pn_x3, W.pn_scale, W.pn_b, ws.P, D, 0, n, kin, D, 1, st, nullptr, 0, gslot(G_GEMM_IN)));
        else
            T2P_TRY(launch_gemm(fin, kin, W.pn_w, W.pn_b, ws.P, D, 0, n, kin, D, 1, st));
        T2P_TRY(launch_rownorm(ws.P, D, n, D, ws.cat, ldcat, slot * D, st));
        slot++;
    }
    if (cfg.use_color && cfg.color_embed) {
        T2P_TRY(launch_gather_rownorm(W.color_embedding, cfg.color_idx + o_lo, n, D, ws.cat, ldcat, slot * D, st));
        slot++;
    } else if (cfg.use_color) {
        T2P_TRY(launch_mlp3_norm(mean_rgb, n, W.col_w1, W.col_b1, W.col_w2, W.col_b2, D, ws.cat, ldcat, slot * D, st));
        slot++;
    }
    if (cfg.use_position) {
        T2P_TRY(launch_mlp3_norm(center, n, W.pos_w1, W.pos_b1, W.pos_w2, W.pos_b2, D, ws.cat, ldcat, slot * D, st));
        slot++;
    }
    const float* emb = ws.cat;  // single feature: embeddings[0] is returned un-merged (object_encoder.py:137-140)
    if (nfeat > 1) {
        if (cfg.precision == 1 && W.merge_x3)
            T2P_TRY(launch_gemm_x3(ws.cat, ldcat, W.merge_x3, W.merge_scale, W.merge_b, ws.emb, D, 0, n, ldcat, D, 1, st, nullptr, 0, gslot(G_GEMM_IN)));
        else
            T2P_TRY(launch_gemm(ws.cat, ldcat, W.merge_w, W.merge_b, ws.emb, D, 0, n, ldcat, D, 1, st));
        emb = ws.emb;
    }
    if (cfg.objects_only) {  // the fine stage consumes ObjectEncoder.forward's output as is
        T2P_TRY(copy_trace(tr->obj_emb + trace_obj0 * D, emb, (size_t)n * D, st));
        T2P_TRY(launch_guard_check(guard, cfg.overflow_flag, gbounds, st));
        return 0;
    }
    // ---- cell head: normalize, DynamicEdgeConv(k, max), global max pool, lin, normalize -------------------------
    T2P_TRY(launch_rownorm(emb, D, n, D, ws.embn, D, 0, st));
    if (cfg.precision == 1 && W.g_wp_x3 && W.g_wq_x3) {
        T2P_TRY(launch_gemm_x3(ws.embn, D, W.g_wp_x3, W.g_wp_scale, W.g_bp, ws.P, D, 0, n, D, D, 0, st, nullptr, 0, gslot(G_GEMM_IN)));
        T2P_TRY(launch_gemm_x3(ws.embn, D, W.g_wq_x3, W.g_wq_scale, nullptr, ws.Q, D, 0, n, D, D, 0, st, nullptr, 0, gslot(G_GEMM_IN)));
    } else {
        T2P_TRY(launch_gemm(ws.embn, D, W.g_wp, W.g_bp, ws.P, D, 0, n, D, D, 0, st));
        T2P_TRY(launch_gemm(ws.embn, D, W.g_wq, nullptr, ws.Q, D, 0, n, D, D, 0, st));
    }
    T2P_TRY(launch_knn(ws.embn, D, ws.seg_ptr, (int)nb, max_cell, cfg.knn_k, ws.knn, st));
    {
        WsParams p{};
        p.A = ws.Q;
        p.lda = D;
        p.Bc = ws.P;
        p.W = W.g_w2;
        p.W_x3 = cfg.precision == 1 ? W.g_w2_x3 : nullptr;   // f16x3 image of layer 2 (absent: fp32 MFMA)
        p.amax_out = gslot(G_GEMM_IN);                                      // (here: the rows this kernel splits itself)
        p.ldw = D;
        p.bias = W.g_b2;
        p.out = ws.x1;
        p.ldo = D;
        p.relu = 1;
        // small calls (the reference's 64-cell batches: ~1,000 objects = 31 groups of 32 on 256 CUs): groups of 8 destinations
        p.knn_group = (n + 31) / 32 < num_cus() ? 8 : 32;
        p.n_groups = (n + p.knn_group - 1) / p.knn_group;
        p.knn_idx = ws.knn;
        p.knn_k = cfg.knn_k;
        p.n_dst = n;
        p.mean = cfg.variation == 1;  // cell_retrieval.py:50-54: DynamicEdgeConv(aggr="mean") + global_mean_pool
        T2P_TRY(launch_ws(WS_EDGE_KNN, D, D, p, st));
    }
    T2P_TRY(launch_segmax(ws.x1, D, ws.seg_ptr, (int)nb, ws.pool, cfg.variation == 1, st));
    T2P_TRY(launch_gemm(ws.pool, D, W.lin_w1, W.lin_b1, ws.l1, D, 0, nb, D, D, 1, st));
    T2P_TRY(launch_gemm(ws.l1, D, W.lin_w2, W.lin_b2, ws.l2, D, 0, nb, D, D, 1, st));
    T2P_TRY(launch_rownorm(ws.l2, D, nb, D, out, D, 0, st));
    T2P_TRY(launch_guard_check(guard, cfg.overflow_flag, gbounds, st));

    if (tr && run_pointnet) {
        for (int l = 0; l < 3; l++) {
            T2P_TRY(copy_trace(tr->fps_idx[l] ? tr->fps_idx[l] + trace_obj0 * g.nc[l] : nullptr, ws.gt.fps_idx[l],
                               (size_t)n * g.nc[l], st));
            T2P_TRY(copy_trace(tr->nbr[l] ? tr->nbr[l] + trace_obj0 * g.nc[l] * 32 : nullptr, ws.gt.nbr[l],
                               (size_t)n * g.nc[l] * 32, st));
            T2P_TRY(copy_trace(tr->cnt[l] ? tr->cnt[l] + trace_obj0 * g.nc[l] : nullptr, ws.gt.cnt[l],
                               (size_t)n * g.nc[l], st));
            if (tr->sa_out[l] && n > 0) {   // [features | xyz 0] of every row; the pad columns behind are left as the caller set them
                hipError_t e = hipMemcpy2DAsync(tr->sa_out[l] + trace_obj0 * g.nc[l] * Geo::LD[l], Geo::LD[l] * sizeof(float), ws.F[l],
                                                Geo::LD[l] * sizeof(float), (Geo::C[l] + 4) * sizeof(float), (size_t)n * g.nc[l],
                                                hipMemcpyDeviceToDevice, st);
                if (e != hipSuccess) {
                    set_error("encode_cells: trace copy failed: %s", hipGetErrorString(e));
                    return (int)e;
                }
            }
        }
        T2P_TRY(copy_trace(tr->features0 ? tr->features0 + trace_obj0 * 1024 : nullptr, ws.f0, (size_t)n * 1024, st));
        T2P_TRY(copy_trace(tr->features1 ? tr->features1 + trace_obj0 * 512 : nullptr, ws.f1, (size_t)n * 512, st));
        T2P_TRY(copy_trace(tr->features2 ? tr->features2 + trace_obj0 * 256 : nullptr, ws.f2, (size_t)n * 256, st));
    }
    if (tr) {
        T2P_TRY(copy_trace(tr->obj_emb ? tr->obj_emb + trace_obj0 * D : nullptr, emb, (size_t)n * D, st));
        // knn indices are chunk-local object rows; tests use a single chunk or add the chunk offset themselves
        T2P_TRY(copy_trace(tr->knn_idx ? tr->knn_idx + trace_obj0 * cfg.knn_k : nullptr, ws.knn,
                           (size_t)n * cfg.knn_k, st));
    }
    return 0;
}

// Whole cells per chunk, at most `chunk` objects (a single larger cell still forms its own chunk).
int64_t next_chunk_end(const int32_t* cp, int64_t c0, int64_t n_cells, int chunk) {
    int64_t c1 = c0 + 1;
    while (c1 < n_cells && (cp[c1 + 1] - cp[c0]) <= chunk) c1++;
    return c1;
}

}  // namespace
}  // namespace t2p

using namespace t2p;

extern "C" {

int t2p_abi_version(void) { return T2P_ABI_VERSION; }

void t2p_profile_enable(int on) {
    std::lock_guard<std::mutex> lock(g_prof_mu);
    g_prof_on = on != 0;
}

// Measurement hook (profiles/energy_table.py): launches recorded under `scope` are issued `reps` times back to back.
void t2p_profile_repeat(const char* scope, int reps) {
    std::lock_guard<std::mutex> lock(g_prof_mu);
    g_repeat_n = 1;
    if (scope != nullptr && reps > 1) {
        strncpy(g_repeat_name, scope, sizeof(g_repeat_name) - 1);
        g_repeat_name[sizeof(g_repeat_name) - 1] = 0;
        g_repeat_n = reps;
    }
}

// Waits for the recorded launches, writes one line per kernel name: "<name> <launches> <total_ms>\n", clears.
int t2p_profile_report(char* buf, size_t n) {
    std::lock_guard<std::mutex> lock(g_prof_mu);
    size_t off = 0;
    if (n > 0) buf[0] = 0;
    for (int i = 0; i < g_prof_n; i++) {
        if (g_prof[i].name == nullptr) continue;
        int cnt = 0;
        double tot = 0.0;
        for (int j = i; j < g_prof_n; j++) {
            if (g_prof[j].name == nullptr || strcmp(g_prof[j].name, g_prof[i].name) != 0) continue;
            float ms = 0.f;
            if (hipEventSynchronize(g_prof[j].b) == hipSuccess && hipEventElapsedTime(&ms, g_prof[j].a, g_prof[j].b) == hipSuccess) {
                tot += ms;
                cnt++;
            }
            (void)hipEventDestroy(g_prof[j].a);
            (void)hipEventDestroy(g_prof[j].b);
            if (j != i) g_prof[j].name = nullptr;
        }
        int w = snprintf(buf + off, off < n ? n - off : 0, "%s %d %.6f\n", g_prof[i].name, cnt, tot);
        if (w > 0) off += (size_t)w;
        g_prof[i].name = nullptr;
    }
    g_prof_n = 0;
    return 0;
}
const char* t2p_last_error(void) { return g_err; }

size_t t2p_encode_cells_workspace_bytes(int64_t n_obj, int64_t n_cells, const t2p_cell_config* cfg) {
    if (cfg == nullptr || n_obj <= 0) return 256;
    // a chunk holds at most max(chunk_objects, largest cell) objects; the caller may not know the largest cell here,
    // so size for min(n_obj, chunk) and let t2p_encode_cells re-check against the actual partition.
    int64_t n = default_chunk(*cfg);
    if (n > n_obj) n = n_obj;
    int64_t nb = n_cells < n ? n_cells : n;
    Bump b{nullptr, 0, 0};
    return carve(b, n, nb > 0 ? nb : 1, *cfg, nullptr) + 256;
}

int t2p_encode_cells(const float* xyz, const float* rgb, const float* center, const float* mean_rgb,
                     const int32_t* cell_ptr_host, const int32_t* cell_ptr, int64_t n_obj, int64_t n_cells,
                     const t2p_cell_weights* w, const t2p_cell_config* cfg, float* out, const t2p_cell_trace* trace,
                     void* workspace, size_t workspace_bytes, t2p_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    T2P_TRY(check_cfg(cfg));
    T2P_CHECK_ARG(w != nullptr && cell_ptr_host != nullptr && cell_ptr != nullptr, "encode_cells: NULL argument");
    T2P_CHECK_ARG(n_cells >= 0 && n_obj >= 0, "encode_cells: negative size");
    if (n_cells == 0) return 0;  // an empty batch has no output rows (and its buffers may be NULL)
    if (cfg->objects_only)
        T2P_CHECK_ARG(trace != nullptr && trace->obj_emb != nullptr, "encode_cells: objects_only needs trace->obj_emb");
    else
        T2P_CHECK_ARG(out != nullptr, "encode_cells: out is NULL");
    T2P_CHECK_ARG(!cfg->class_embed || w->class_embedding != nullptr, "encode_cells: class_embedding weights missing");
    T2P_CHECK_ARG(!cfg->color_embed || w->color_embedding != nullptr, "encode_cells: color_embedding weights missing");
    if (cfg->precision == 1)
        T2P_CHECK_ARG(w->sa_w2_x3[0] && w->sa_w2_x3[1] && w->sa_w2_x3[2] && w->sa_b2_x3[0] && w->sa_b2_x3[1] &&
                          w->sa_b2_x3[2] && w->sa_w2_scale[0] > 0.f && w->sa_w2_scale[1] > 0.f && w->sa_w2_scale[2] > 0.f &&
                          w->sa_w1_x3[1] && w->sa_w1_x3[2] &&
                          w->ga_w1_x3 && w->ga_w2_x3,
                      "encode_cells: precision = f16x3 needs the packed *_x3 weight images");
    if (cfg->precision == 1 && cfg->overflow_flag != nullptr)
        T2P_CHECK_ARG(w->ga_w1_l1 > 0.f && w->ga_b1_absmax >= 0.f && w->sa_a1_l1 > 0.f && w->sa_wp_l1[0] >= 0.f,
                      "encode_cells: the fp16-range guard needs the weight norms of t2p_cell_weights (packing.py)");
    T2P_CHECK_ARG(cell_ptr_host[0] == 0 && cell_ptr_host[n_cells] == n_obj,
                  "encode_cells: cell_ptr must start at 0 and end at n_obj=%lld (got %d..%d)", (long long)n_obj,
                  cell_ptr_host[0], cell_ptr_host[n_cells]);
    for (int64_t c = 0; c < n_cells; c++)
        T2P_CHECK_ARG(cell_ptr_host[c + 1] > cell_ptr_host[c],
                      "encode_cells: cell %lld is empty (the reference asserts >= 1 object per cell, "
                      "dataloading/kitti360pose/utils.py:108)", (long long)c);
    T2P_CHECK_ARG((((uintptr_t)xyz | (uintptr_t)rgb | (uintptr_t)workspace) & 15) == 0,
                  "encode_cells: xyz, rgb and workspace must be 16-byte aligned");
    const int chunk = default_chunk(*cfg);
    for (int64_t c0 = 0; c0 < n_cells;) {
        const int64_t c1 = next_chunk_end(cell_ptr_host, c0, n_cells, chunk);
        const int32_t o_lo = cell_ptr_host[c0], o_hi = cell_ptr_host[c1];
        const int64_t n = o_hi - o_lo, nb = c1 - c0;
        int max_cell = 0;
        for (int64_t c = c0; c < c1; c++) {
            const int m = cell_ptr_host[c + 1] - cell_ptr_host[c];
            max_cell = m > max_cell ? m : max_cell;
        }
        T2P_CHECK_ARG(n <= T2P_MAX_CHUNK_OBJECTS,
                      "encode_cells: cell %lld alone holds %lld objects; a chunk (whole cells) is limited to %d objects",
                      (long long)c0, (long long)n, T2P_MAX_CHUNK_OBJECTS);
        Bump b{(char*)workspace, 0, workspace_bytes};
        CellWs ws;
        const size_t need = carve(b, n, nb, *cfg, &ws);
        if (need > workspace_bytes) {
            set_error("encode_cells: workspace %zu B < %zu B needed for a chunk of %lld objects / %lld cells",
                      workspace_bytes, need, (long long)n, (long long)nb);
            return T2P_E_WORKSPACE;
        }
        const int64_t P3 = (int64_t)cfg->n_pts * 3;
        T2P_TRY(encode_chunk(xyz + o_lo * P3, rgb + o_lo * P3, center + (int64_t)o_lo * 3, mean_rgb + (int64_t)o_lo * 3,
                             cell_ptr + c0, o_lo, n, nb, max_cell, *w, *cfg, out ? out + c0 * cfg->embed_dim : nullptr, trace, o_lo, ws,
                             st));
        c0 = c1;
    }
    return 0;
}

size_t t2p_encode_text_workspace_bytes(int64_t batch, int32_t vocab, int32_t embed_dim) {
    const size_t D = embed_dim;
    return align_up((size_t)2 * (vocab + 1) * 4 * D * sizeof(float), 256) + align_up((size_t)2 * batch * D * sizeof(float), 256) +
           align_up((size_t)batch * D * sizeof(float), 256) + 256;
}

int t2p_encode_text(const int32_t* tokens, const int32_t* lengths, int64_t batch, int32_t max_len, int32_t vocab,
                    int32_t embed_dim, const t2p_text_weights* w, float* out_raw, float* out, void* workspace,
                    size_t workspace_bytes, t2p_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    T2P_CHECK_ARG(w != nullptr && tokens != nullptr && lengths != nullptr && out != nullptr, "encode_text: NULL argument");
    T2P_CHECK_ARG(batch >= 0 && batch < (1 << 30) && max_len >= 1 && vocab >= 1, "encode_text: bad sizes");
    if (batch == 0) return 0;
    const size_t need = t2p_encode_text_workspace_bytes(batch, vocab, embed_dim);
    if (workspace == nullptr || workspace_bytes < need) {
        set_error("encode_text: workspace %zu B < required %zu B", workspace_bytes, need);
        return T2P_E_WORKSPACE;
    }
    const int D = embed_dim;
    Bump b{(char*)workspace, 0, workspace_bytes};
    const int rows = vocab + 1;   // + one zero row per direction (lstm.hip)
    float* table = b.take<float>((size_t)2 * rows * 4 * D);
    float* hdir = b.take<float>((size_t)2 * batch * D);
    float* raw = out_raw ? out_raw : b.take<float>((size_t)batch * D);
    // gate table [dir][V][4D] = embedding [V][D] x W_ih^T [D][4D] + (b_ih + b_hh)
    for (int dir = 0; dir < 2; dir++) {
        T2P_TRY(launch_gemm(w->embedding, D, w->w_ih + (size_t)dir * D * 4 * D, w->bias + (size_t)dir * 4 * D,
                            table + (size_t)dir * rows * 4 * D, 4 * D, 0, vocab, D, 4 * D, 0, st));
        hipError_t e = hipMemsetAsync(table + ((size_t)dir * rows + vocab) * 4 * D, 0, (size_t)4 * D * sizeof(float), st);
        if (e != hipSuccess) {
            set_error("encode_text: memset failed: %s", hipGetErrorString(e));
            return (int)e;
        }
    }
    T2P_TRY(launch_bilstm_impl(table, w->w_hh, w->w_hh_x3, w->w_hh_scale, tokens, lengths, (int)batch, max_len, rows, D, hdir, raw, st));
    T2P_TRY(launch_rownorm(raw, D, batch, D, out, D, 0, st));
    return 0;
}

int t2p_lstm_cell_forward(const float* pre, const float* gate_table, const int32_t* tokens, const int32_t* lengths,
                          int64_t batch, int32_t max_len, int32_t embed_dim, int32_t step, int32_t reverse, const float* c_prev,
                          const float* h_prev, float* gates, float* c, float* h, t2p_stream_t stream) {
    T2P_CHECK_ARG(pre && gate_table && tokens && lengths && c_prev && h_prev && gates && c && h, "lstm_cell_forward: NULL argument");
    T2P_CHECK_ARG(batch >= 0 && max_len >= 1 && embed_dim >= 1 && step >= 0 && step < max_len, "lstm_cell_forward: bad sizes");
    return launch_lstm_cell_fwd(pre, gate_table, tokens, lengths, batch, max_len, embed_dim, step, reverse, c_prev, h_prev, gates,
                                c, h, (hipStream_t)stream);
}

int t2p_lstm_cell_backward(const float* dh_gemm, const float* dh_carry_in, const float* dc_in, const float* gates,
                           const float* c_prev, const float* c, const int32_t* lengths, int64_t batch, int32_t embed_dim,
                           int32_t step, float* d_pre, float* dc_out, float* dh_carry_out, t2p_stream_t stream) {
    T2P_CHECK_ARG(dh_carry_in && dc_in && gates && c_prev && c && lengths && d_pre && dc_out && dh_carry_out,
                  "lstm_cell_backward: NULL argument");
    T2P_CHECK_ARG(batch >= 0 && embed_dim >= 1 && step >= 0, "lstm_cell_backward: bad sizes");
    return launch_lstm_cell_bwd(dh_gemm, dh_carry_in, dc_in, gates, c_prev, c, lengths, batch, embed_dim, step, d_pre, dc_out,
                                dh_carry_out, (hipStream_t)stream);
}

int t2p_lstm_train_forward(const float* gate_table, const float* w_hh_k, const int32_t* tokens, const int32_t* lengths,
                           int64_t batch, int32_t max_len, int32_t embed_dim, int32_t reverse, float* gates, float* cs, float* hs,
                           float* pre_ws, t2p_stream_t stream) {
    T2P_CHECK_ARG(gate_table && w_hh_k && tokens && lengths && gates && cs && hs && pre_ws, "lstm_train_forward: NULL argument");
    T2P_CHECK_ARG(batch >= 0 && max_len >= 1 && embed_dim >= 1, "lstm_train_forward: bad sizes");
    if (embed_dim % 4 != 0) {
        set_error("lstm_train_forward: embed_dim=%d is not a multiple of 4", embed_dim);
        return T2P_E_UNSUPPORTED;
    }
    if (batch == 0) return 0;
    const int D = embed_dim;
    const size_t bd = (size_t)batch * D;
    hipStream_t st = (hipStream_t)stream;
    for (int s = 0; s < max_len; s++) {       // the time loop of modules._LstmTrainFn.forward, one launch pair per step
        T2P_TRY(launch_gemm_skinny(hs + s * bd, D, w_hh_k, pre_ws, 4 * D, batch, D, 4 * D, st));
        T2P_TRY(launch_lstm_cell_fwd(pre_ws, gate_table, tokens, lengths, batch, max_len, D, s, reverse, cs + s * bd, hs + s * bd,
                                     gates + (size_t)s * batch * 4 * D, cs + (s + 1) * bd, hs + (s + 1) * bd, st));
    }
    return 0;
}

int t2p_lstm_train_backward(const float* dh_last, const float* w_hh_t, const float* gates, const float* cs, const int32_t* lengths,
                            int64_t batch, int32_t max_len, int32_t embed_dim, float* d_pre, float* ws, t2p_stream_t stream) {
    T2P_CHECK_ARG(dh_last && w_hh_t && gates && cs && lengths && d_pre && ws, "lstm_train_backward: NULL argument");
    T2P_CHECK_ARG(batch >= 0 && max_len >= 1 && embed_dim >= 1, "lstm_train_backward: bad sizes");
    if (embed_dim % 4 != 0) {
        set_error("lstm_train_backward: embed_dim=%d is not a multiple of 4", embed_dim);
        return T2P_E_UNSUPPORTED;
    }
    if (batch == 0) return 0;
    const int D = embed_dim;
    const size_t bd = (size_t)batch * D;
    hipStream_t st = (hipStream_t)stream;
    float* dc[2] = {ws, ws + bd};              // ws: [5][B][D] = dc ping-pong | dh_carry ping-pong | dh_gemm
    float* carry[2] = {ws + 2 * bd, ws + 3 * bd};
    float* dh_gemm = ws + 4 * bd;
    hipError_t e = hipMemsetAsync(dc[0], 0, bd * sizeof(float), st);
    if (e == hipSuccess) e = hipMemcpyAsync(carry[0], dh_last, bd * sizeof(float), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) {
        set_error("lstm_train_backward: %s", hipGetErrorString(e));
        return (int)e;
    }
    int cur = 0;
    for (int s = max_len - 1; s >= 0; s--) {
        T2P_TRY(launch_lstm_cell_bwd(s == max_len - 1 ? nullptr : dh_gemm, carry[cur], dc[cur], gates + (size_t)s * batch * 4 * D,
                                     cs + s * bd, cs + (s + 1) * bd, lengths, batch, D, s, d_pre + (size_t)s * batch * 4 * D,
                                     dc[cur ^ 1], carry[cur ^ 1], st));
        if (s > 0) T2P_TRY(launch_gemm_skinny(d_pre + (size_t)s * batch * 4 * D, 4 * D, w_hh_t, dh_gemm, D, batch, 4 * D, D, st));
        cur ^= 1;
    }
    return 0;
}

int t2p_pairwise_ranking(const float* scores, int32_t batch, float margin, float* row_loss, float* d_scores, float* row_count,
                         t2p_stream_t stream) {
    T2P_CHECK_ARG(scores && row_loss && d_scores && row_count, "pairwise_ranking: NULL argument");
    T2P_CHECK_ARG(batch >= 0 && batch <= 32768, "pairwise_ranking: batch=%d outside [0, 32768]", batch);
    return launch_pairwise_ranking(scores, batch, margin, row_loss, d_scores, row_count, (hipStream_t)stream);
}

size_t t2p_bn_train_workspace_bytes(int64_t rows, int32_t n_seg, int32_t channels) {
    return bn_train_workspace_bytes(rows, n_seg, channels);
}

int t2p_bn_relu_train_forward(const float* x, const int32_t* seg_ptr, int32_t n_seg, int64_t rows, int32_t channels,
                              const float* gamma, const float* beta, float eps, int32_t relu, float* y, float* mean,
                              float* invstd, float* var_unbiased, void* workspace, size_t workspace_bytes,
                              t2p_stream_t stream) {
    T2P_CHECK_ARG(x && seg_ptr && gamma && beta && y && mean && invstd && var_unbiased, "bn_relu_train_forward: NULL argument");
    T2P_CHECK_ARG(n_seg >= 0 && channels >= 1 && rows >= 0, "bn_relu_train_forward: bad sizes");
    if (n_seg > 0 && (workspace == nullptr || workspace_bytes < bn_train_workspace_bytes(rows, n_seg, channels))) {
        set_error("bn_relu_train_forward: workspace %zu B < required %zu B", workspace_bytes,
                  bn_train_workspace_bytes(rows, n_seg, channels));
        return T2P_E_WORKSPACE;
    }
    return launch_bn_relu_train_forward(x, seg_ptr, n_seg, rows, channels, gamma, beta, eps, relu, y, mean, invstd,
                                        var_unbiased, (double*)workspace, (hipStream_t)stream);
}

int t2p_bn_relu_train_backward(const float* dy, const float* x, const float* beta, const int32_t* seg_ptr, int32_t n_seg,
                               int64_t rows, int32_t channels, const float* mean, const float* invstd, const float* gamma,
                               int32_t relu, float* dx, float* dgamma_seg, float* dbeta_seg, void* workspace,
                               size_t workspace_bytes, t2p_stream_t stream) {
    T2P_CHECK_ARG(dy && x && beta && seg_ptr && mean && invstd && gamma && dx && dgamma_seg && dbeta_seg,
                  "bn_relu_train_backward: NULL argument");
    T2P_CHECK_ARG(n_seg >= 0 && channels >= 1 && rows >= 0, "bn_relu_train_backward: bad sizes");
    if (n_seg > 0 && (workspace == nullptr || workspace_bytes < bn_train_workspace_bytes(rows, n_seg, channels))) {
        set_error("bn_relu_train_backward: workspace %zu B < required %zu B", workspace_bytes,
                  bn_train_workspace_bytes(rows, n_seg, channels));
        return T2P_E_WORKSPACE;
    }
    return launch_bn_relu_train_backward(dy, x, beta, seg_ptr, n_seg, rows, channels, mean, invstd, gamma, relu, dx, dgamma_seg,
                                         dbeta_seg, (double*)workspace, (hipStream_t)stream);
}

int t2p_edge_features_forward(const float* x, const float* pos, const float* pos_c, const int32_t* src, const int32_t* dst,
                              int64_t n_edges, int32_t channels, int32_t width, float* out, t2p_stream_t stream) {
    T2P_CHECK_ARG(pos && pos_c && src && dst && out && (x || channels == 0), "edge_features_forward: NULL argument");
    T2P_CHECK_ARG(n_edges >= 0 && channels >= 0, "edge_features_forward: bad sizes");
    return launch_edge_feat_fwd(x, pos, pos_c, src, dst, n_edges, channels, width, out, (hipStream_t)stream);
}
int t2p_edge_features_backward(const float* d_out, const int32_t* src, int64_t n_edges, int32_t channels, int32_t width, float* dx,
                               t2p_stream_t stream) {
    T2P_CHECK_ARG(d_out && src && (dx || channels == 0), "edge_features_backward: NULL argument");
    T2P_CHECK_ARG(n_edges >= 0 && channels >= 0, "edge_features_backward: bad sizes");
    return launch_edge_feat_bwd(d_out, src, n_edges, channels, width, dx, (hipStream_t)stream);
}
int t2p_pair_features_forward(const float* x, const int32_t* tgt, const int32_t* src, int64_t n_edges, int32_t dim, float* out,
                              t2p_stream_t stream) {
    T2P_CHECK_ARG(x && tgt && src && out, "pair_features_forward: NULL argument");
    T2P_CHECK_ARG(n_edges >= 0 && dim >= 1, "pair_features_forward: bad sizes");
    return launch_pair_feat_fwd(x, tgt, src, n_edges, dim, out, (hipStream_t)stream);
}
int t2p_pair_features_backward(const float* d_out, const int32_t* tgt, const int32_t* src, int64_t n_edges, int32_t dim,
                               float* dx, t2p_stream_t stream) {
    T2P_CHECK_ARG(d_out && tgt && src && dx, "pair_features_backward: NULL argument");
    T2P_CHECK_ARG(n_edges >= 0 && dim >= 1, "pair_features_backward: bad sizes");
    return launch_pair_feat_bwd(d_out, tgt, src, n_edges, dim, dx, (hipStream_t)stream);
}
int t2p_rownorm_backward(const float* x, const float* dy, int64_t n_rows, int32_t dim, float* dx, t2p_stream_t stream) {
    T2P_CHECK_ARG(x && dy && dx, "rownorm_backward: NULL argument");
    T2P_CHECK_ARG(n_rows >= 0 && dim >= 1, "rownorm_backward: bad sizes");
    return launch_rownorm_bwd(x, dy, n_rows, dim, dx, (hipStream_t)stream);
}

int t2p_segment_mean_forward(const float* x, const int32_t* seg_ptr, int32_t n_seg, int32_t channels, float* out,
                             t2p_stream_t stream) {
    T2P_CHECK_ARG(x && seg_ptr && out, "segment_mean_forward: NULL argument");
    T2P_CHECK_ARG(n_seg >= 0 && channels >= 1, "segment_mean_forward: bad sizes");
    return launch_segment_mean(x, seg_ptr, n_seg, channels, out, (hipStream_t)stream);
}
int t2p_segment_mean_backward(const float* dout, const int32_t* seg_ptr, int32_t n_seg, int32_t channels, float* dx,
                              t2p_stream_t stream) {
    T2P_CHECK_ARG(dout && seg_ptr && dx, "segment_mean_backward: NULL argument");
    T2P_CHECK_ARG(n_seg >= 0 && channels >= 1, "segment_mean_backward: bad sizes");
    return launch_segment_mean_backward(dout, seg_ptr, n_seg, channels, dx, (hipStream_t)stream);
}

int t2p_segment_max_forward(const float* x, const int32_t* seg_ptr, int32_t n_seg, int32_t channels, float* out, int32_t* arg,
                            t2p_stream_t stream) {
    T2P_CHECK_ARG(x && seg_ptr && out && arg, "segment_max_forward: NULL argument");
    T2P_CHECK_ARG(n_seg >= 0 && channels >= 1, "segment_max_forward: bad sizes");
    return launch_segment_max(x, seg_ptr, n_seg, channels, out, arg, (hipStream_t)stream);
}

int t2p_segment_max_backward(const float* dout, const int32_t* arg, const int32_t* seg_ptr, int32_t n_seg, int32_t channels,
                             float* dx, t2p_stream_t stream) {
    T2P_CHECK_ARG(dout && arg && seg_ptr && dx, "segment_max_backward: NULL argument");
    T2P_CHECK_ARG(n_seg >= 0 && channels >= 1, "segment_max_backward: bad sizes");
    return launch_segment_max_backward(dout, arg, seg_ptr, n_seg, channels, dx, (hipStream_t)stream);
}

int t2p_hardest_ranking(const float* scores, int32_t batch, float margin, float* best, int32_t* where, float* d_scores,
                        t2p_stream_t stream) {
    T2P_CHECK_ARG(scores && best && where && d_scores, "hardest_ranking: NULL argument");
    T2P_CHECK_ARG(batch >= 0 && batch <= 32768, "hardest_ranking: batch=%d outside [0, 32768]", batch);
    return launch_hardest_ranking(scores, batch, margin, best, where, d_scores, (hipStream_t)stream);
}

size_t t2p_sim_topk_workspace_bytes(int64_t nq, int64_t nc, int32_t k) { return sim_topk_workspace_bytes(nq, nc, k); }

int t2p_sim_topk(const float* queries, const float* cells, int64_t nq, int64_t nc, int32_t dim, int32_t k,
                 int64_t index_offset, int64_t* out_idx, double* out_score, void* workspace, size_t workspace_bytes,
                 t2p_stream_t stream) {
    T2P_CHECK_ARG(queries != nullptr && cells != nullptr && out_idx != nullptr && out_score != nullptr,
                  "sim_topk: NULL argument");
    T2P_CHECK_ARG(nq >= 0 && nc >= 0, "sim_topk: negative size");
    return launch_sim_topk(queries, cells, nq, nc, dim, k, index_offset, out_idx, out_score, workspace, workspace_bytes,
                           (hipStream_t)stream);
}

int t2p_sample_group(const float* xyz, int64_t n_obj, int32_t n_pts, const float* radius_host,
                     uint8_t* const* fps_idx, uint8_t* const* nbr, uint8_t* const* cnt, t2p_stream_t stream) {
    T2P_CHECK_ARG(xyz && radius_host && fps_idx && nbr && cnt, "sample_group: NULL argument");
    Geo g(n_pts);
    GroupTables gt{};
    gt.self_loops = 0;
    for (int l = 0; l < 3; l++) {
        gt.fps_idx[l] = fps_idx[l];
        gt.nbr[l] = nbr[l];
        gt.cnt[l] = cnt[l];
        gt.rows[l] = nullptr;
        gt.n_rows[l] = nullptr;
        gt.n_dense[l] = g.nd[l];
        gt.n_cent[l] = g.nc[l];
    }
    return launch_sample_group(xyz, n_obj, n_pts, radius_host, gt, (hipStream_t)stream);
}

int t2p_group_rows(const float* xyz, int64_t n_obj, int32_t n_pts, const float* radius_host, int32_t self_loops,
                   uint8_t* const* fps_idx, uint16_t* const* rows, uint16_t* const* n_rows, t2p_stream_t stream) {
    T2P_CHECK_ARG(xyz && radius_host && fps_idx && rows && n_rows, "group_rows: NULL argument");
    Geo g(n_pts);
    GroupTables gt{};
    gt.self_loops = self_loops ? 1 : 0;
    for (int l = 0; l < 3; l++) {
        T2P_CHECK_ARG(fps_idx[l] && rows[l] && n_rows[l], "group_rows: NULL table of level %d", l);
        gt.fps_idx[l] = fps_idx[l];
        gt.rows[l] = rows[l];
        gt.n_rows[l] = n_rows[l];
        gt.n_dense[l] = g.nd[l];
        gt.n_cent[l] = g.nc[l];
    }
    return launch_sample_group(xyz, n_obj, n_pts, radius_host, gt, (hipStream_t)stream);
}

int t2p_edge_counts(const uint16_t* rows, const uint16_t* n_rows, const int32_t* first_obj, int64_t n_obj, int32_t n_dense,
                    int32_t n_cent, int32_t self_loops, int32_t* counts, t2p_stream_t stream) {
    T2P_CHECK_ARG(n_obj >= 0 && (n_obj == 0 || (rows && n_rows && first_obj && counts)), "edge_counts: NULL argument");
    return launch_edge_counts(rows, n_rows, first_obj, n_obj, n_dense, n_cent, self_loops, counts, (hipStream_t)stream);
}

int t2p_edge_expand(const uint16_t* rows, const uint16_t* n_rows, const int32_t* first_obj, const int32_t* cent_ptr, int64_t n_obj,
                    int32_t n_dense, int32_t n_cent, int32_t self_loops, int32_t* src, int32_t* dst, t2p_stream_t stream) {
    T2P_CHECK_ARG(n_obj >= 0 && (n_obj == 0 || (rows && n_rows && first_obj && cent_ptr)), "edge_expand: NULL argument");
    return launch_edge_expand(rows, n_rows, first_obj, cent_ptr, n_obj, n_dense, n_cent, self_loops, src, dst, (hipStream_t)stream);
}

int t2p_pack_objects(const float* raw_xyz, const float* raw_rgb, const int32_t* obj_ptr, const int32_t* sample_idx,
                     const float* rot_cos_sin, int64_t n_obj, int32_t n_pts, float* xyz, float* rgb, float* center, float* mean_rgb,
                     t2p_stream_t stream) {
    T2P_CHECK_ARG(raw_xyz && raw_rgb && obj_ptr && sample_idx && xyz && rgb && center && mean_rgb, "pack_objects: NULL argument");
    T2P_CHECK_ARG(n_obj >= 0 && n_pts >= 1, "pack_objects: bad sizes");
    return launch_pack_objects(raw_xyz, raw_rgb, obj_ptr, sample_idx, rot_cos_sin, n_obj, n_pts, xyz, rgb, center, mean_rgb,
                               (hipStream_t)stream);
}

int t2p_pack_scene_objects(const float* raw_xyz, const float* raw_rgb, const int32_t* obj_ptr, const int32_t* obj_id,
                           const uint64_t* key, const float* scene_center, const float* scene_color, int64_t n_out, int32_t n_pts,
                           float* xyz, float* rgb, float* center, float* mean_rgb, int32_t* sample_idx_out, t2p_stream_t stream) {
    T2P_CHECK_ARG(n_out >= 0 && n_pts >= 1, "pack_scene_objects: bad sizes");
    T2P_CHECK_ARG(n_out == 0 || (raw_xyz && obj_ptr && obj_id && key && xyz), "pack_scene_objects: NULL argument");
    T2P_CHECK_ARG(n_out == 0 || rgb == nullptr || raw_rgb != nullptr, "pack_scene_objects: rgb wanted but raw_rgb is NULL");
    T2P_CHECK_ARG(n_out == 0 || ((center == nullptr || scene_center != nullptr) && (mean_rgb == nullptr || scene_color != nullptr)),
                  "pack_scene_objects: center / mean_rgb wanted but the scene table is NULL");
    return launch_pack_scene(raw_xyz, raw_rgb, obj_ptr, obj_id, key, scene_center, scene_color, n_out, n_pts, xyz, rgb, center,
                             mean_rgb, sample_idx_out, (hipStream_t)stream);
}

int t2p_dedup_rows(const float* xyz, const float* rgb, int64_t n_obj, int32_t n_pts, uint16_t* rows, uint16_t* n_rows,
                   t2p_stream_t stream) {
    T2P_CHECK_ARG(n_obj >= 0, "dedup_rows: negative size");
    T2P_CHECK_ARG((((uintptr_t)xyz | (uintptr_t)rgb | (uintptr_t)rows) & 15) == 0, "dedup_rows: xyz, rgb and rows must be 16-byte aligned");
    return launch_dedup_rows(xyz, rgb, n_obj, n_pts, rows, n_rows, (n_pts + 1) / 2, (hipStream_t)stream);
}

int t2p_knn(const float* x, int32_t dim, const int32_t* seg_ptr, int32_t n_seg, int32_t max_seg_rows, int32_t k,
            int32_t* out_idx, t2p_stream_t stream) {
    return launch_knn(x, dim, seg_ptr, n_seg, max_seg_rows, k, out_idx, (hipStream_t)stream);
}

int t2p_gemm(const float* a, int32_t lda, const float* w, const float* bias, float* c, int32_t ldc, int32_t c0,
             int64_t m, int32_t k, int32_t n, int32_t relu, t2p_stream_t stream) {
    return launch_gemm(a, lda, w, bias, c, ldc, c0, m, k, n, relu, (hipStream_t)stream);
}

size_t t2p_gemm_tn_workspace_bytes(int64_t m, int32_t k1, int32_t n) { return gemm_tn_workspace_bytes(m, k1, n); }

int t2p_gemm_tn(const float* a, int32_t lda, const float* b, int32_t ldb, float* c, int32_t ldc, int64_t m, int32_t k1, int32_t n,
                void* workspace, size_t workspace_bytes, t2p_stream_t stream) {
    return launch_gemm_tn(a, lda, b, ldb, c, ldc, m, k1, n, workspace, workspace_bytes, (hipStream_t)stream);
}

size_t t2p_linear_wgrad_workspace_bytes(int64_t m, int32_t k1, int32_t n) { return linear_wgrad_workspace_bytes(m, k1, n); }

int t2p_linear_wgrad_f32(const float* dy, int32_t lda, const float* x, int32_t ldb, float* dw, int32_t ldc, float* colsum, int64_t m,
                         int32_t k1, int32_t n, void* workspace, size_t workspace_bytes, t2p_stream_t stream) {
    return launch_linear_wgrad_f32(dy, lda, x, ldb, dw, ldc, colsum, m, k1, n, workspace, workspace_bytes, (hipStream_t)stream);
}

int t2p_rownorm(const float* x, int64_t n_rows, int32_t dim, float* out, t2p_stream_t stream) {
    return launch_rownorm(x, dim, n_rows, dim, out, dim, 0, (hipStream_t)stream);
}

}  // extern "C"

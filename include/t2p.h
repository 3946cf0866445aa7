/*
 * t2p.h -- C ABI of libt2p_hip.so: the MI355X (gfx950) implementation of the Text2Pos coarse
 * cell-retrieval forward path.
 *
 * The reference (mako443/Text2Pos-CVPR2022) is pure Python and has no FFI/plugin interface; its boundary for
 * this path is the Python class models/cell_retrieval.py::CellRetrievalNetwork plus the retrieval loop of
 * training/coarse.py::eval_epoch.  The entry points below are what a ctypes binding of that class calls
 * (INTEGRATION.md shows the stub); each one names the reference interface it replaces.
 *
 * Conventions
 *   - Every pointer is a DEVICE pointer (HBM) unless its name ends in _host.  Buffers are owned by the caller;
 *     the library never allocates or frees device memory and keeps no global mutable state besides the
 *     thread-local error string.  Scratch space is a caller-provided workspace, sized by the *_workspace_bytes
 *     functions.
 *   - Every call enqueues work on `stream` (a hipStream_t; NULL = the default stream) and returns without
 *     synchronising.  Re-entrant; one process per GPU for multi-GPU use.
 *   - Return value: 0 = success; > 0 = hipError_t of a failed launch; < 0 = T2P_E_* argument / workspace /
 *     unsupported-configuration error.  t2p_last_error() returns the message of the calling thread's last failure.
 *   - All floating-point tensors are fp32 row-major; weights are "k-major" ([in_features][out_features], i.e. the
 *     transpose of torch.nn.Linear.weight) with eval-mode BatchNorm folded in by the host
 *     (models/modules.py:21-29: Linear -> BatchNorm1d -> ReLU).
 */
#ifndef T2P_H
#define T2P_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define T2P_ABI_VERSION 27
#define T2P_DEFAULT_CHUNK_OBJECTS 65000 /* t2p_cell_config.chunk_objects == 0 */
#define T2P_MAX_CHUNK_OBJECTS 65535     /* 32-bit table offsets / 16-bit local indices: chunk_objects and the largest single
                                           cell may not exceed it (T2P_E_ARG otherwise).  The caller-provided workspace holds
                                           one chunk: ~0.6 MB per object, i.e. ~38 GB at the default chunk for a batch that
                                           fills it (t2p_encode_cells_workspace_bytes gives the exact figure; a second
                                           stream's call needs its own workspace) */
#define T2P_TUNING_MASK 0x1              /* t2p_cell_config.tuning: the bits that select a built plan */
#define T2P_E_ARG (-1)
#define T2P_E_WORKSPACE (-2)
#define T2P_E_UNSUPPORTED (-3)

typedef void* t2p_stream_t; /* hipStream_t */

int t2p_abi_version(void);
const char* t2p_last_error(void);

/* ------------------------------------------------------------------------------------------------------------
 * Cell branch: CellRetrievalNetwork.encode_objects  (models/cell_retrieval.py:77-107), i.e.
 * ObjectEncoder.forward (models/object_encoder.py:61-142) -> PointNet2.forward
 * (models/pointcloud/pointnet2.py:80-100) -> DynamicEdgeConv + global_max_pool + lin + F.normalize.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct t2p_cell_weights {
    /* PointNet2.sa{1,2,3}.point_conv.local_nn (models/pointcloud/pointnet2.py:57-59).  Layer 1 is applied to
     * [x_j | pos_j - pos_i]; sa_w1[l] is [Kpad_l][H_l] with rows = [x features (3,64,128) | pos (3) | zero pad]
     * (Kpad = 6, 96, 160; H = 32, 128, 256); sa_w2[l] is [H_l][C_l] (C = 64, 128, 256). */
    const float* sa_w1[3];
    const float* sa_b1[3];
    const float* sa_w2[3];
    const float* sa_b2[3];
    /* PointNet2.ga.mlp (pointnet2.py:60): [288][512] (rows = [x 256 | pos 3 | pad 29]) and [512][1024] */
    const float* ga_w1;
    const float* ga_b1;
    const float* ga_w2;
    const float* ga_b2;
    /* PointNet2.lin1 / lin2 (pointnet2.py:62-63,89-90): [1024][512], [512][256] */
    const float* lin1_w;
    const float* lin1_b;
    const float* lin2_w;
    const float* lin2_b;
    /* ObjectEncoder.mlp_pointnet (object_encoder.py:53-58,98): [dim_f][D], dim_f = 1024/512/256 */
    const float* pn_w;
    const float* pn_b;
    /* ObjectEncoder.color_encoder / pos_encoder (object_encoder.py:40-41): [3][64], [64][D] */
    const float* col_w1;
    const float* col_b1;
    const float* col_w2;
    const float* col_b2;
    const float* pos_w1;
    const float* pos_b1;
    const float* pos_w2;
    const float* pos_b2;
    /* ObjectEncoder.mlp_merge (object_encoder.py:59,137-138): [n_features*D][D]; unused when n_features == 1 */
    const float* merge_w;
    const float* merge_b;
    /* CellRetrievalNetwork.graph1.nn (cell_retrieval.py:46-48).  Layer 1 on [x_i | x_j - x_i] is split as
     * P_i + Q_j: g_wp = (W1a - W1b)^T [D][D] with bias g_bp, g_wq = W1b^T [D][D]; layer 2 g_w2 [D][D], g_b2. */
    const float* g_wp;
    const float* g_bp;
    const float* g_wq;
    const float* g_w2;
    const float* g_b2;
    /* CellRetrievalNetwork.lin (cell_retrieval.py:49): [D][D] x 2 */
    const float* lin_w1;
    const float* lin_b1;
    const float* lin_w2;
    const float* lin_b2;
    /* Optional "f16x3" split-precision images of the MFMA-heavy layer-2 weights (used when cfg->precision == 1):
     * every weight is split w = hi + lo/2048 with hi, lo in fp16, stored in MFMA B-operand register order
     * uint16 [2 planes hi,lo][N/32 column tiles][K/16 steps][2 lane halves][32 lanes][8]  holding
     * w[k = half*K/2 + 8*step + e][n = 32*tile + lane]  (packing.py::pack_f16x3). */
    const void* sa_w2_x3[3];
    const void* ga_w2_x3;
    /* The SA layer-2 images (sa_w2_x3) use the scaled single-accumulator form instead: w' = s w with s = sa_w2_scale[l]
     * a power of two (so that |w'| uses fp16's range), hi = fp16(w'), lo = fp16(w' - hi) (no 2048 factor: MFMA honours
     * fp16 denormals), the three products hi.hi + hi.lo + lo.hi share ONE fp32 accumulator that starts at
     * sa_b2_x3[l] = s b2; the kernel divides by s when it drains an object (packing.py::pack_f16x3_scaled). */
    const float* sa_b2_x3[3];
    float sa_w2_scale[3];
    /* Scaled split images of the dense layers behind the trunk, for the LDS-tiled f16x3 GEMM
     * (packing.py::pack_gemm_x3): uint16 [2 planes hi,lo][N][K rounded up to 32] holding s w[k][n], lo = s w - hi. */
    const void* lin1_x3;
    const void* lin2_x3;
    const void* merge_x3;
    const void* pn_x3;   /* mlp_pointnet */
    const void* g_wp_x3; /* DynamicEdgeConv layer-1 tables P, Q */
    const void* g_wq_x3;
    float lin1_scale, lin2_scale, merge_scale, pn_scale, g_wp_scale, g_wq_scale;
    /* pack_f16x3 images of the layer-1 matrices that read SA output rows: of their first C + 16 rows only
     * (K = 80 / 144 for sa_w1[1], sa_w1[2]; 272 for ga_w1: [features C | xyz | zero rows]; the fp32 matrices keep C + 32
     * rows).  Levels 1 and 2 only (level 0 has K = 6 and runs on the VALU); [0] is ignored */
    const void* sa_w1_x3[3];
    const void* ga_w1_x3;
    /* ObjectEncoder.class_embedding / color_embedding (object_encoder.py:31-38), used by the --class_embed /
     * --color_embed ablations only: [n_classes + 1][D], [8][D] */
    const float* class_embedding;
    const float* color_embedding;
    /* fp16-range guard (cfg->overflow_flag): bound of GA layer 1's output from its input's magnitude,
     * |gh| <= ga_w1_l1 * max(|F_3|, 1) + ga_b1_absmax with ga_w1_l1 = max over output columns of sum_k |ga_w1[k][col]|
     * and ga_b1_absmax = max |ga_b1| (that kernel's epilogue is too tight for a running maximum of its own). */
    float ga_w1_l1;
    float ga_b1_absmax;
    /* the same kind of bound for the layer-1 tables that depend on the inputs only: sa_wp_l1[l] = max over columns of
     * |W1p_l[0][c]| + |W1p_l[1][c]| + |W1p_l[2][c]| (position rows of SA level l's first layer: |B_l| <= sa_wp_l1[l] max|xyz|),
     * sa_a1_l1 = max over columns of sum_k |sa_w1[0][k][c]| and sa_b1_absmax = max |sa_b1[0]| (|A_1| <= sa_a1_l1 max|input|
     * + sa_b1_absmax). */
    float sa_wp_l1[3];
    float sa_a1_l1;
    float sa_b1_absmax;
    /* optional f16x3 image (packing.py::pack_f16x3 layout, as ga_w2_x3) of g_w2, DynamicEdgeConv's second layer */
    const void* g_w2_x3;
} t2p_cell_weights;

typedef struct t2p_cell_config {
    int32_t n_pts;             /* points per object after T.FixedPoints (training/args.py:53); 256 */
    int32_t embed_dim;         /* D; 256 */
    int32_t pointnet_features; /* 0 | 1 | 2 (training/args.py:58) */
    int32_t use_class;         /* "class" / "color" / "position" in args.use_features (training/args.py:21) */
    int32_t use_color;
    int32_t use_position;
    int32_t self_loops;        /* 1 = PyG PointConv(add_self_loops=True) semantics (upstream default), 0 = none */
    int32_t knn_k;             /* DynamicEdgeConv k (cell_retrieval.py:47); 8 */
    int32_t variation;         /* args.variation (cell_retrieval.py:45-54): 0 = max, 1 = mean aggregation + mean pool */
    float radius[3];           /* SA ball radii (pointnet2.py:57-59); 0.2, 0.3, 0.4 */
    int32_t chunk_objects;     /* objects processed per internal chunk (whole cells); 0 = T2P_DEFAULT_CHUNK_OBJECTS */
    int32_t precision;         /* 0 = fp32 MFMA (exact fp32 fma chains); 1 = f16x3 split-precision MFMA with fp32
                                  accumulation (hi.hi + hi.lo + lo.hi), same 1e-4 parity bar (for every cell whose DynamicEdgeConv
                                  kNN graph has no near-tie: DESIGN.md section 2), 5.3x the MFMA rate */
    /* ground-truth embedding ablations (training/args.py:60-61, object_encoder.py:74-84,103-120): when class_embed
     * is set the PointNet++ is skipped and the "class" feature is F.normalize(class_embedding[class_idx]); when
     * color_embed is set the "color" feature is F.normalize(color_embedding[color_idx]).  Index arrays are DEVICE
     * int32 [n_obj] (NULL when the flag is 0). */
    int32_t class_embed;
    int32_t color_embed;
    const int32_t* class_idx;
    const int32_t* color_idx;
    /* 1 = stop after ObjectEncoder.forward (models/object_encoder.py:61-142): no cell head; `out` is ignored (may be
     * NULL), the [n_obj][D] result is returned through trace->obj_emb (required).  This is the entry the fine stage
     * uses (SuperGlueMatch.forward, models/superglue_matcher.py:101-103); embed_dim may then be any multiple of 64
     * up to 512 (the fine stage trains with 128, README.md:62). */
    int32_t objects_only;
    /* fp16-range guard of the f16x3 path (precision == 1): the split-precision kernels convert fp32 activations to fp16
     * (round to nearest: a magnitude past 65504 would become inf).  When non-NULL, this DEVICE word receives a
     * sticky OR of a non-zero code whenever a conversion site of the call may have left fp16's range (bits 0-2: SA level
     * 1-3 edge inputs, judged by max|A_l| + max|B_l|; bit 3: SA output rows split by the dense table kernels; bit 4: GA
     * hidden planes, judged by a norm bound; bit 5: rows of the LDS-tiled GEMMs; bit 6: a NaN among the input points / colours -
     * the float-max aggregation of the f16x3 kernels would drop it where the reference's scatter-max propagates it; bit 7: LOW side - the
     * largest hidden activation or the largest output of an SA level is below 2^-7, where the fp16 pieces keep an absolute 2^-25
     * instead of a relative 2^-22 and a later BatchNorm that rescales would expose the loss).  The tests are conservative: they may
     * fire for a checkpoint that would just have fitted, never the other way round.  The caller clears it, reads it after the stream has drained, and must
     * not trust the call's output when it is set (the Python host raises or re-runs with precision = 0).  NULL: no check. */
    int32_t* overflow_flag;
    /* A/B switch between equivalent execution plans (0 = the default plan):
     *   bit 0: keep the edge rows of repeated points in SA level 1's row lists (default: t2p_dedup_rows drops them; every
     *          output bit is the same either way)
     * Bits outside T2P_TUNING_MASK are refused (T2P_E_ARG).  (Rounds 1-3 kept alternative SA kernels behind further bits; the
     * measured record is docs/notebook.md, the code is in the git history.) */
    int32_t tuning;
} t2p_cell_config;

/* Optional stage outputs for parity tests (any member may be NULL).  Layouts:
 *   fps_idx[l] uint8 [n_obj][n_cent_l]      local FPS indices into level l's dense ordering
 *   nbr[l]     uint8 [n_obj][n_cent_l][32]  ball-query neighbours (first cnt valid), cnt[l] uint8 [n_obj][n_cent_l]
 *   sa_out[l]  fp32  [n_obj*n_cent_l][C_l+32] rows = [features C_l | centroid xyz | 0 | 28 columns the call leaves untouched]
 *   features0  fp32  [n_obj][1024]; features1 [n_obj][512]; features2 [n_obj][256]; obj_emb [n_obj][D] (ObjectEncoder output)
 *   knn_idx    int32 [n_obj][knn_k] global object rows (-1 = none) */
typedef struct t2p_cell_trace {
    uint8_t* fps_idx[3];
    uint8_t* nbr[3];
    uint8_t* cnt[3];
    float* sa_out[3];
    float* features0;
    float* features2;
    float* obj_emb;
    int32_t* knn_idx;
    float* features1;
} t2p_cell_trace;

size_t t2p_encode_cells_workspace_bytes(int64_t n_obj, int64_t n_cells, const t2p_cell_config* cfg);

/* xyz, rgb [n_obj][n_pts][3] (PyG batch .pos / .x of dataloading/kitti360pose/utils.py:99-109, objects of all
 * cells concatenated); center, mean_rgb [n_obj][3] (Object3d.get_center / get_color_rgb,
 * datapreparation/kitti360pose/imports.py:28-41); cell_ptr [n_cells+1] int32 CSR over objects, given both in host
 * memory (chunk planning) and in device memory.  out [n_cells][D], L2-normalised rows. */
int t2p_encode_cells(const float* xyz, const float* rgb, const float* center, const float* mean_rgb,
                     const int32_t* cell_ptr_host, const int32_t* cell_ptr, int64_t n_obj, int64_t n_cells,
                     const t2p_cell_weights* w, const t2p_cell_config* cfg, float* out, const t2p_cell_trace* trace,
                     void* workspace, size_t workspace_bytes, t2p_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Input packing on the device: replaces the per-object host work of
 * dataloading/kitti360pose/utils.py:89-110 (Data -> T.FixedPoints -> T.NormalizeScale -> Batch) and the per-object
 * NumPy means of models/object_encoder.py:121-131 (Object3d.get_color_rgb / get_center).
 * raw_xyz, raw_rgb [n_points_total][3] fp32: the raw points of all objects back to back; obj_ptr [n_obj+1] int32 CSR;
 * sample_idx [n_obj][n_pts] int32: per-object LOCAL indices drawn with replacement by the host's seeded generator
 * (T.FixedPoints is random even at evaluation time, evaluation/pipeline.py:290-293 -- the draw stays on the host so that
 * runs are reproducible).  rot_cos_sin: NULL, or [n_obj][2] fp32 (cos, sin) of one angle per object = the training
 * transform's T.RandomRotate(120, axis=2) between FixedPoints and NormalizeScale (training/coarse.py:192-198):
 * pos <- pos @ [[c, s, 0], [-s, c, 0], [0, 0, 1]] (PyG 1.7 - 2.0 LinearTransformation; later releases multiply by the
 * transpose, i.e. the opposite angle -- the same distribution for a symmetric range); the angle is drawn on the host.
 * Outputs are exactly the inputs of t2p_encode_cells.
 * ---------------------------------------------------------------------------------------------------------- */
int t2p_pack_objects(const float* raw_xyz, const float* raw_rgb, const int32_t* obj_ptr, const int32_t* sample_idx,
                     const float* rot_cos_sin, int64_t n_obj, int32_t n_pts, float* xyz, float* rgb, float* center, float* mean_rgb,
                     t2p_stream_t stream);

/* The dataloader of a scene that is RESIDENT in HBM - what evaluation/pipeline.py:282-342 builds per batch on the host
 * (Kitti360CoarseDatasetMulti / Kitti360TopKDataset -> batch_object_points, dataloading/kitti360pose/utils.py:89-110,
 * dataloading/kitti360pose/eval.py:117-189) for every cell of the database and again for every (query, candidate cell)
 * pair of the fine stage.  The raw points of all objects of the scene are uploaded once (raw_xyz, raw_rgb [n_points][3]
 * fp32, obj_ptr [n_scene_obj + 1] int32 CSR) together with the per-object means scene_center / scene_color
 * [n_scene_obj][3] fp32 (Object3d.get_center() / get_color_rgb(): the reference's float64 NumPy means, cast); a call then
 * packs any list of them: output slot s takes scene object obj_id[s] (int32 [n_out]; the same object may fill many slots:
 * a cell retrieved for several queries, the padding object of dataloading/kitti360pose/eval.py:147-149).
 * T.FixedPoints(n_pts) is counter-based: point p of slot s is raw point
 *       ((mix64(key[s] ^ p * 0xD6E8FEB86659FD93) >> 32) * m) >> 32      of the object's m points, with
 *       mix64(x): x += 0x9E3779B97F4A7C15; x = (x ^ x >> 30) * 0xBF58476D1CE4E5B9; x = (x ^ x >> 27) * 0x94D049BB133111EB;
 *                 x ^ x >> 31   (64-bit wrap-around),
 * key [n_out] uint64 chosen by the host (pipeline.PerCellTransform: a hash of (seed, global cell index, slot in the cell),
 * so a cell's sample does not depend on the rank, batch or stream that packs it).  T.NormalizeScale follows, bit for bit
 * as ATen computes it on the host (csrc/small_kernels.hip).  n_pts <= 1024.
 * Outputs: xyz, rgb [n_out][n_pts][3], center, mean_rgb [n_out][3] = the inputs of t2p_encode_cells; rgb, center, mean_rgb
 * and sample_idx_out (int32 [n_out][n_pts]: the draws, for checking) may each be NULL (not written). */
int t2p_pack_scene_objects(const float* raw_xyz, const float* raw_rgb, const int32_t* obj_ptr, const int32_t* obj_id,
                           const uint64_t* key, const float* scene_center, const float* scene_color, int64_t n_out, int32_t n_pts,
                           float* xyz, float* rgb, float* center, float* mean_rgb, int32_t* sample_idx_out, t2p_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Fine stage: hint <-> object matching + offset regression.
 * Replaces SuperGlue.forward models/superglue.py:239-330 (GNN of ["self","cross"] x num_layers AttentionalPropagation
 * layers, final_proj, scores / sqrt(D), log-space optimal transport with dustbin, mutual nearest neighbours + 0.2
 * threshold) and mlp_offsets (models/superglue_matcher.py:116).  Inputs are the L2-normalised object / hint encodings
 * of SuperGlueMatch.forward (:94-103), tokens-major: desc0 [B][n_obj][D], desc1 [B][n_hints][D] fp32.
 * All matrices k-major [in][out], BatchNorm (eval) folded into mlp.0: per GNN layer l
 *   wqkv [D][3D] = attn.proj.{0,1,2} side by side (q | k | v), bqkv [3D];  wm [D][D], bm [D] = attn.merge;
 *   w1 [2D][2D], b1 [2D] = mlp.0 + mlp.1 (BN);  w2 [2D][D], b2 [D] = mlp.3;   cross[l] (HOST array) 0 = self, 1 = cross.
 * Outputs: P [B][n_obj+1][n_hints+1] fp32; matches0 [B][n_obj], matches1 [B][n_hints] int64 (-1 = no match);
 * matching_scores0/1 fp32; offsets [B][n_hints][2] fp32.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct t2p_match_weights {
    int32_t n_layers;
    const int32_t* cross;
    const float *wqkv, *bqkv, *wm, *bm, *w1, *b1, *w2, *b2; /* [n_layers][...] device */
    const float *wf, *bf;                                   /* final_proj [D][D], [D] */
    float bin_score;
    const float *wo1, *bo1, *wo2, *bo2;                     /* mlp_offsets: [D][D/2], [D/2], [D/2][2], [2] */
    /* Optional scaled split images (packing.py::pack_gemm_x3, one [2][N][Kp] image per layer, layers back to back) of
     * the GNN / final_proj matrices for the f16x3 GEMM; NULL = fp32 MFMA.  One power-of-two scale per matrix family. */
    const void *wqkv_x3, *wm_x3, *w1_x3, *w2_x3, *wf_x3;
    float scale_qkv, scale_m, scale_1, scale_2, scale_f;
} t2p_match_weights;

size_t t2p_match_workspace_bytes(int64_t batch, int32_t n_obj, int32_t n_hints, int32_t embed_dim);

int t2p_match(const float* desc0, const float* desc1, int64_t batch, int32_t n_obj, int32_t n_hints, int32_t embed_dim,
              const t2p_match_weights* w, int32_t sinkhorn_iters, float match_threshold, float* P, int64_t* matches0,
              int64_t* matches1, float* mscores0, float* mscores1, float* offsets, void* workspace,
              size_t workspace_bytes, t2p_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Text branch: CellRetrievalNetwork.encode_text (models/cell_retrieval.py:69-75) on token ids produced by the
 * host tokeniser of LanguageEncoder.forward (models/modules.py:60-72).
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct t2p_text_weights {
    const float* embedding; /* word_embedding.weight [V][D] (row 0 = padding/<unk>) */
    const float* w_ih;      /* [2][D][4D] k-major: lstm.weight_ih_l0^T, lstm.weight_ih_l0_reverse^T */
    const float* w_hh;      /* [2][D][4D] k-major: lstm.weight_hh_l0^T, ..._reverse^T */
    const float* bias;      /* [2][4D] = bias_ih + bias_hh (gate order i, f, g, o) */
    /* NULL (exact fp32 MFMA recurrence), or the f16x3 images of w_hh: per direction one packing.py::pack_f16x3_scaled image
     * (as t2p_cell_weights.sa_w2_x3: uint16 [2 planes hi,lo][4D/32 tiles][D/16 steps][2 halves][32 lanes][8] of
     * w' = w_hh_scale * w, w_hh_scale a power of two) - the recurrent product then runs as 3 f16 MFMAs per 16 k with fp32
     * accumulation: fp32-class error, a time step 5x shorter (a call's latency is max_len time steps whatever the batch) */
    const void* w_hh_x3;
    float w_hh_scale;
} t2p_text_weights;

size_t t2p_encode_text_workspace_bytes(int64_t batch, int32_t vocab, int32_t embed_dim);

/* tokens [batch][max_len] int32 right-padded with 0, lengths [batch] int32.  out_raw (nullable) receives the
 * LanguageEncoder output (mean of the two final hidden states), out the L2-normalised rows; both [batch][D]. */
int t2p_encode_text(const int32_t* tokens, const int32_t* lengths, int64_t batch, int32_t max_len, int32_t vocab,
                    int32_t embed_dim, const t2p_text_weights* w, float* out_raw, float* out, void* workspace,
                    size_t workspace_bytes, t2p_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Retrieval: replaces the per-query NumPy loop of training/coarse.py:134-140
 * (float64 `cell_encodings @ text_encodings[q]`, `argsort(-scores)[:k]`).
 * queries [nq][dim], cells [nc][dim] fp32; out_idx [nq][k] int64 (= cell row + index_offset, ties -> lower index,
 * -1 when nc < k), out_score [nq][k] float64.
 * ---------------------------------------------------------------------------------------------------------- */
size_t t2p_sim_topk_workspace_bytes(int64_t nq, int64_t nc, int32_t k);
int t2p_sim_topk(const float* queries, const float* cells, int64_t nq, int64_t nc, int32_t dim, int32_t k,
                 int64_t index_offset, int64_t* out_idx, double* out_score, void* workspace, size_t workspace_bytes,
                 t2p_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Opt-in kernel timing for bench.py: when enabled, every kernel launch of the calls above is bracketed by hipEvents
 * recorded on the launch stream (the only process-global state of the library; not thread-safe, off by default).
 * t2p_profile_report waits for the recorded launches and writes "<kernel> <launches> <total_ms>\n" lines into buf.
 * ---------------------------------------------------------------------------------------------------------- */
void t2p_profile_enable(int on);
int t2p_profile_report(char* buf, size_t buf_bytes);
/* Measurement hook for the per-kernel energy table (profiles/energy_table.py): from now on every launch the report above
 * lists under `scope` (the ten largest kernels of the cell branch have the hook) is issued `reps` times back to back with
 * the same arguments - each of them rewrites its outputs from its inputs, so results do not change - which holds ONE kernel
 * on the chip long enough for a power sampler.  reps <= 1 or scope == NULL: off (the default). */
void t2p_profile_repeat(const char* scope, int32_t reps);

/* ------------------------------------------------------------------------------------------------------------
 * Training-mode text branch (first part): one step of the LSTM recurrence of
 * LanguageEncoder.forward (models/modules.py:77-90: nn.LSTM on a PackedSequence, gates i, f, g, o) with its activations
 * kept, and the matching backward step.  The host loop (text2pos-cvpr2022_amd/modules.py::_LstmTrainFn) alternates them with
 * t2p_gemm for the recurrent products; the caller is training/coarse.py:44 (anchor = model.encode_text(...); loss.backward()).
 *   forward:  pre [B][4D] = h_{s-1} W_hh (k-major product from t2p_gemm); gate_table [V][4D] = E W_ih + b_ih + b_hh;
 *             sequences with step >= length carry (c, h) over and store zero gates; reverse != 0 reads token len-1-step.
 *   backward: dh = dh_gemm (d_pre of step+1 times W_hh^T, NULL for the last step) + dh_carry_in; d_pre [B][4D] is the
 *             gradient of the step's gate pre-activations (zero for finished sequences, whose dh / dc pass through
 *             dh_carry_out / dc_out).
 * ---------------------------------------------------------------------------------------------------------- */
/* The whole recurrence of ONE direction with the time loop inside the library (one call instead of 2 T: the per-step kernels are
 * tiny and the Python loop around them was bound by the host's launch rate).  Layouts as modules._LstmTrainFn keeps them:
 * gates [T][B][4D], cs / hs [T+1][B][D] with slice 0 = the initial state (zeros), w_hh_k [D][4D] = W_hh^T, pre_ws [B][4D] scratch.
 * backward: dh_last [B][D] = the gradient of the final hidden state, w_hh_t [4D][D] = W_hh, d_pre [T][B][4D] (out),
 * ws [5][B][D] scratch.  embed_dim must be a multiple of 4 (else T2P_E_UNSUPPORTED: the host falls back to the per-step calls).
 * The recurrent products run on a few-rows kernel (tg_gemm.hip::k_gemm_skinny): fp32 sums in another order than t2p_gemm's. */
int t2p_lstm_train_forward(const float* gate_table, const float* w_hh_k, const int32_t* tokens, const int32_t* lengths,
                           int64_t batch, int32_t max_len, int32_t embed_dim, int32_t reverse, float* gates, float* cs, float* hs,
                           float* pre_ws, t2p_stream_t stream);
int t2p_lstm_train_backward(const float* dh_last, const float* w_hh_t, const float* gates, const float* cs, const int32_t* lengths,
                            int64_t batch, int32_t max_len, int32_t embed_dim, float* d_pre, float* ws, t2p_stream_t stream);
int t2p_lstm_cell_forward(const float* pre, const float* gate_table, const int32_t* tokens, const int32_t* lengths,
                          int64_t batch, int32_t max_len, int32_t embed_dim, int32_t step, int32_t reverse, const float* c_prev,
                          const float* h_prev, float* gates, float* c, float* h, t2p_stream_t stream);
int t2p_lstm_cell_backward(const float* dh_gemm, const float* dh_carry_in, const float* dc_in, const float* gates,
                           const float* c_prev, const float* c, const int32_t* lengths, int64_t batch, int32_t embed_dim,
                           int32_t step, float* d_pre, float* dc_out, float* dh_carry_out, t2p_stream_t stream);

/* Training-mode building blocks of the cell branch (second part; csrc/train_ops.hip).
 * BatchNorm1d in training mode + optional ReLU over row SEGMENTS (models/modules.py:21-29 in model.train(); the reference
 * runs its PointNet++ once per cell, models/object_encoder.py:92-95, so those layers take their statistics per cell):
 * x, y [M][C] fp32; seg_ptr [n_seg+1] int32 (device) tiles the rows; mean / invstd / var_unbiased [n_seg][C] (biased
 * variance normalises, the unbiased one feeds the running estimate; float64 accumulation in a fixed order: row chunks of a
 * segment are reduced by separate workgroups into `workspace` - t2p_bn_train_workspace_bytes - and combined in order).
 * backward: dx [M][C]; dgamma_seg / dbeta_seg [n_seg + 1][C]: per segment, and in row n_seg their sum over the segments (the
 * layer's weight / bias gradient, added in float64 in segment order); it takes x
 * and the layer's beta [C], not y: the ReLU mask (y > 0) is recomputed from x exactly as the forward formed y.
 * Segment max (PointConv aggr="max", gnn.global_max_pool, DynamicEdgeConv aggr="max" over rows sorted by destination):
 * out [n_seg][C], arg [n_seg][C] = winning row (first one on ties, -1 and out = 0 for an empty segment); the backward
 * routes dout to the winning rows. */
size_t t2p_bn_train_workspace_bytes(int64_t rows, int32_t n_seg, int32_t channels);
int t2p_bn_relu_train_forward(const float* x, const int32_t* seg_ptr, int32_t n_seg, int64_t rows, int32_t channels,
                              const float* gamma, const float* beta, float eps, int32_t relu, float* y, float* mean,
                              float* invstd, float* var_unbiased, void* workspace, size_t workspace_bytes,
                              t2p_stream_t stream);
int t2p_bn_relu_train_backward(const float* dy, const float* x, const float* beta, const int32_t* seg_ptr, int32_t n_seg,
                               int64_t rows, int32_t channels, const float* mean, const float* invstd, const float* gamma,
                               int32_t relu, float* dx, float* dgamma_seg, float* dbeta_seg, void* workspace,
                               size_t workspace_bytes, t2p_stream_t stream);
int t2p_segment_max_forward(const float* x, const int32_t* seg_ptr, int32_t n_seg, int32_t channels, float* out, int32_t* arg,
                            t2p_stream_t stream);
int t2p_segment_max_backward(const float* dout, const int32_t* arg, const int32_t* seg_ptr, int32_t n_seg, int32_t channels,
                             float* dx, t2p_stream_t stream);
/* segment mean (variation 1: DynamicEdgeConv aggr="mean", gnn.global_mean_pool; models/cell_retrieval.py:50-54, :100-103) */
int t2p_segment_mean_forward(const float* x, const int32_t* seg_ptr, int32_t n_seg, int32_t channels, float* out,
                             t2p_stream_t stream);
int t2p_segment_mean_backward(const float* dout, const int32_t* seg_ptr, int32_t n_seg, int32_t channels, float* dx,
                              t2p_stream_t stream);

/* Message inputs of the two graph operators and F.normalize, with their backward (training mode):
 *   edge features  out [E][width] = [x[src] | pos[src] - pos_c[dst] | 0 ..]   (PointConv, models/pointcloud/pointnet2.py:31-35;
 *                  width >= C + 3 is the row pitch of out / d_out: a multiple of 8 spares the Linear behind it a padded copy);
 *                  backward: dx [rows of x][C] += d_out[:, :C] at src (dx zeroed by the caller; float atomics)
 *   pair features  out [E][2D] = [x[tgt] | x[src] - x[tgt]]          (DynamicEdgeConv, models/cell_retrieval.py:46-48)
 *   rownorm backward: gradient of t2p_rownorm (F.normalize, eps 1e-12) */
int t2p_edge_features_forward(const float* x, const float* pos, const float* pos_c, const int32_t* src, const int32_t* dst,
                              int64_t n_edges, int32_t channels, int32_t width, float* out, t2p_stream_t stream);
int t2p_edge_features_backward(const float* d_out, const int32_t* src, int64_t n_edges, int32_t channels, int32_t width, float* dx,
                               t2p_stream_t stream);
int t2p_pair_features_forward(const float* x, const int32_t* tgt, const int32_t* src, int64_t n_edges, int32_t dim, float* out,
                              t2p_stream_t stream);
int t2p_pair_features_backward(const float* d_out, const int32_t* tgt, const int32_t* src, int64_t n_edges, int32_t dim,
                               float* dx, t2p_stream_t stream);
int t2p_rownorm_backward(const float* x, const float* dy, int64_t n_rows, int32_t dim, float* dx, t2p_stream_t stream);

/* PairwiseRankingLoss (training/losses.py:126-164, margin training/args.py:46) on the score matrix of the L2-normalised
 * anchor / positive embeddings, scores [B][B] = im_n s_n^T:  row_loss [B] (loss = sum(row_loss) / B),
 * d_scores [B][B] = dLoss / dScores, row_count [B] scratch.  Deterministic (fixed-order reductions). */
int t2p_pairwise_ranking(const float* scores, int32_t batch, float margin, float* row_loss, float* d_scores, float* row_count,
                         t2p_stream_t stream);

/* HardestRankingLoss (training/losses.py:167-201, --ranking_loss hardest) on the same score matrix: best [2B] = the largest
 * hinge of every row (first B) and column (last B), where [2B] = its position (-1 if none is positive),
 * d_scores [B][B] = dLoss / dScores; loss = (sum best[:B] + sum best[B:]) / B. */
int t2p_hardest_ranking(const float* scores, int32_t batch, float margin, float* best, int32_t* where, float* d_scores,
                        t2p_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Stage-level exports (used by the stage-wise parity tests).
 * ---------------------------------------------------------------------------------------------------------- */
/* Fused gnn.fps + gnn.radius of the three SA levels (pointnet2.py:26-30); outputs as in t2p_cell_trace.
 * n_cent_l = ceil(n_dense_l / 2), n_dense_0 = n_pts. */
int t2p_sample_group(const float* xyz, int64_t n_obj, int32_t n_pts, const float* radius_host /*[3]*/,
                     uint8_t* const* fps_idx /*[3]*/, uint8_t* const* nbr /*[3]*/, uint8_t* const* cnt /*[3]*/,
                     t2p_stream_t stream);
/* The same kernel with its compact edge-row lists as output (what the SA edge kernels of t2p_encode_cells consume):
 * rows[l] uint16 [n_obj][n_cent_l * 33], per object sorted by centroid: (centroid | 0x80 for the self-loop row) << 8 | source,
 * hits in ascending source order (<= 32), then - with self_loops - the centroid's self-loop row (source byte = centroid);
 * n_rows[l] uint16 [n_obj].  t2p_edge_counts / t2p_edge_expand turn a level's lists into the edge arrays of
 * gnn.PointConv(...)(x, (pos, pos[idx]), edge_index) with torch_geometric's self-loop rewrite (pointnet2.py:26-35) for the
 * training-mode path: counts [n_obj * n_cent] kept rows per centroid (a hit whose two CELL-local indices agree is dropped: the
 * appended loop (i, i) replaces it; first_obj [n_obj] = first object of the object's cell); with cent_ptr = exclusive prefix of
 * counts (n_obj * n_cent + 1 entries), src / dst [cent_ptr[last]] int32 = (dense row, centroid row) of every edge, sorted by dst. */
int t2p_group_rows(const float* xyz, int64_t n_obj, int32_t n_pts, const float* radius_host /*[3]*/, int32_t self_loops,
                   uint8_t* const* fps_idx /*[3]*/, uint16_t* const* rows /*[3]*/, uint16_t* const* n_rows /*[3]*/,
                   t2p_stream_t stream);
int t2p_edge_counts(const uint16_t* rows, const uint16_t* n_rows, const int32_t* first_obj, int64_t n_obj, int32_t n_dense,
                    int32_t n_cent, int32_t self_loops, int32_t* counts, t2p_stream_t stream);
int t2p_edge_expand(const uint16_t* rows, const uint16_t* n_rows, const int32_t* first_obj, const int32_t* cent_ptr, int64_t n_obj,
                    int32_t n_dense, int32_t n_cent, int32_t self_loops, int32_t* src, int32_t* dst, t2p_stream_t stream);
/* Level-1 (SA1) row lists without the edges of repeated points.  rows uint16 [n_obj][(n_pts/2) * 33]: per object the compact
 * edge-row list of k_sample_group, sorted by centroid: (centroid | 0x80 for the self-loop row) << 8 | source point, terminated
 * by four 0xFFFF; n_rows uint16 [n_obj].  A row whose source point repeats an EARLIER point of the object bit for bit (xyz and
 * rgb; T.FixedPoints draws with replacement, dataloading/kitti360pose/utils.py:99-109) carries the same message as that
 * point's row for the same centroid, so under PointConv's max-aggregation (pointnet2.py:31-35) it can be dropped.  In place;
 * detection is conservative (a repeat may survive, a non-repeat is never dropped).  n_pts must be 256. */
int t2p_dedup_rows(const float* xyz, const float* rgb, int64_t n_obj, int32_t n_pts, uint16_t* rows, uint16_t* n_rows,
                   t2p_stream_t stream);
/* knn of DynamicEdgeConv (cell_retrieval.py:46-48): x [n][dim], seg_ptr [n_seg+1] int32, out [n][k] int32 */
int t2p_knn(const float* x, int32_t dim, const int32_t* seg_ptr, int32_t n_seg, int32_t max_seg_rows, int32_t k,
            int32_t* out_idx, t2p_stream_t stream);
/* C[M][ldc] (+c0) = act(A[M][lda] W[K][N] + bias[N]);  K % 4 == 0, N % 8 == 0, lda % 4 == 0 */
int t2p_gemm(const float* a, int32_t lda, const float* w, const float* bias, float* c, int32_t ldc, int32_t c0,
             int64_t m, int32_t k, int32_t n, int32_t relu, t2p_stream_t stream);
/* C[k1][n] (ldc) = A[m][k1]^T B[m][n]: a product whose reduction runs over the ROWS - the weight gradient dW = dY^T X of
 * every nn.Linear, the recurrent / input weight gradients of the LSTM and its gate-table gradient in the training-mode path
 * (training/coarse.py:31-62 through autograd).  The rows are split over the grid and the partial products added in a fixed
 * order (deterministic); fp32 MFMA.  workspace: t2p_gemm_tn_workspace_bytes. */
size_t t2p_gemm_tn_workspace_bytes(int64_t m, int32_t k1, int32_t n);
int t2p_gemm_tn(const float* a, int32_t lda, const float* b, int32_t ldb, float* c, int32_t ldc, int64_t m, int32_t k1, int32_t n,
                void* workspace, size_t workspace_bytes, t2p_stream_t stream);
/* Weight and bias gradient of an nn.Linear in the training-mode path (every `get_mlp` block under model.train(),
 * models/modules.py:11-36), exact fp32 MFMA (csrc/train_gemm.hip):
 *   dW[K1][N] = dY[M][K1]^T X[M][N] and colsum[K1] = column sums of dY (the bias gradient; may be null): every operand row is
 *   read once per row range, the row ranges' partial blocks are added in a fixed order (deterministic).
 *   lda, ldb multiples of 4, operands 16-byte aligned.  workspace: t2p_linear_wgrad_workspace_bytes. */
size_t t2p_linear_wgrad_workspace_bytes(int64_t m, int32_t k1, int32_t n);
int t2p_linear_wgrad_f32(const float* dy, int32_t lda, const float* x, int32_t ldb, float* dw, int32_t ldc, float* colsum, int64_t m,
                         int32_t k1, int32_t n, void* workspace, size_t workspace_bytes, t2p_stream_t stream);
/* F.normalize(x, dim=-1), eps 1e-12 */
int t2p_rownorm(const float* x, int64_t n_rows, int32_t dim, float* out, t2p_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* T2P_H */

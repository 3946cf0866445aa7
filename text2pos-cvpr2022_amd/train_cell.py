"""Training-mode forward of the cell branch with a backward pass (second part): what
`positive = model.encode_objects(...)` computes under `model.train()` in training/coarse.py:32-58.

Same graph as the inference kernels, but every BatchNorm1d normalises with the statistics of the current rows
(models/modules.py:21-29 in train mode) - per CELL inside the PointNet++, which the reference runs once per cell
(models/object_encoder.py:92-95), over the whole batch elsewhere - so nothing is folded.  Layer by layer:
  * index-producing stages (FPS, ball query, kNN) are the inference kernels (t2p_sample_group, t2p_knn): no gradient
    flows through them in the reference either;
  * the arithmetic runs on the HIP building blocks of train_ops.py (Linear on the tiled GEMM, batch-statistics
    BatchNorm + ReLU per segment, segment max with arg-max routing);
  * the message inputs of the two graph operators and F.normalize are HIP kernels with backward kernels as well
    (t2p_edge_features_*, t2p_pair_features_*, t2p_rownorm / t2p_rownorm_backward); what is left to torch tensor ops is
    a few gathers and row-pointer products, the concatenation of the three object feature parts and the ReLU behind the two
    plain Linear heads; PointConv's edge lists come from ops.group_edges (device-side), everything the cell sizes alone
    determine from one host plan (_host_plan).
The PointConv self-loop rewrite of torch_geometric (remove edges whose two CELL-local indices agree, append (i, i) for
every centroid row i of the cell: dense row i of the cell feeds centroid row i) is reproduced on the edge lists exactly as
the inference path encodes it in its row tables."""
import numpy as np
import torch

from . import ops
from . import train_ops as TO


def _i32(t):
    return t.to(torch.int32).contiguous()


def _mlp_train(x, mlp, seg_ptr, rows_min):
    """get_mlp block list (Linear, BatchNorm1d, ReLU) in training mode; statistics per row segment.  rows_min: the
    smallest segment's row count (known on the host); a single-row segment raises as nn.BatchNorm1d does."""
    for blk in mlp:
        x = TO.bn_relu_train(TO.linear(x, blk[0]), seg_ptr, blk[1], relu=True, rows_min=rows_min)
    return x


def _host_plan(cp: np.ndarray, k: int, dev):
    """Everything the step's integer plumbing can know from the cell sizes alone, computed on the host and uploaded in ONE
    (pinned, asynchronous) copy: a tensor created from host data in the middle of the step is a blocking copy queued behind every
    kernel launched so far - sixteen of those per step kept the host waiting for the GPU.
    first_obj [n_obj], cell_ptr [B + 1]; the kNN edges of DynamicEdgeConv: knn_ptr [n_obj + 1] (min(k, cell size) edges per
    object), knn_tgt [E] (target object of every edge), knn_slot [E] (its position in the flattened [n_obj, k] neighbour table);
    seg[n] = the one-segment row pointer [0, n] for n in {n_obj, n_cells, E}.  All int32."""
    sizes = cp[1:] - cp[:-1]
    n_obj, n_cells = int(cp[-1]), int(sizes.shape[0])
    first = np.repeat(cp[:-1], sizes)
    kk = np.minimum(k, np.repeat(sizes, sizes))
    knn_ptr = np.zeros(n_obj + 1, dtype=np.int64)
    np.cumsum(kk, out=knn_ptr[1:])
    e = int(knn_ptr[-1])
    tgt = np.repeat(np.arange(n_obj), kk)
    slot = np.arange(e) - np.repeat(knn_ptr[:-1], kk) + tgt * k
    segs = sorted({n_obj, n_cells, e})
    parts = [first, cp, knn_ptr, tgt, slot] + [np.array([0, n]) for n in segs]
    flat = np.concatenate(parts).astype(np.int32)
    host = torch.from_numpy(flat)
    if dev.type == "cuda":
        host = host.pin_memory()
    d = host.to(dev, non_blocking=True)
    views, a = [], 0
    for p_ in parts:
        views.append(d[a: a + p_.shape[0]])
        a += p_.shape[0]
    return dict(first_obj=views[0], cell_ptr=views[1], knn_ptr=views[2], knn_tgt=views[3], knn_slot=views[4],
                seg={n: v for n, v in zip(segs, views[5:])}, _host=host)   # (the pinned buffer outlives the copy with the plan)


def encode_objects_train(model, xyz, rgb, center, mean_rgb, cell_ptr, class_idx=None, color_idx=None):
    """model: CellRetrievalNetwork in train(); packed device inputs as encode_objects_packed.  Returns [B, D] unit rows with
    a grad_fn; BatchNorm running estimates are updated as the reference's per-cell / per-batch module calls would."""
    a = model.args
    class_embed, color_embed = bool(getattr(a, "class_embed", False)), bool(getattr(a, "color_embed", False))
    if class_embed != (class_idx is not None) or color_embed != (color_idx is not None):
        raise RuntimeError("args.class_embed / args.color_embed need the per-object class / colour indices")
    dev = xyz.device
    oe, pn = model.object_encoder, model.object_encoder.pointnet
    n_obj, n_pts = xyz.shape[0], xyz.shape[1]
    cp = np.ascontiguousarray(np.asarray(cell_ptr), dtype=np.int64)
    n_cells = cp.shape[0] - 1
    k = model.graph1.k
    plan = _host_plan(cp, k, dev)
    first_obj, cell_ptr_dev = plan["first_obj"], plan["cell_ptr"]
    one = lambda n: plan["seg"][n]

    parts = []

    def pointnet_branch():
        """models/object_encoder.py:86-98: the PointNet++ (one call per cell) + mlp_pointnet."""
        # FPS + ball query + torch_geometric's self-loop rewrite as edge lists, built on the device (ops.group_edges; the torch
        # formulation of the same lists - hit mask, nonzero, remove / append self loops, stable sort - is the statement
        # tests/test_gpu_parity.py::test_group_edges_equal_the_tensor_formulation compares it against)
        levels = ops.group_edges(xyz.contiguous(), first_obj, pn.radii, model.add_self_loops)
        pos = xyz.reshape(n_obj * n_pts, 3)
        x = rgb.reshape(n_obj * n_pts, 3)
        if "color" not in a.use_features:                       # models/object_encoder.py:87-90
            x = torch.zeros_like(x)
        nd = n_pts
        for lvl, sa in enumerate((pn.sa1, pn.sa2, pn.sa3)):
            nc = (nd + 1) // 2
            g = levels[lvl]
            fps = g["fps_idx"].long()
            pos_c = pos.view(n_obj, nd, 3).gather(1, fps[:, :, None].expand(-1, -1, 3)).reshape(n_obj * nc, 3)
            cent_ptr = g["cent_ptr"]
            cell_edge_ptr = cent_ptr[cell_ptr_dev.long() * nc].contiguous()      # edges per cell = BatchNorm's row segments
            msg = TO.edge_features(x, pos, pos_c, g["src"], g["dst"])
            h = _mlp_train(msg, sa.point_conv.local_nn, cell_edge_ptr, nc)    # >= one self loop / hit per centroid
            x, pos, nd = TO.segment_max(h, cent_ptr, covers_all_rows=True), pos_c, nc    # (cent_ptr is the CSR over ALL edge rows)
        h = _mlp_train(torch.cat([x, pos], dim=1), pn.ga.mlp, _i32(cell_ptr_dev.long() * nd), nd)
        f0 = TO.segment_max(h, _i32(torch.arange(n_obj + 1, device=dev) * nd), covers_all_rows=True)
        f1 = torch.relu(TO.linear(f0, pn.lin1))
        f2 = torch.relu(TO.linear(f1, pn.lin2))
        feats = (f0, f1, f2)[a.pointnet_features]
        return _mlp_train(feats, oe.mlp_pointnet, one(n_obj), n_obj)

    if "class" in a.use_features and class_embed:               # models/object_encoder.py:103-109: no PointNet++ at all
        parts.append(TO.normalize(oe.class_embedding(class_idx.long())))
    elif "class" in a.use_features:
        parts.append(TO.normalize(pointnet_branch()))
    elif not class_embed:
        # the reference runs the PointNet++ whenever class_embed is off (models/object_encoder.py:86-98) and only then
        # decides whether the result is a feature: in train() mode its BatchNorm running statistics still move
        with torch.no_grad():
            pointnet_branch()
    if "color" in a.use_features and color_embed:               # models/object_encoder.py:112-120
        parts.append(TO.normalize(oe.color_embedding(color_idx.long())))
    elif "color" in a.use_features:
        parts.append(TO.normalize(_mlp_train(mean_rgb.float(), oe.color_encoder, one(n_obj), n_obj)))
    if "position" in a.use_features:
        parts.append(TO.normalize(_mlp_train(center.float(), oe.pos_encoder, one(n_obj), n_obj)))
    emb = _mlp_train(torch.cat(parts, dim=-1), oe.mlp_merge, one(n_obj), n_obj) if len(parts) > 1 else parts[0]
    emb = TO.normalize(emb)

    # DynamicEdgeConv(k = 8, max) inside each cell (models/cell_retrieval.py:46-48, :97), pool, lin, normalize (:98-106).
    # t2p_knn lists an object's min(k, cell size) neighbours first and pads with -1, so which of its slots are edges is known from
    # the cell sizes: the edge arrays are gathers through host-built index lists (no mask, no nonzero, no size read-back)
    knn = ops.knn(emb.detach().contiguous(), cell_ptr_dev, k, max_seg_rows=int((cp[1:] - cp[:-1]).max()))
    tgt = plan["knn_tgt"]
    srcn = knn.reshape(-1)[plan["knn_slot"].long()].contiguous()
    msg = TO.pair_features(emb, tgt, srcn)
    e_knn = int(tgt.numel())
    h = _mlp_train(msg, model.graph1.nn, one(e_knn), e_knn)
    pool = TO.segment_max if model.variation == 0 else TO.segment_mean    # models/cell_retrieval.py:46-54, :98-103
    xg = pool(h, plan["knn_ptr"])
    xc = pool(xg, cell_ptr_dev)
    return TO.normalize(_mlp_train(xc, model.lin, one(n_cells), n_cells))

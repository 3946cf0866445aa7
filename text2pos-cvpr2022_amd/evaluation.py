"""Retrieval / localisation metrics of the coarse stage, vectorised over all queries (host NumPy; a few KB of data).

Mirrors the bookkeeping of the reference's evaluation loop:
  * hit@k and close-by@k                      training/coarse.py:142-163
  * calc_sample_accuracies (recall@k within thresholds, cross-scene masking, world-coordinate prediction)
                                              evaluation/utils.py:31-54, driven from evaluation/pipeline.py:122-137
  * print_accuracies table                    evaluation/utils.py:57-69
  * run_fine (query x top-k cells through the fine model, pose from matches + offsets, three accuracy tables)
                                              evaluation/pipeline.py:172-279, dataloading/kitti360pose/eval.py:117-189
The top-k indices come from `retrieve_topk` (csrc/sim_topk.hip), so `eval_retrieval` is the drop-in for the part of
`eval_epoch` that follows the two encoding loops.
"""
from copy import copy
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .retrieval import retrieve_topk


def retrieval_accuracies(top_idx: np.ndarray, db_cell_ids: Sequence[str], query_cell_ids: Sequence[str],
                         query_poses_w: np.ndarray, cell_centers_xy: np.ndarray, cell_size: float, top_k: Sequence[int]):
    """top_idx [Nq, max(top_k)] cell rows per query (best first).
    Returns (accuracies {k: float}, accuracies_close {k: float}, top_retrievals {query_idx: cell ids})."""
    top_idx = np.asarray(top_idx)
    db_cell_ids, query_cell_ids = np.asarray(db_cell_ids), np.asarray(query_cell_ids)
    assert top_idx.shape[1] >= max(top_k)
    retrieved_ids = db_cell_ids[top_idx]                                     # [Nq, kmax]
    hits = retrieved_ids == query_cell_ids[:, None]
    dists = np.linalg.norm(np.asarray(query_poses_w)[:, None, 0:2] - np.asarray(cell_centers_xy)[top_idx], axis=2)
    close = dists <= cell_size / 2
    accuracies = {k: float(np.mean(hits[:, :k].any(axis=1))) for k in top_k}
    accuracies_close = {k: float(np.mean(close[:, :k].any(axis=1))) for k in top_k}
    top_retrievals = {q: retrieved_ids[q, : max(top_k)] for q in range(top_idx.shape[0])}
    return accuracies, accuracies_close, top_retrievals


def calc_sample_accuracies(pose, top_cells, pos_in_cells, top_k, threshs):
    """One sample, exactly the reference signature (evaluation/utils.py:31)."""
    return {k: {t: bool(v) for t, v in d.items()} for k, d in
            _sample_hits(np.asarray(pose.pose_w)[None, 0:2], [pose.cell_id.split("_")[0]],
                         np.array([[c.bbox_w[0:2] for c in top_cells]], dtype=np.float64),
                         np.array([[c.cell_size for c in top_cells]], dtype=np.float64),
                         [[c.id.split("_")[0] for c in top_cells]], np.asarray(pos_in_cells)[None], top_k, threshs,
                         reduce=False).items()}


def _sample_hits(pose_xy, pose_scene, bbox_xy, cell_size, cell_scene, pos_in_cells, top_k, threshs, reduce=True):
    assert bbox_xy.shape[1] == max(top_k) == pos_in_cells.shape[1]
    pred_w = bbox_xy + pos_in_cells * cell_size[..., None]                   # world-coordinate prediction per cell
    dists = np.linalg.norm(pose_xy[:, None, :] - pred_w, axis=2)
    dists = np.where(np.asarray(cell_scene) != np.asarray(pose_scene)[:, None], np.inf, dists)
    out = {}
    for k in top_k:
        best = dists[:, :k].min(axis=1)
        out[k] = {t: (float(np.mean(best <= t)) if reduce else best[0] <= t) for t in threshs}
    return out


def localisation_accuracies(poses, retrievals: List[Sequence[str]], cells_dict: Dict[str, object], top_k, threshs,
                            pos_in_cells=None):
    """evaluation/pipeline.py:122-137 over all samples at once: mean recall@k within each threshold, predicting the
    cell centre (pos_in_cells = 0.5) unless offsets are given ([Nq, max(top_k), 2])."""
    nq, kmax = len(retrievals), max(top_k)
    cells = [[cells_dict[cid] for cid in r[:kmax]] for r in retrievals]
    bbox = np.array([[c.bbox_w[0:2] for c in row] for row in cells], dtype=np.float64)
    size = np.array([[c.cell_size for c in row] for row in cells], dtype=np.float64)
    scene = [[c.id.split("_")[0] for c in row] for row in cells]
    if pos_in_cells is None:
        pos_in_cells = 0.5 * np.ones((nq, kmax, 2))
    return _sample_hits(np.array([p.pose_w[0:2] for p in poses], dtype=np.float64),
                        [p.cell_id.split("_")[0] for p in poses], bbox, size, scene, np.asarray(pos_in_cells), top_k, threshs)


def eval_retrieval(cell_encodings, text_encodings, db_cell_ids, query_cell_ids, query_poses_w, cells_dict, cell_size,
                   top_k):
    """The tail of eval_epoch (training/coarse.py:133-167): top-k on the GPU, metrics on the host."""
    idx, _ = retrieve_topk(cell_encodings, text_encodings, int(np.max(top_k)))
    idx = idx.cpu().numpy()
    centers = np.array([cells_dict[cid].get_center()[0:2] for cid in db_cell_ids], dtype=np.float64)
    return retrieval_accuracies(idx, db_cell_ids, query_cell_ids, query_poses_w, centers, cell_size, top_k)


def print_accuracies(accs, name=""):
    """evaluation/utils.py:57-69."""
    if name:
        print(f"\t\t{name}:")
    top_k = list(accs.keys())
    threshs = list(accs[top_k[0]].keys())
    print("", end="")
    for k in top_k:
        print(f"\t\t\t\t{k}", end="")
    print()
    print("/".join([str(t) for t in threshs]) + ":", end="")
    for k in top_k:
        print("\t" + "/".join([f"{accs[k][t]:0.2f}" for t in threshs]), end="")
    print("\n\n", flush=True)


def create_hint_description(pose) -> List[str]:
    """One sentence per description of the pose (dataloading/kitti360pose/base.py:57-66)."""
    return [f"The pose is {d.direction} of a {d.object_color_text} {d.object_label}." for d in pose.descriptions]


def positions_in_cell(center_xy: np.ndarray, matches0: np.ndarray, offsets: np.ndarray) -> np.ndarray:
    """get_pos_in_cell (models/superglue_matcher.py:139-161) for a batch: center_xy [B, n_obj, 2] object centres, matches0
    [B, n_obj] (hint index or -1), offsets [B, n_hints, 2] -> [B, 2]: mean over the matched objects of centre + offset of
    the matched hint; (0.5, 0.5) where nothing matched."""
    m0 = np.asarray(matches0)
    mask = m0 >= 0
    idx = np.broadcast_to(np.clip(m0, 0, None)[:, :, None], m0.shape + (2,))
    pred = np.asarray(center_xy, dtype=np.float64) + np.take_along_axis(np.asarray(offsets, dtype=np.float64), idx, axis=1)
    cnt = mask.sum(axis=1)
    tot = (pred * mask[:, :, None]).sum(axis=1)
    return np.where(cnt[:, None] > 0, tot / np.maximum(cnt, 1)[:, None], 0.5)


def run_fine(model, poses, cells_dict: Dict[str, object], retrievals: List[Sequence[str]], transform, pad_size: int,
             top_k, threshs, queries_per_call: Optional[int] = None, group=None, scene_dev=None):
    """Fine localisation of every query against its max(top_k) retrieved cells (evaluation/pipeline.py:172-279).
    `model(objects, hints, object_points)` is SuperGlueMatch (or anything returning .matches0 [B, pad] / .offsets
    [B, hints, 2]); unlike the reference, which calls the model once per query (10 samples), `queries_per_call` queries
    share a call - the memory knob of this function: one call packs queries_per_call x max(top_k) x pad_size objects.  None = 64
    on the host chain, 256 on the on-device path (256 queries x 10 candidates x 16 objects fill the GPU); a given value is honoured
    on both.  Returns (accuracies_mean, accuracies_offset, accuracies_mean_conf).
    With an initialised torch.distributed process group the queries are split over the ranks in contiguous blocks
    (samples are independent) and the per-query estimates are all-gathered: every rank returns the same tables.
    scene_dev (scene.DeviceScene holding every cell of `retrievals` and >= pad_size padding objects) + a counter-based
    transform (pipeline.PerCellTransform) + a model with forward_packed: the samples are packed on the GPU straight from the
    resident scene (sample q * kmax + c draws what `transform.for_cell(q * kmax + c)` draws on the host), a query's hints are
    encoded once for all its candidates, and the pose estimates are computed for a whole call at once."""
    from . import distributed as TD
    from .data import Object3d, batch_object_points
    from .superglue_matcher import get_pos_in_cell
    kmax = max(top_k)
    assert all(len(r) == kmax for r in retrievals), "retrievals must be trimmed to max(top_k)"
    padded = {}

    def pad(cell_id):  # objects of a cell cut / padded to pad_size (dataloading/kitti360pose/eval.py:141-149)
        if cell_id not in padded:
            objs = list(cells_dict[cell_id].objects)[:pad_size]
            while len(objs) < pad_size:
                objs.append(Object3d.create_padding())
            padded[cell_id] = objs
        return padded[cell_id]

    nq = len(poses)
    rank, world = TD.rank_and_world(group)
    q_lo, q_hi = TD.shard_range(nq, rank, world)
    pos_mean = np.zeros((nq, kmax, 2))
    pos_off = np.zeros((nq, kmax, 2))
    conf = np.zeros((nq, kmax), dtype=np.int64)
    # the on-device path needs everything it calls on the model: a model that only has forward_packed takes the host chain
    on_dev = (scene_dev is not None and hasattr(transform, "keys")
              and all(hasattr(model, a) for a in ("forward_packed", "encode_hints", "args", "object_encoder")))
    if queries_per_call is None:
        queries_per_call = 256 if on_dev else 64
    queries_per_call = max(1, int(queries_per_call))
    if on_dev:
        ids_of_cell = scene_dev.padded_object_ids(pad_size)                                  # [n_cells, pad]
        cell_rows = np.array([[scene_dev.row_of[cid] for cid in r] for r in retrievals], dtype=np.int64).reshape(nq, kmax)
        class_all, color_all = scene_dev.feature_indices(model)
        want_rgb = "color" in model.args.use_features or bool(getattr(model.args, "class_embed", False))
        per_call = queries_per_call
        slot = np.arange(pad_size, dtype=np.int64)
        for q0 in range(q_lo, q_hi, per_call):
            q1 = min(q0 + per_call, q_hi)
            nb = (q1 - q0) * kmax
            ids = ids_of_cell[cell_rows[q0:q1].reshape(-1)]                                  # [nb, pad]
            sample = (np.arange(q0, q1, dtype=np.int64)[:, None] * kmax + np.arange(kmax, dtype=np.int64)[None, :]).reshape(-1)
            keys = transform.keys(sample[:, None], slot[None, :])
            xyz, rgb, center, mean_rgb = scene_dev.pack(ids.reshape(-1), keys.reshape(-1), transform.n_pts, want_rgb=want_rgb)
            if rgb is None:
                rgb = torch.zeros_like(xyz)
            with torch.no_grad():  # evaluation/pipeline.py:171
                hint_enc = model.encode_hints([create_hint_description(poses[q]) for q in range(q0, q1)])
            hint_enc = hint_enc.repeat_interleave(kmax, dim=0)
            ci = co = None
            if class_all is not None or color_all is not None:
                flat_ids = torch.from_numpy(ids.reshape(-1)).to(scene_dev.device)
                ci = None if class_all is None else class_all[flat_ids].contiguous()
                co = None if color_all is None else color_all[flat_ids].contiguous()
            cp = np.arange(nb + 1, dtype=np.int32) * pad_size
            with torch.no_grad():
                out = model.forward_packed(xyz, rgb, center, mean_rgb, cp, hint_enc, ci, co)
            m0, off = out.matches0.cpu().numpy(), out.offsets.cpu().numpy()
            cxy = scene_dev.center64[ids][:, :, 0:2]
            pos_mean[q0:q1] = positions_in_cell(cxy, m0, np.zeros_like(off)).reshape(q1 - q0, kmax, 2)
            pos_off[q0:q1] = positions_in_cell(cxy, m0, off).reshape(q1 - q0, kmax, 2)
            conf[q0:q1] = (m0 >= 0).sum(axis=1).reshape(q1 - q0, kmax)
    for q0 in (range(q_lo, q_hi, queries_per_call) if not on_dev else ()):
        q1 = min(q0 + queries_per_call, q_hi)
        objects, hints, points = [], [], []
        for q in range(q0, q1):
            h = create_hint_description(poses[q])
            for c, cid in enumerate(retrievals[q]):
                objs = pad(cid)
                objects.append(objs)
                hints.append(h)
                tf = transform.for_cell(q * kmax + c) if hasattr(transform, "for_cell") else transform
                points.append(batch_object_points(objs, tf))
        with torch.no_grad():  # evaluation/pipeline.py:171
            out = model(objects, hints, points)
        m0 = np.asarray(out.matches0.detach().cpu() if hasattr(out.matches0, "detach") else out.matches0)
        off = np.asarray(out.offsets.detach().cpu() if hasattr(out.offsets, "detach") else out.offsets)
        for i, objs in enumerate(objects):
            q, c = q0 + i // kmax, i % kmax
            pos_mean[q, c] = get_pos_in_cell(objs, m0[i], np.zeros_like(off[i]))
            pos_off[q, c] = get_pos_in_cell(objs, m0[i], off[i])
            conf[q, c] = int(np.sum(m0[i] >= 0))
    if world > 1:   # every rank gets every query's estimates (3 small gathers; float64 / int64 travel unchanged)
        import torch.distributed as dist
        dev = getattr(model, "device", None) if dist.get_backend(group) == "nccl" else None
        def gather(a):
            # (explicit trailing size: reshape(0, -1) of an empty query block - world larger than nq - is ambiguous to NumPy, and
            # a rank that raised here would leave the others waiting in the collective)
            t = torch.from_numpy(np.ascontiguousarray(a[q_lo:q_hi]).reshape(q_hi - q_lo, int(np.prod(a.shape[1:]))))
            t = TD.all_gather_rows(t.to(dev) if dev is not None else t, nq, group)
            return t.cpu().numpy().reshape(a.shape)
        pos_mean, pos_off, conf = gather(pos_mean), gather(pos_off), gather(conf)
    acc_mean = localisation_accuracies(poses, retrievals, cells_dict, top_k, threshs, pos_mean)
    acc_off = localisation_accuracies(poses, retrievals, cells_dict, top_k, threshs, pos_off)
    # the single most confident candidate (most matched objects; first on ties), evaluated as top-1
    best = np.argmax(conf, axis=1)
    acc_conf = localisation_accuracies(poses, [[r[b]] for r, b in zip(retrievals, best)], cells_dict, [1], threshs,
                                       pos_mean[np.arange(nq), best][:, None, :])
    return acc_mean, acc_off, acc_conf

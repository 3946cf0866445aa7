"""Retrieval / localisation metrics of the coarse stage, vectorised over all queries (host NumPy; a few KB of data).

Mirrors the bookkeeping of the reference's evaluation loop:
  * hit@k and close-by@k                      training/coarse.py:142-163
  * calc_sample_accuracies (recall@k within thresholds, cross-scene masking, world-coordinate prediction)
                                              evaluation/utils.py:31-54, driven from evaluation/pipeline.py:122-137
  * print_accuracies table                    evaluation/utils.py:57-69
The top-k indices come from `retrieve_topk` (csrc/sim_topk.hip), so `eval_retrieval` is the drop-in for the part of
`eval_epoch` that follows the two encoding loops.
"""
from typing import Dict, List, Sequence

import numpy as np

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

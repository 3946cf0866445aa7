"""Generates the golden fixtures under tests/golden/ by executing the REFERENCE's own code from /root/reference.

Run in the development container only (the reference never travels to the GPU box):

    python tests/golden/make_golden.py

What runs for real (reference code, unmodified, imported from /root/reference):
  * models/modules.py::LanguageEncoder, models/cell_retrieval.py::CellRetrievalNetwork.encode_text   (pure torch)
  * the NumPy retrieval statements of training/coarse.py:136-140 (restated verbatim below: they live inside a
    150-line function that needs the dataset, so they cannot be imported separately)
  * the reference's glue for the cell branch -- PointNet2.forward, ObjectEncoder.forward,
    CellRetrievalNetwork.encode_objects -- on top of oracle/pyg_restated.py bound as `torch_geometric.nn`, because
    torch_geometric / torch_cluster are absent from the reference tree and from this image ("parity unpinned" at
    that boundary; see oracle/__init__.py).
Import-time stand-ins (no arithmetic): easydict, cv2, the absent dataloading.semantic3d / datapreparation.semantic3d
packages, and np.int (removed from NumPy >= 1.24, used at models/modules.py:69).
"""
import argparse
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from oracle import pyg_restated  # noqa: E402
import weights as W  # noqa: E402


def install_standins():
    np.int = int

    class EasyDict(dict):
        def __init__(self, *a, **kw):
            super().__init__(*a, **kw)
            self.__dict__ = self

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("easydict", EasyDict=EasyDict)
    mod("cv2")
    for pkg in ("dataloading.semantic3d", "datapreparation.semantic3d"):
        mod(pkg).__path__ = []
    dummy = type("Absent", (), {})
    mod("dataloading.semantic3d.semantic3d", Semantic3dCellRetrievalDataset=dummy, Semantic3dPoseReferenceMockDataset=dummy,
        Semantic3dObjectReferenceDataset=dummy)
    mod("dataloading.semantic3d.semantic3d_poses", Semantic3dPosesDataset=dummy)
    mod("dataloading.semantic3d.semantic3d_pointcloud", Semantic3dObjectDataset=dummy)
    mod("datapreparation.semantic3d.imports", Object3D=dummy, ViewObject=dummy, Pose=dummy, DescriptionObject=dummy)
    tg = mod("torch_geometric")
    tg.__path__ = []
    tg.nn = mod("torch_geometric.nn", **{k: getattr(pyg_restated, k) for k in
                                         ("fps", "radius", "knn", "PointConv", "DynamicEdgeConv", "global_max_pool",
                                          "global_mean_pool")})
    tg.transforms = mod("torch_geometric.transforms", FixedPoints=pyg_restated.FixedPoints,
                        NormalizeScale=pyg_restated.NormalizeScale, Compose=pyg_restated.Compose)
    tg.data = mod("torch_geometric.data", Data=pyg_restated.Data, Batch=pyg_restated.Batch)
    sys.path.insert(0, REF)


def ref_args(**kw):
    a = dict(embed_dim=256, use_features=["class", "color", "position"], variation=0, class_embed=False,
             color_embed=False, pointnet_layers=3, pointnet_variation=0, pointnet_numpoints=256, pointnet_freeze=False,
             pointnet_features=2)
    a.update(kw)
    return argparse.Namespace(**a)


def build_reference_model(known_classes, known_colors, known_words, seed, **kw):
    from models.cell_retrieval import CellRetrievalNetwork
    from models.pointcloud.pointnet2 import PointNet2
    args = ref_args(**kw)
    with tempfile.TemporaryDirectory() as td:
        args.pointnet_path = os.path.join(td, "pn.pth")
        torch.save(PointNet2(len(known_classes), len(known_colors), args).state_dict(), args.pointnet_path)
        model = CellRetrievalNetwork(known_classes, known_colors, known_words, args)
    W.fill_state_dict(model, seed)
    model.eval()
    return model


def golden_text(model, S):
    cases = {}
    base = S.make_texts(101, 0, 64)
    ragged = [base[0], "The pose is north of a gray road.", base[2] + " The pose is on-top of a green parking.",
              "Pose west, of a BEIGE wall.", "zzz unknown words only here", base[5], "a"]
    for name, texts in (("b1", base[:1]), ("b7_ragged_unk", ragged), ("b64", base)):
        with torch.no_grad():
            raw = model.language_encoder(texts)
            out = model.encode_text(texts)
        cases[name] = dict(texts=np.array(texts), raw=raw.numpy(), out=out.numpy())
    return cases


def golden_cells(model, S):
    from datapreparation.kitti360pose.imports import Object3d
    xyz, rgb, center, mean_rgb, _ = S.make_cells(202, 16, fixed_n=1)      # 16 single objects, regrouped below
    cell_ptr = np.array([0, 1, 7, 16], dtype=np.int32)                   # cells of 1, 6 and 9 objects
    objects, points = [], []
    for c in range(3):
        lo, hi = cell_ptr[c], cell_ptr[c + 1]
        # Object3d whose raw points have exactly the packed mean colour / centre (get_color_rgb / get_center)
        objects.append([Object3d(i, i, np.tile(center[i].astype(np.float64), (2, 1)),
                                 np.tile(mean_rgb[i].astype(np.float64), (2, 1)), "box") for i in range(lo, hi)])
        n = hi - lo
        points.append(pyg_restated.Batch(x=torch.from_numpy(rgb[lo:hi].reshape(n * 256, 3).copy()),
                                         pos=torch.from_numpy(xyz[lo:hi].reshape(n * 256, 3).copy()),
                                         batch=torch.arange(n).repeat_interleave(256)))
    grabbed = {}
    hooks = [model.object_encoder.register_forward_hook(lambda m, i, o: grabbed.__setitem__("obj_emb", o.detach().clone())),
             model.object_encoder.pointnet.register_forward_hook(
                 lambda m, i, o: grabbed.setdefault("features2", []).append(o.features2.detach().clone()))]
    with torch.no_grad():
        out = model.encode_objects(objects, points)
    for h in hooks:
        h.remove()
    return dict(xyz=xyz, rgb=rgb, center=center, mean_rgb=mean_rgb, cell_ptr=cell_ptr,
                features2=torch.cat(grabbed["features2"]).numpy(), obj_emb=grabbed["obj_emb"].numpy(), out=out.numpy())


def golden_retrieval():
    rng = np.random.default_rng(303)
    c = rng.standard_normal((300, 256)).astype(np.float32)
    q = rng.standard_normal((16, 256)).astype(np.float32)
    c /= np.linalg.norm(c, axis=1, keepdims=True)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    # training/coarse.py:100,103: float64 arrays filled with the fp32 encodings
    cell_encodings = np.zeros((len(c), 256))
    cell_encodings[:] = c
    text_encodings = np.zeros((len(q), 256))
    text_encodings[:] = q
    top = []
    for query_idx in range(len(text_encodings)):
        scores = cell_encodings[:] @ text_encodings[query_idx]     # training/coarse.py:136
        sorted_indices = np.argsort(-1.0 * scores)                 # training/coarse.py:138
        top.append(sorted_indices[0:10])                           # training/coarse.py:140 (max(top_k) = 10 at eval)
    return dict(cells=c, queries=q, top10=np.stack(top).astype(np.int64))


def golden_eval():
    """Metrics: the reference's own calc_sample_accuracies (evaluation/utils.py:31) on reference Pose / Cell objects, and
    the hit / close-by statements of training/coarse.py:142-163 (restated verbatim: they sit inside eval_epoch)."""
    from datapreparation.kitti360pose.imports import Cell, Pose, DescriptionBestCell
    from evaluation.utils import calc_sample_accuracies
    rng = np.random.default_rng(404)
    n_cells, n_q, top_k, threshs = 60, 40, [1, 5, 10], [5, 10, 15]
    scenes = ["0003", "0005"]
    cell_xy = rng.uniform(0, 200, (n_cells, 2))
    cells = []
    for i in range(n_cells):
        bbox = np.array([cell_xy[i, 0], cell_xy[i, 1], 0.0, cell_xy[i, 0] + 30, cell_xy[i, 1] + 30, 10.0])
        cells.append(Cell(i, scenes[i % 2], [], 30.0, bbox))
    cells_dict = {c.id: c for c in cells}
    db_cell_ids = np.array([c.id for c in cells])
    descr = [DescriptionBestCell.__new__(DescriptionBestCell)]
    poses = []
    for q in range(n_q):
        c = cells[rng.integers(n_cells)]
        pw = np.array([*(c.bbox_w[0:2] + rng.uniform(0, 30, 2)), 1.0])
        poses.append(Pose((pw[0:2] - c.bbox_w[0:2]) / 30.0, pw, c.id, c.scene_name, descr))
    top_idx = np.stack([rng.permutation(n_cells)[:10] for _ in range(n_q)])
    for q in range(0, n_q, 3):                       # plant the true cell at a random rank for a third of the queries
        true = int(np.where(db_cell_ids == poses[q].cell_id)[0][0])
        if true not in top_idx[q]:
            top_idx[q, rng.integers(10)] = true
    query_cell_ids = np.array([p.cell_id for p in poses])
    query_poses_w = np.array([p.pose_w[0:2] for p in poses])
    cell_size = cells[0].cell_size
    accuracies = {k: [] for k in top_k}
    accuracies_close = {k: [] for k in top_k}
    for query_idx in range(n_q):
        retrieved_cell_ids = db_cell_ids[top_idx[query_idx]]                       # training/coarse.py:143
        target_cell_id = query_cell_ids[query_idx]
        for k in top_k:
            accuracies[k].append(target_cell_id in retrieved_cell_ids[0:k])        # :146-147
        target_pose_w = query_poses_w[query_idx]
        retrieved_cell_poses = [cells_dict[cell_id].get_center()[0:2] for cell_id in retrieved_cell_ids]
        dists = np.linalg.norm(target_pose_w - retrieved_cell_poses, axis=1)       # :156
        for k in top_k:
            accuracies_close[k].append(np.any(dists[0:k] <= cell_size / 2))        # :158
    sample = {k: {t: [] for t in threshs} for k in top_k}
    for q in range(n_q):                                                            # evaluation/pipeline.py:124-131
        tc = [cells_dict[cid] for cid in db_cell_ids[top_idx[q]]]
        accs = calc_sample_accuracies(poses[q], tc, 0.5 * np.ones((10, 2)), top_k, threshs)
        for k in top_k:
            for t in threshs:
                sample[k][t].append(accs[k][t])
    return dict(cell_bbox=np.stack([c.bbox_w for c in cells]), cell_scene=np.array([c.scene_name for c in cells]),
                pose_w=np.stack([p.pose_w for p in poses]), pose_cell=query_cell_ids, top_idx=top_idx,
                hit=np.array([np.mean(accuracies[k]) for k in top_k]),
                close=np.array([np.mean(accuracies_close[k]) for k in top_k]),
                recall=np.array([[np.mean(sample[k][t]) for t in threshs] for k in top_k]))


def golden_fine(S):
    """Fine stage.  (a) `sg.*`: the reference's own models/superglue.py::SuperGlue (pure torch, runs
    unmodified) on random unit descriptors, default depth (6 x [self, cross]), 50 Sinkhorn iterations;  (b) `m.*`: the
    reference's SuperGlueMatch.forward glue (models/superglue_matcher.py:87-128) on two samples of 16 synthetic objects
    and 6 hints, with the PyG primitives restated as for the coarse fixture."""
    from models.superglue import SuperGlue
    from models.superglue_matcher import SuperGlueMatch
    from models.pointcloud.pointnet2 import PointNet2
    from datapreparation.kitti360pose.imports import Object3d
    out = {}
    torch.manual_seed(0)
    sg = SuperGlue({"descriptor_dim": 128, "GNN_layers": ["self", "cross"] * 6, "sinkhorn_iterations": 50,
                    "match_threshold": 0.2})
    W.fill_state_dict(sg, 13)
    sg.eval()
    rng = np.random.default_rng(404)
    d0 = rng.standard_normal((5, 16, 128)).astype(np.float32)
    d1 = rng.standard_normal((5, 6, 128)).astype(np.float32)
    # make some hints near-copies of objects so that real matches appear
    d1[:, :4] = d0[:, [3, 7, 0, 12]] + 0.15 * rng.standard_normal((5, 4, 128)).astype(np.float32)
    d0 /= np.linalg.norm(d0, axis=-1, keepdims=True)
    d1 /= np.linalg.norm(d1, axis=-1, keepdims=True)
    with torch.no_grad():
        r = sg(torch.from_numpy(d0).transpose(1, 2).contiguous(), torch.from_numpy(d1).transpose(1, 2).contiguous())
    out.update({"sg.desc0": d0, "sg.desc1": d1, "sg.P": r["P"].numpy(), "sg.matches0": r["matches0"].numpy(),
                "sg.matches1": r["matches1"].numpy(), "sg.matching_scores0": r["matching_scores0"].numpy(),
                "sg.matching_scores1": r["matching_scores1"].numpy()})

    known_classes = S.LABELS + ["pad"]
    args = ref_args(embed_dim=128, num_layers=2, sinkhorn_iters=50)
    with tempfile.TemporaryDirectory() as td:
        args.pointnet_path = os.path.join(td, "pn.pth")
        torch.save(PointNet2(len(known_classes), len(S.COLOR_NAMES), args).state_dict(), args.pointnet_path)
        model = SuperGlueMatch(known_classes, S.COLOR_NAMES, S.known_words(), args)
    W.fill_state_dict(model, 14)
    model.eval()
    xyz, rgb, center, mean_rgb, cell_ptr = S.make_cells(505, 2, fixed_n=16)
    objects, points = [], []
    for c in range(2):
        lo, hi = cell_ptr[c], cell_ptr[c + 1]
        objects.append([Object3d(i, i, np.tile(center[i].astype(np.float64), (2, 1)),
                                 np.tile(mean_rgb[i].astype(np.float64), (2, 1)), "box") for i in range(lo, hi)])
        n = hi - lo
        points.append(pyg_restated.Batch(x=torch.from_numpy(rgb[lo:hi].reshape(n * 256, 3).copy()),
                                         pos=torch.from_numpy(xyz[lo:hi].reshape(n * 256, 3).copy()),
                                         batch=torch.arange(n).repeat_interleave(256)))
    flat = S.make_texts(606, 0, 12, n_hints=1)
    hints = [flat[0:6], flat[6:12]]
    grabbed = {}
    h = model.superglue.register_forward_pre_hook(lambda m, a: grabbed.update(desc0=a[0].detach().clone(), desc1=a[1].detach().clone()))
    with torch.no_grad():
        o = model(objects, hints, points)
    h.remove()
    out.update({"m.xyz": xyz, "m.rgb": rgb, "m.center": center, "m.mean_rgb": mean_rgb, "m.cell_ptr": cell_ptr,
                "m.hints": np.array(hints), "m.object_encodings": grabbed["desc0"].transpose(1, 2).numpy(),
                "m.hint_encodings": grabbed["desc1"].transpose(1, 2).numpy(), "m.P": o.P.numpy(),
                "m.matches0": o.matches0.numpy(), "m.matches1": o.matches1.numpy(), "m.offsets": o.offsets.numpy(),
                "m.matching_scores0": o.matching_scores0.numpy(), "m.matching_scores1": o.matching_scores1.numpy()})
    return out


def golden_scene():
    """A toy KITTI360Pose scene written by the REFERENCE's own classes (datapreparation/kitti360pose/imports.py), once under
    the current module path and once under the legacy `datapreparation.kitti360` path that dataloading/__init__.py:8-10
    keeps loadable: tests/golden/scene/{cells,poses}/{toy,toy_legacy}.pkl (pure data: 3 cells, 4 poses)."""
    import pickle
    import datapreparation.kitti360pose.imports as I
    rng = np.random.default_rng(7)
    cells = []
    for i in range(3):
        objs = [I.Object3d(j, 100 + j, rng.random((12, 3)), rng.random((12, 3)), ["box", "road", "pole"][j % 3]) for j in range(2 + i)]
        cells.append(I.Cell(i, "toy0", objs, 30.0, np.array([30.0 * i, 0.0, 0.0, 30.0 * i + 30.0, 30.0, 10.0])))
    poses = []
    for q in range(4):
        cell = cells[q % 3]
        descs = []
        for h in range(6):
            o = cell.objects[h % len(cell.objects)]
            dp = I.DescriptionPoseCell(o, ["north", "south", "east", "west", "on-top"][h % 5], rng.random(3), rng.random(3), rng.random(3))
            descs.append(I.DescriptionBestCell.from_matched(dp, o.id, rng.random(3), rng.random(3), rng.random(3)) if h % 2 == 0
                         else I.DescriptionBestCell.from_unmatched(dp))
        poses.append(I.Pose(rng.random(3), np.array([30.0 * (q % 3) + 12.0, 14.0, 0.0]), cell.id, "toy0", descs))
    base = os.path.join(HERE, "scene")
    for sub, obj in (("cells", cells), ("poses", poses)):
        os.makedirs(os.path.join(base, sub), exist_ok=True)
        with open(os.path.join(base, sub, "toy.pkl"), "wb") as f:
            pickle.dump(obj, f, protocol=4)
    # legacy module path: what a file written before the package was renamed looks like
    import datapreparation.kitti360pose as pkg
    sys.modules["datapreparation.kitti360"] = pkg
    sys.modules["datapreparation.kitti360.imports"] = I
    classes = [I.Object3d, I.Cell, I.Pose, I.DescriptionPoseCell, I.DescriptionBestCell]
    for c in classes:
        c.__module__ = "datapreparation.kitti360.imports"
    try:
        for sub, obj in (("cells", cells), ("poses", poses)):
            with open(os.path.join(base, sub, "toy_legacy.pkl"), "wb") as f:
                pickle.dump(obj, f, protocol=4)
    finally:
        for c in classes:
            c.__module__ = "datapreparation.kitti360pose.imports"
    return dict(cell_ids=np.array([c.id for c in cells]), n_objects=np.array([len(c.objects) for c in cells]),
                centers=np.array([c.get_center() for c in cells]), obj0_center=cells[0].objects[0].get_center(),
                obj0_color=np.array(cells[0].objects[0].get_color_text()), pose_w=np.array([p.pose_w for p in poses]),
                pose_cell=np.array([p.cell_id for p in poses]),
                hint0=np.array(f"The pose is {poses[0].descriptions[0].direction} of a "
                               f"{poses[0].descriptions[0].object_color_text} {poses[0].descriptions[0].object_label}."),
                matched=np.array([[d.is_matched for d in p.descriptions] for p in poses]))


def main():
    install_standins()
    import importlib
    S = importlib.import_module("text2pos-cvpr2022_amd.synthetic")
    known_classes = S.LABELS + ["pad"]
    model = build_reference_model(known_classes, S.COLOR_NAMES, S.known_words(), seed=11)
    text = golden_text(model, S)
    np.savez_compressed(os.path.join(HERE, "text_encoder.npz"),
                        **{f"{case}.{k}": v for case, d in text.items() for k, v in d.items()})
    cells = golden_cells(model, S)
    np.savez_compressed(os.path.join(HERE, "cell_encoder.npz"), **cells)
    np.savez_compressed(os.path.join(HERE, "retrieval.npz"), **golden_retrieval())
    np.savez_compressed(os.path.join(HERE, "eval_metrics.npz"), **golden_eval())
    np.savez_compressed(os.path.join(HERE, "fine.npz"), **golden_fine(S))
    np.savez_compressed(os.path.join(HERE, "scene_expect.npz"), **golden_scene())
    for f in ("text_encoder.npz", "cell_encoder.npz", "retrieval.npz", "eval_metrics.npz", "fine.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()

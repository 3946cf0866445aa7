"""End-to-end evaluation: coarse retrieval, then fine localisation of every query against its top-k cells
(the reference's evaluation/pipeline.py:60-137 run_coarse, :172-279 run_fine, :282-342 main), on the MI355X path.

    python -m text2pos_amd.pipeline --base_path ./data/k360_30-10_scG_pd10_pc4_spY_all/ \\
        --path_coarse ./checkpoints/coarse.pth --path_fine ./checkpoints/fine.pth [--scenes 2013_05_28_drive_0010_sync ...]

Checkpoints are the reference's whole-module pickles (io.load_reference_checkpoint) or plain state_dicts.
BASELINE.json configs[4].  Multi-GPU: launched with torch.distributed.run (one process per GPU), run_coarse shards the
cells and the queries over the ranks through distributed.sharded_retrieval - the function bench.py's step runs - with one
RCCL all-gather of the cell embeddings; the fine stage then splits the queries over the ranks (evaluate()).
"""
import argparse
from types import SimpleNamespace
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import data as D
from . import evaluation as E
from . import io as IO
from .retrieval import retrieve_topk

# validation scenes of KITTI360Pose (datapreparation/kitti360pose/utils.py: SCENE_NAMES_VAL)
SCENE_NAMES_VAL = ["2013_05_28_drive_0010_sync"]


def default_transform(n_pts: int = 256, seed: Optional[int] = None):
    """evaluation/pipeline.py:290-293: FixedPoints(pointnet_numpoints) + NormalizeScale (no augmentation)."""
    return D.Compose([D.FixedPoints(n_pts, generator=np.random.default_rng(seed)), D.NormalizeScale()])


class PerCellTransform:
    """FixedPoints + NormalizeScale whose random draw depends on (seed, global cell index, object slot) only - a counter-based
    hash (data.sample_keys / data.keyed_draws), not a sequential generator - so that a cell gets the same 256 points whichever
    rank, batch or device packs it: the sharded pipeline reproduces the single-process result bit for bit, and the on-device
    dataloader (scene.DeviceScene, csrc/small_kernels.hip::k_pack_scene) draws exactly what `for_cell`'s host chain draws.
    (`default_transform` keeps one sequential generator, like the reference's dataloader; its draws depend on the order in
    which a process walks the cells.)"""

    def __init__(self, n_pts: int = 256, seed: int = 0):
        self.n_pts, self.seed = n_pts, seed

    def for_cell(self, cell_index: int):
        """The host chain for one cell / sample (applied to its objects in order)."""
        return D.Compose([D.KeyedFixedPoints(self.n_pts, self.seed, cell_index), D.NormalizeScale()])

    def keys(self, sample_index, slot) -> np.ndarray:
        """uint64 sampling keys of (sample, slot) pairs (arrays broadcast): the `key` argument of t2p_pack_scene_objects."""
        return D.sample_keys(self.seed, sample_index, slot)


def on_device_input(model, transform) -> bool:
    """True when the input side can run on the GPU: a counter-based per-cell transform (PerCellTransform) and a model with
    the scene entry points on a GPU.  Anything else (a sequential generator like default_transform, an arbitrary callable,
    the gloo tests' stand-in models) takes the reference's host chain."""
    dev = getattr(model, "device", None)
    return hasattr(transform, "keys") and hasattr(transform, "n_pts") and dev is not None and torch.device(dev).type == "cuda" and \
        (hasattr(model, "encode_scene_cells") or hasattr(model, "forward_packed"))


@torch.no_grad()
def run_coarse(model, scenes: IO.Scenes, transform, top_k: Sequence[int], threshs: Sequence[int], cells_per_call: int = 512,
               texts_per_call: int = 1024, group=None, topk_fn=None, scene_dev=None, timings: Optional[dict] = None,
               on_device: Optional[bool] = None):
    """Encode every cell and every query, rank in float64, report hit@k / close-by@k and recall within the thresholds
    when the retrieved cell's centre is the estimate.  Returns (retrievals, accuracies dict).

    The retrieval goes through distributed.sharded_retrieval: with an initialised torch.distributed process group (one
    process per GPU, backend "nccl" = RCCL) every rank encodes its contiguous block of the cells and of the queries, ONE
    all-gather exchanges the cell embeddings, every rank ranks its query block against the full database, and the [Nq, k]
    index lists are gathered, so every rank returns the same tables; without a process group it is the single-GPU path
    (BASELINE configs[4]; evaluation/pipeline.py:60-137).
    Input side: with a PerCellTransform and the product model the cells' raw objects are uploaded once (scene.DeviceScene;
    `scene_dev` = one that already holds the WHOLE scene, else this rank's block is built here) and every cell is resampled,
    normalised and encoded on the GPU (model.encode_scene_cells) - the same packed inputs, bit for bit, as the host chain
    `transform.for_cell(i)` + encode_objects, which remains the path of every other transform / model (cells_per_call cells per
    call there: the reference's loader hands over 64, evaluation/pipeline.py:303-308; cells are independent).
    on_device: None (default) = the on-device input side whenever transform and model allow it, falling back to the host chain
    (with a warning) when this rank's block cannot be held as one DeviceScene (an empty cell, more than 2^31 raw points); True =
    insist (such a block raises); False = the host chain.
    topk_fn(queries, cells, k): the ranking kernel (default retrieval.retrieve_topk; the gloo CPU test injects the oracle's).
    timings: optional dict that receives the wall time of the upload (`scene_s`)."""
    import time
    import warnings
    from . import distributed as TD
    cells, poses = scenes.all_cells, scenes.all_poses
    texts = scenes.texts
    can = (scene_dev is not None or on_device_input(model, transform)) and hasattr(model, "encode_scene_cells")
    if on_device and not can:
        raise RuntimeError("run_coarse(on_device=True) needs a counter-based transform (PerCellTransform) and a model with encode_scene_cells")
    on_dev = can and on_device is not False

    def encode_cells(lo, hi):
        if on_dev:
            if scene_dev is not None:
                return model.encode_scene_cells(scene_dev, transform, lo, hi)
            from .scene import DeviceScene
            t0 = time.perf_counter()
            try:
                block = DeviceScene(cells[lo:hi], model.device)
            except RuntimeError as e:
                if on_device:
                    raise
                warnings.warn(f"run_coarse: cells [{lo}, {hi}) do not fit one DeviceScene ({e}); this block takes the host chain",
                              RuntimeWarning)
                block = None
            if block is not None:
                if timings is not None:
                    timings["scene_s"] = time.perf_counter() - t0
                return model.encode_scene_cells(block, transform, 0, hi - lo, cell_offset=lo)
        enc = []
        for a in range(lo, hi, cells_per_call):
            b = min(a + cells_per_call, hi)
            # (the cells' own lists, not copies: CellRetrievalNetwork.object_means_cache recognises a cell by its list)
            objects = [c.objects if isinstance(c.objects, list) else list(c.objects) for c in cells[a:b]]
            if hasattr(transform, "for_cell"):
                points = [D.batch_object_points(o, transform.for_cell(a + i)) for i, o in enumerate(objects)]
            else:
                points = [D.batch_object_points(o, transform) for o in objects]
            enc.append(model.encode_objects(objects, points))
        return torch.cat(enc) if enc else torch.zeros((0, model.embed_dim), device=model.device)

    def encode_queries(lo, hi):
        enc = [model.encode_text(texts[a: min(a + texts_per_call, hi)]) for a in range(lo, hi, texts_per_call)]
        return torch.cat(enc) if enc else torch.zeros((0, model.embed_dim), device=model.device)

    kmax = int(max(top_k))
    rank_fn = topk_fn if topk_fn is not None else (lambda q, c, k: retrieve_topk(c, q, k))
    idx, _ = TD.sharded_retrieval(encode_cells, encode_queries, rank_fn, len(cells), len(texts), kmax, group)
    idx = np.asarray(idx.cpu()) if hasattr(idx, "cpu") else np.asarray(idx)
    db_ids = [c.id for c in cells]
    centers = np.array([c.get_center()[0:2] for c in cells])
    cell_size = float(cells[0].cell_size)
    acc, acc_close, _ = E.retrieval_accuracies(idx, db_ids, [p.cell_id for p in poses],
                                               np.array([p.pose_w for p in poses]), centers, cell_size, list(top_k))
    retrievals = [[db_ids[j] for j in row[:kmax]] for row in idx]
    loc = E.localisation_accuracies(poses, retrievals, scenes.cells_dict, list(top_k), list(threshs))
    return retrievals, dict(hit=acc, close=acc_close, localisation=loc)


@torch.no_grad()
def evaluate(model_coarse, model_fine, scenes: IO.Scenes, transform, top_k=(1, 5, 10), threshs=(5, 10, 15), pad_size=16,
             queries_per_call: Optional[int] = None, group=None, topk_fn=None, timings: Optional[dict] = None,
             on_device: Optional[bool] = None) -> Dict[str, object]:
    """Coarse retrieval + fine localisation.  `group`: torch.distributed process group (None = the default group when one
    is initialised, else single GPU); every rank returns the same tables.
    With a PerCellTransform and the product models the whole scene is uploaded once (scene.DeviceScene, with the fine stage's
    padding objects) and serves both stages: the coarse stage resamples and encodes its cells from it, the fine stage packs its
    (query, candidate cell) samples from it - no per-object host work after the upload.  The WHOLE scene is uploaded on every rank
    only when the fine stage needs it (its candidates are cells of any rank's block); a coarse-only evaluation uploads each rank's own
    block (run_coarse), so input-side cost and HBM use shrink with the number of ranks.
    on_device: None (default) = on-device input wherever possible, host chain (with a warning) when the scene cannot be held as one
    DeviceScene (an empty cell, more than 2^31 raw points); True = insist; False = the host chain for both stages.
    queries_per_call: evaluation.run_fine's memory knob (None = its defaults).
    timings: optional dict that receives wall times (`scene_s` upload, `coarse_s`, `fine_s`)."""
    import time
    import warnings
    scene_dev = None
    t0 = time.perf_counter()
    coarse_can = on_device_input(model_coarse, transform) and hasattr(model_coarse, "encode_scene_cells")
    fine_can = (model_fine is not None and on_device_input(model_fine, transform)
                and all(hasattr(model_fine, a) for a in ("forward_packed", "encode_hints")))
    if on_device and not coarse_can:
        raise RuntimeError("evaluate(on_device=True) needs a counter-based transform (PerCellTransform) and the product models")
    if on_device is not False and coarse_can and fine_can:
        from .scene import DeviceScene
        try:
            scene_dev = DeviceScene(scenes.all_cells, model_coarse.device, n_pad=pad_size,
                                    pad_seed=getattr(transform, "seed", 0))
        except RuntimeError as e:
            if on_device:
                raise
            warnings.warn(f"evaluate: the scene does not fit one DeviceScene ({e}); both stages take the host chain", RuntimeWarning)
            on_device = False
    t1 = time.perf_counter()
    retrievals, out = run_coarse(model_coarse, scenes, transform, top_k, threshs, group=group, topk_fn=topk_fn, scene_dev=scene_dev,
                                 on_device=on_device, timings=timings if scene_dev is None else None)
    out["retrievals"] = retrievals
    t2 = time.perf_counter()
    if model_fine is not None:
        mean, off, conf = E.run_fine(model_fine, scenes.all_poses, scenes.cells_dict, retrievals, transform, pad_size,
                                     list(top_k), list(threshs), queries_per_call, group=group, scene_dev=scene_dev)
        out.update(fine_mean=mean, fine_offset=off, fine_mean_conf=conf)
    if timings is not None:
        own_upload = timings.get("scene_s", 0.0) if scene_dev is None else 0.0      # (run_coarse's upload of this rank's block)
        timings.update(scene_s=(t1 - t0) + own_upload, coarse_s=(t2 - t1) - own_upload, fine_s=time.perf_counter() - t2)
    return out


def _model_args(embed_dim, num_layers=6, sinkhorn_iters=50, use_features=("class", "color", "position")):
    return SimpleNamespace(embed_dim=embed_dim, use_features=list(use_features), variation=0, class_embed=False,
                           color_embed=False, pointnet_layers=3, pointnet_variation=0, pointnet_numpoints=256,
                           pointnet_path=None, pointnet_freeze=False, pointnet_features=2, num_layers=num_layers,
                           sinkhorn_iters=sinkhorn_iters)


_ARCH_KEYS = ("embed_dim", "use_features", "variation", "class_embed", "color_embed", "pointnet_features", "num_layers",
              "sinkhorn_iters", "pointnet_numpoints")


def args_from_checkpoint(cli_args: SimpleNamespace, ckpt_args: dict, what: str) -> SimpleNamespace:
    """The architecture switches a state_dict cannot express (variation 0 / 1 share every key; so do use_features subsets
    of equal size) come from the arguments pickled inside the reference's whole-module checkpoint, which is what the
    reference evaluates with (evaluation/pipeline.py:313-314).  Values present in the checkpoint win over the command
    line; a difference is reported."""
    a = SimpleNamespace(**vars(cli_args))
    for k in _ARCH_KEYS:
        if k in ckpt_args and ckpt_args[k] is not None:
            v = ckpt_args[k]
            v = list(v) if k == "use_features" else v
            if hasattr(a, k) and getattr(a, k) != v:
                print(f"[{what}] checkpoint was trained with {k}={v!r} (command line / default: {getattr(a, k)!r}): using the checkpoint's")
            setattr(a, k, v)
    return a


def main(argv: Optional[List[str]] = None):
    from . import CellRetrievalNetwork, SuperGlueMatch
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--base_path", required=True)
    ap.add_argument("--path_coarse", required=True)
    ap.add_argument("--path_fine", default=None)
    ap.add_argument("--scenes", nargs="*", default=SCENE_NAMES_VAL)
    ap.add_argument("--top_k", type=int, nargs="+", default=[1, 5, 10])
    ap.add_argument("--threshs", type=int, nargs="+", default=[5, 10, 15])
    ap.add_argument("--pad_size", type=int, default=16)
    ap.add_argument("--coarse_embed_dim", type=int, default=256)
    ap.add_argument("--fine_embed_dim", type=int, default=128)
    ap.add_argument("--fine_num_layers", type=int, default=6)
    ap.add_argument("--sinkhorn_iters", type=int, default=50)
    ap.add_argument("--use_features", nargs="+", default=["class", "color", "position"])
    ap.add_argument("--seed", type=int, default=0, help="seed of the T.FixedPoints draw")
    a = ap.parse_args(argv)
    # one process per GPU under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from its environment)
    import os
    world, rank, local_rank = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    say = print if rank == 0 else (lambda *x, **k: None)
    scenes = IO.load_scenes(a.base_path, a.scenes)
    say(f"{len(scenes.all_cells)} cells, {len(scenes.all_poses)} poses from {a.scenes}")
    words, classes = scenes.get_known_words(), scenes.get_known_classes()
    sd, ck = IO.load_reference_checkpoint(a.path_coarse, return_args=True)
    ca = args_from_checkpoint(_model_args(a.coarse_embed_dim, use_features=a.use_features), ck, "coarse")
    coarse = CellRetrievalNetwork(classes, D.COLOR_NAMES, words, ca)
    coarse.load_state_dict(sd)
    coarse = coarse.to(dev).eval()
    fine, n_pts = None, int(getattr(ca, "pointnet_numpoints", 256))
    if a.path_fine:
        sd, ck = IO.load_reference_checkpoint(a.path_fine, return_args=True)
        fa = args_from_checkpoint(_model_args(a.fine_embed_dim, a.fine_num_layers, a.sinkhorn_iters, a.use_features), ck, "fine")
        fine = SuperGlueMatch(classes, D.COLOR_NAMES, words, fa)
        fine.load_state_dict(sd)
        fine = fine.to(dev).eval()
    # per-cell seeding: the sampled points of a cell do not depend on which rank encodes it
    out = evaluate(coarse, fine, scenes, PerCellTransform(n_pts, a.seed), a.top_k, a.threshs, a.pad_size)
    if rank == 0:
        print("Retrieval accuracies (hit@k):", out["hit"], " close-by@k:", out["close"])
        print("Coarse (cell centre):")
        E.print_accuracies(out["localisation"])
        if fine is not None:
            for name in ("fine_mean", "fine_offset", "fine_mean_conf"):
                print(name + ":")
                E.print_accuracies(out[name])
    if world > 1:
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()

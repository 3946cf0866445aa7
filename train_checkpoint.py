#!/usr/bin/env python
"""train_checkpoint.py -- a TRAINED checkpoint of the coarse model, made by this repo's own training path.

The released Text2Pos checkpoints and KITTI360Pose are not available here (no network), so every number of rounds 1-5 was
measured on random-init weights with calibrated BatchNorm statistics.  This script removes that unknown as far as the box
allows: it runs the reference's coarse training loop (training/coarse.py:31-62, mirrored by text2pos_amd.training.train_epoch:
model.train(); anchor = encode_text(texts); positive = encode_objects(objects, object_points); PairwiseRankingLoss(0.35); Adam
1e-3 - training/args.py:30,46) on synthetic (description, cell) PAIRS (synthetic.make_paired_texts: text i describes six
objects of cell i), writes the result the reference's way - `torch.save(model, path)`, the whole pickled module
(training/coarse.py:323-324) - and reads it back through io.load_reference_checkpoint, the loader a reference checkpoint would
go through.  What comes out has what a trained model has and a random one lacks: weights moved by a few hundred Adam steps,
BatchNorm running estimates accumulated with momentum 0.1 over the training batches, embeddings that separate cells (hit@k far
above chance on held-out cells).

    python train_checkpoint.py --out gpurun_out/trained.pth          # needs cuda:0 (the training path is HIP)

bench.py --weights trained | <file> and tests/test_gpu_headline.py (checkpoint "trained") use the functions below.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

TRAIN_SEED = 20220077      # the training cells' stream (bench.py's workload is 20220002: held out)
EVAL_SEED = 20220078       # held-out (description, cell) pairs for hit@k
DEFAULTS = dict(steps=320, batch=64, train_cells=4096, lr=1e-3, margin=0.35)


def collate(seed, n_total, lo, hi):
    """Cells [lo, hi) of the `seed` stream in the shape Kitti360CoarseDataset.collate_fn hands to the training loop
    (dataloading/kitti360pose/cells.py:98-110): texts, List[List[Object3d]], List[Batch].  The Object3d of a synthetic object
    carries two copies of its centre / mean colour as its "raw" points, so that get_center() / get_color_rgb()
    (models/object_encoder.py:121-131) return what the generator drew."""
    import torch
    from text2pos_amd import data as D, synthetic as S
    xyz, rgb, center, mean_rgb, ptr = S.make_cells(seed, n_total, lo, hi)
    shape, _, _ = S.object_attributes(seed, *(int(v) for v in _object_range(S, seed, n_total, lo, hi)))
    objects, points = [], []
    for c in range(hi - lo):
        a, b = int(ptr[c]), int(ptr[c + 1])
        objects.append([D.Object3d(i - a, i, np.tile(center[i].astype(np.float64), (2, 1)), np.tile(mean_rgb[i].astype(np.float64), (2, 1)),
                                   S.LABEL_GROUPS[int(shape[i])][0]) for i in range(a, b)])
        points.append(D.Batch(x=torch.from_numpy(rgb[a:b].reshape(-1, 3)), pos=torch.from_numpy(xyz[a:b].reshape(-1, 3)),
                              batch=torch.arange(b - a).repeat_interleave(xyz.shape[1])))
    return dict(texts=S.make_paired_texts(seed, n_total, lo, hi), objects=objects, object_points=points)


def _object_range(S, seed, n_total, lo, hi):
    ptr = np.zeros(n_total + 1, dtype=np.int64)
    ptr[1:] = np.cumsum(S.cell_sizes(seed, n_total))
    return ptr[lo], ptr[hi]


def train(model, steps=DEFAULTS["steps"], batch=DEFAULTS["batch"], train_cells=DEFAULTS["train_cells"], lr=DEFAULTS["lr"],
          margin=DEFAULTS["margin"], seed=TRAIN_SEED, log=None):
    """`steps` optimizer steps of training/coarse.py's loop over `train_cells` synthetic pairs (epochs of train_cells / batch
    steps, batch order reshuffled per epoch as --shuffle does).  Returns the per-epoch mean losses.  The model is left in
    eval() mode, as the reference's loop leaves it after eval_epoch."""
    import torch
    from text2pos_amd import training as T
    batches = [collate(seed, train_cells, lo, min(lo + batch, train_cells)) for lo in range(0, train_cells - batch + 1, batch)]
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    crit = T.make_criterion(argparse.Namespace(ranking_loss="pairwise", margin=margin))
    rng = np.random.default_rng(seed)
    losses, done = [], 0
    while done < steps:
        order = rng.permutation(len(batches))[: steps - done]
        loss, seen = T.train_epoch(model, [batches[i] for i in order], opt, crit)
        done += len(seen)
        losses.append(loss)
        if log:
            log(f"training: {done}/{steps} steps, epoch mean loss {loss:.4f}")
    model.eval()
    return losses


def hit_at_k(model, n_cells=2048, ks=(1, 5, 10), seed=EVAL_SEED):
    """Retrieval accuracy on HELD-OUT pairs (training/coarse.py:142-158's hit@k): text i against the n_cells cells of the `seed`
    stream, a hit when cell i is among the top k.  Chance = k / n_cells."""
    import torch
    import text2pos_amd as t2p
    from text2pos_amd import synthetic as S
    dev = model.device
    xyz, rgb, center, mean_rgb, ptr = S.make_cells(seed, n_cells)
    with torch.no_grad():
        cells = model.encode_objects_packed(*(torch.from_numpy(a).to(dev) for a in (xyz, rgb, center, mean_rgb)), ptr)
        queries = model.encode_text(S.make_paired_texts(seed, n_cells))
        idx, _ = t2p.retrieve_topk(cells, queries, max(ks))
    target = torch.arange(n_cells, device=idx.device)[:, None]
    return {int(k): float((idx[:, :k] == target).any(dim=1).float().mean().item()) for k in ks}


def save_reference_style(model, path):
    """training/coarse.py:323-324: torch.save(model, model_path) - the whole module, on the CPU."""
    import copy
    import torch
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(copy.deepcopy(model).cpu(), path)


def load_into(model, path):
    """A checkpoint file (whole pickled module, the reference's format, or a bare state_dict) -> model.load_state_dict."""
    from text2pos_amd import io as IO
    sd, args = IO.load_reference_checkpoint(path, return_args=True)
    for key in ("embed_dim", "variation", "pointnet_features"):
        if key in args and getattr(model.args, key, args[key]) != args[key]:
            raise RuntimeError(f"{path} was trained with {key}={args[key]!r}, the model was built with {getattr(model.args, key)!r}")
    model.load_state_dict(sd, strict=True)
    return model


def fresh_model(precision="f16x3", device="cuda:0"):
    import torch
    import text2pos_amd as t2p
    from text2pos_amd import synthetic as S
    torch.manual_seed(1234)
    return t2p.CellRetrievalNetwork(S.LABELS + ["pad"], S.COLOR_NAMES, S.known_words(), S.default_args(), precision=precision).to(device)


def trained_model(path=None, precision="f16x3", device="cuda:0", log=None, **kw):
    """A model carrying trained weights: from `path` when it exists, else trained here (and written to `path` when given) and
    then - either way - loaded back through the reference-format file, so the weights always took the route a reference
    checkpoint takes.  Returns (model in eval mode, info dict)."""
    import tempfile
    info = {"source": path}
    if path is None or not os.path.exists(path):
        m = fresh_model(precision, device)
        before = hit_at_k_untrained(m)
        t0 = time.perf_counter()
        losses = train(m, log=log, **kw)
        info.update(trained_here=True, train_s=round(time.perf_counter() - t0, 1), epoch_losses=[round(v, 4) for v in losses],
                    hit_at_k_before_training=before, **{k: kw.get(k, DEFAULTS[k]) for k in DEFAULTS})
        tmp = None
        if path is None:
            tmp = tempfile.NamedTemporaryFile(suffix=".pth", delete=False)
            tmp.close()
            path = tmp.name
        save_reference_style(m, path)
        info["bytes"] = os.path.getsize(path)
        del m
    model = load_into(fresh_model(precision, device), path).eval()
    if info.get("trained_here") and info["source"] is None:
        os.unlink(path)
    info["hit_at_k_held_out_2048_cells"] = hit_at_k(model)
    return model, info


def hit_at_k_untrained(model):
    """hit@k of a random-init model needs BatchNorm statistics to exist: eval() on the constructor's (mean 0, var 1)."""
    model.eval()
    try:
        return hit_at_k(model)
    except FloatingPointError:      # random init + unit statistics may leave the f16x3 range: not what this script is about
        return None
    finally:
        model.train()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "trained.pth"))
    for k, v in DEFAULTS.items():
        ap.add_argument("--" + k.replace("_", "-"), type=type(v), default=v)
    args = ap.parse_args()
    log = lambda m: print(m, file=sys.stderr, flush=True)
    if os.path.exists(args.out):
        os.unlink(args.out)
    model, info = trained_model(args.out, log=log, **{k: getattr(args, k) for k in DEFAULTS})
    code = model.overflow_detected()
    info["fp16_range_guard_after_hit_at_k"] = "clear" if code == 0 else hex(code)
    print(json.dumps(info))


if __name__ == "__main__":
    main()

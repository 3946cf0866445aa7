"""Host loop of the reference's coarse training (training/coarse.py:31-62, `train_epoch`) on the HIP path: same batch
dictionary (`texts`, `objects`, `object_points` as the reference's Kitti360CoarseDataset.collate_fn yields them), same
order of calls; the arithmetic is docs/notebook.md 4.8.  The data side (datasets, augmentation, plotting) stays with the caller."""
from typing import Iterable, Optional

import numpy as np
import torch

from . import ops
from .losses import HardestRankingLoss, PairwiseRankingLoss


def make_criterion(args) -> torch.nn.Module:
    """training/coarse.py:279-284: --ranking_loss pairwise (the default, training/args.py:48) or hardest.  `triplet` is not
    built: it needs the dataset's negative cells, and the reference's own branch cannot run as written - it calls
    `model.encode_objects(negative_cell_objects)` without the `object_points` argument the method requires
    (training/coarse.py:48-51 against models/cell_retrieval.py:77)."""
    kind = getattr(args, "ranking_loss", "pairwise")
    margin = getattr(args, "margin", 0.35)
    if kind == "pairwise":
        return PairwiseRankingLoss(margin=margin)
    if kind == "hardest":
        return HardestRankingLoss(margin=margin)
    raise NotImplementedError(f"ranking_loss={kind!r}: 'pairwise' and 'hardest' are built")


_TEXT_STREAMS = {}    # device -> the text branch's stream (picked once: ops.concurrent_stream probes the hardware queues)


def train_epoch(model, dataloader: Iterable[dict], optimizer, criterion, max_batches: Optional[int] = None,
                overlap_text: bool = True):
    """One pass over `dataloader` (training/coarse.py:31-62).  Returns (mean loss, the batches seen).
    overlap_text: the text branch runs on a second HIP stream beside the cell branch - the two meet only in the loss, and autograd
    runs a node's backward on the stream of its forward, so the biLSTM's step-by-step recurrence (a chain of small latency-bound
    kernels, ~3 ms of a 64 + 64 step) runs beside the cell branch's matrix kernels in both directions.  Same kernels, same
    arithmetic: the loss of a step is bit-identical to overlap_text=False, its gradients agree to the rounding noise that the
    float atomics of the scatter-backward kernels have from run to run anyway (tested).  Worth 0.4 ms of 27.8 on one MI355X: the
    step is paced by the host's launch / size-read-back ping-pong, not by the GPU."""
    model.train()
    epoch_losses, batches = [], []
    dev = model.device
    side = None
    if overlap_text and dev.type == "cuda":
        if str(dev) not in _TEXT_STREAMS:
            _TEXT_STREAMS[str(dev)] = ops.concurrent_stream(dev)
        side = _TEXT_STREAMS[str(dev)]
    for i_batch, batch in enumerate(dataloader):
        if max_batches is not None and i_batch >= max_batches:
            break
        optimizer.zero_grad()
        if side is not None:
            main = torch.cuda.current_stream(dev)
            side.wait_stream(main)                       # zero_grad / the previous optimizer step
            with torch.cuda.stream(side):
                anchor = model.encode_text(batch["texts"])
            positive = model.encode_objects(batch["objects"], batch["object_points"])
            main.wait_stream(side)
            anchor.record_stream(main)                   # allocated on the side stream's pool, consumed by the loss on the main one
        else:
            anchor = model.encode_text(batch["texts"])
            positive = model.encode_objects(batch["objects"], batch["object_points"])
        loss = criterion(anchor, positive)
        loss.backward()
        optimizer.step()
        epoch_losses.append(loss.item())
        batches.append(batch)
    return float(np.mean(epoch_losses)) if epoch_losses else float("nan"), batches

"""Multi-GPU layout of the path: one process per GPU, cells sharded in contiguous blocks, one all-gather of the cell
embeddings (RCCL over xGMI when the backend is "nccl"), queries sharded for the similarity + top-k.

The reference is single-process; this is the layout BASELINE.json's north_star prescribes.
Contiguous blocks keep `global cell index = shard offset + local index`, so the gathered top-k indices are identical
to the single-GPU result.  The compute steps are injected (encode_fn / topk_fn), which is how the gloo CPU tests
exercise the partition + collective + index logic without a GPU.
"""
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of rank `rank`; the first n % world ranks hold one extra item."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def rank_and_world(group=None) -> Tuple[int, int]:
    """(rank, world size) of the process group; (0, 1) when torch.distributed is not initialised (single-GPU runs)."""
    if not (dist.is_available() and dist.is_initialized()):
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def all_gather_rows(local: torch.Tensor, n_total: int, group=None, force: bool = False) -> torch.Tensor:
    """Gather row blocks of unequal size (shard_range layout) into the full [n_total, D] matrix on every rank.
    One collective: shards are padded to the largest block so that all_gather_into_tensor applies.
    force: issue the collective even in a process group of ONE rank (bench.py --force-exchange: the RCCL leg - communicator
    set-up, device-tensor all_gather_into_tensor, stream ordering - then runs on a single GPU exactly as it will on eight)."""
    _, world = rank_and_world(group)
    if world == 1 and not (force and dist.is_available() and dist.is_initialized()):
        return local
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    max_rows = max(hi - lo for lo, hi in sizes)
    d = local.shape[1]
    padded = local
    if local.shape[0] < max_rows:
        padded = torch.zeros((max_rows, d), dtype=local.dtype, device=local.device)
        padded[: local.shape[0]] = local
    gathered = torch.empty((world * max_rows, d), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(gathered, padded.contiguous(), group=group)
    if all(hi - lo == max_rows for lo, hi in sizes):
        return gathered
    return torch.cat([gathered[r * max_rows: r * max_rows + (hi - lo)] for r, (lo, hi) in enumerate(sizes)], 0)


def sharded_retrieval(encode_local_cells: Callable[[int, int], torch.Tensor],
                      encode_local_queries: Callable[[int, int], torch.Tensor],
                      topk_fn: Callable[[torch.Tensor, torch.Tensor, int], Tuple[torch.Tensor, torch.Tensor]],
                      n_cells: int, n_queries: int, k: int, group=None, gather_result: bool = True,
                      around_exchange: Optional[Callable[[str], None]] = None, force_exchange: bool = False):
    """encode_local_cells(lo, hi) -> [hi-lo, D] embeddings of this rank's cell block (same for queries);
    topk_fn(queries, cells, k) -> (idx int64 [nq, k], score f64 [nq, k]).
    Returns (idx, score) for all queries on every rank (gather_result) or for this rank's query block.
    around_exchange("begin" | "end") is called right before / after the one collective of the path (bench.py records
    stream events there).  Without an initialised process group this is the single-GPU path (no collective);
    force_exchange runs the collective in a one-rank group too (all_gather_rows(force=True))."""
    rank, world = rank_and_world(group)
    c_lo, c_hi = shard_range(n_cells, rank, world)
    q_lo, q_hi = shard_range(n_queries, rank, world)
    cells_local = encode_local_cells(c_lo, c_hi)
    queries_local = encode_local_queries(q_lo, q_hi)
    exchanging = world > 1 or (force_exchange and dist.is_available() and dist.is_initialized())
    if around_exchange is not None and exchanging:
        around_exchange("begin")
    cells_all = all_gather_rows(cells_local, n_cells, group, force=force_exchange)   # the one exchange step of the path
    if around_exchange is not None and exchanging:
        around_exchange("end")
    idx, score = topk_fn(queries_local, cells_all, k)
    if not gather_result or world == 1:
        return idx, score
    idx_all = all_gather_rows(idx, n_queries, group)
    score_all = all_gather_rows(score, n_queries, group)
    return idx_all, score_all

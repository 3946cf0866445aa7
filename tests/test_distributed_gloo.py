"""world_size-2 (and 3) gloo tests of the multi-GPU layout on CPU: cell/query sharding, the single all-gather with
uneven shards, and index bookkeeping -- bit-for-bit against the single-process result.  The compute steps are
injected from the oracle (tests only), as text2pos_amd.distributed is compute-agnostic."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _embeddings(n, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.nn.functional.normalize(torch.randn(n, 256, generator=g), dim=-1)
    if n > 20:
        x[11] = x[3]          # duplicates straddling shard boundaries -> tie handling must survive sharding
        x[n - 1] = x[3]
    return x


def _topk(q, c, k):
    from oracle.model import retrieve_topk_f64
    idx, sc = retrieve_topk_f64(c.numpy(), q.numpy(), k)
    return torch.from_numpy(idx), torch.from_numpy(sc)


def _worker(rank, world, port, n_cells, n_q, k, out_dir):
    import text2pos_amd  # noqa: F401
    from text2pos_amd import distributed as TD
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cells, queries = _embeddings(n_cells, 1), _embeddings(n_q, 2)
    calls = []

    def enc_c(lo, hi):
        calls.append(("c", lo, hi))
        return cells[lo:hi].clone()

    def enc_q(lo, hi):
        calls.append(("q", lo, hi))
        return queries[lo:hi].clone()

    idx, sc = TD.sharded_retrieval(enc_c, enc_q, _topk, n_cells, n_q, k)
    assert calls == [("c", *TD.shard_range(n_cells, rank, world)), ("q", *TD.shard_range(n_q, rank, world))]
    torch.save((idx, sc), os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_cells,n_q", [(2, 101, 17), (2, 64, 8), (3, 50, 7)])
def test_sharded_retrieval_equals_single_process(tmp_path, world, n_cells, n_q):
    k = 10
    mp.spawn(_worker, args=(world, _free_port(), n_cells, n_q, k, str(tmp_path)), nprocs=world, join=True)
    want_idx, want_sc = _topk(_embeddings(n_q, 2), _embeddings(n_cells, 1), k)
    for r in range(world):
        idx, sc = torch.load(os.path.join(str(tmp_path), f"r{r}.pt"))
        assert torch.equal(idx, want_idx), f"rank {r}"
        assert torch.equal(sc, want_sc), f"rank {r}"


def test_all_gather_rows_single_process_is_identity():
    import text2pos_amd  # noqa: F401
    from text2pos_amd import distributed as TD
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(_free_port())
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        x = torch.randn(5, 4)
        assert TD.all_gather_rows(x, 5) is x
    finally:
        dist.destroy_process_group()

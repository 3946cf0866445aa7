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


# ---- BASELINE configs[2] shard shapes: 100,000 (and 100,001) cells over 8 ranks, 10,000 queries -----------------------
_DIM3 = 32     # the sharding / gather / index logic does not depend on D; 32 keeps the float64 products small on CPU


def _embeddings_c3(n, seed, world):
    from text2pos_amd import distributed as TD
    g = torch.Generator().manual_seed(seed)
    x = torch.nn.functional.normalize(torch.randn(n, _DIM3, generator=g), dim=-1)
    # exact duplicates sitting on both sides of every shard edge (and of a far-away row): the tie order must survive the
    # partition, the padding of uneven shards and the gather
    for r in range(1, world):
        edge = TD.shard_range(n, r, world)[0]
        x[edge - 1] = x[5 * r]
        x[edge] = x[5 * r]
    return x


def _topk_blocked(q, c, k):
    """float64 scores + argsort(kind="stable") semantics (ties -> lower index) without sorting 100k scores per query:
    the k-th largest score bounds a small candidate set, which is ordered exactly."""
    c64, idx, sc = c.double().numpy(), np.zeros((q.shape[0], k), np.int64), np.zeros((q.shape[0], k), np.float64)
    for lo in range(0, q.shape[0], 250):
        s = q[lo: lo + 250].double().numpy() @ c64.T
        kth = -np.partition(-s, k - 1, axis=1)[:, k - 1]
        for i in range(s.shape[0]):
            cand = np.flatnonzero(s[i] >= kth[i])
            order = cand[np.lexsort((cand, -s[i, cand]))][:k]
            idx[lo + i], sc[lo + i] = order, s[i, order]
    return torch.from_numpy(idx), torch.from_numpy(sc)


def _worker_c3(rank, world, port, n_cells, n_q, k, out_dir):
    import text2pos_amd  # noqa: F401
    from text2pos_amd import distributed as TD
    torch.set_num_threads(1)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cells, queries = _embeddings_c3(n_cells, 1, world), _embeddings_c3(n_q, 2, world)
    idx, sc = TD.sharded_retrieval(lambda lo, hi: cells[lo:hi].clone(), lambda lo, hi: queries[lo:hi].clone(), _topk_blocked,
                                   n_cells, n_q, k, gather_result=False)
    q_lo, q_hi = TD.shard_range(n_q, rank, world)
    assert idx.shape == (q_hi - q_lo, k)
    torch.save((idx, sc), os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_cells", [100_000, 100_001])
def test_sharded_retrieval_world8_config3_shapes(tmp_path, n_cells):
    """8 ranks x 12,500 (or 12,501 / 12,500: uneven, padded) cells, one all-gather of the shards, 10,000 queries in blocks
    of 1,250: every rank's block of the top-10 equals the single-process result bit for bit, duplicates across the shard
    edges included."""
    from text2pos_amd import distributed as TD
    world, n_q, k = 8, 10_000, 10
    mp.spawn(_worker_c3, args=(world, _free_port(), n_cells, n_q, k, str(tmp_path)), nprocs=world, join=True)
    want_idx, want_sc = _topk_blocked(_embeddings_c3(n_q, 2, world), _embeddings_c3(n_cells, 1, world), k)
    for r in range(world):
        lo, hi = TD.shard_range(n_q, r, world)
        idx, sc = torch.load(os.path.join(str(tmp_path), f"r{r}.pt"))
        assert torch.equal(idx, want_idx[lo:hi]), f"rank {r}"
        assert torch.equal(sc, want_sc[lo:hi]), f"rank {r}"
    # the duplicated rows do compete: a query equal to the duplicated vector ranks its copies first, in index order
    cells = _embeddings_c3(n_cells, 1, world)
    probe_idx, _ = _topk_blocked(cells[5:6], cells, 3)
    assert probe_idx[0].tolist() == sorted(probe_idx[0].tolist()) and probe_idx[0, 0].item() == 5


def test_all_gather_rows_single_process_is_identity():
    import text2pos_amd  # noqa: F401
    from text2pos_amd import distributed as TD
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(_free_port())
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        x = torch.randn(5, 4)
        assert TD.all_gather_rows(x, 5) is x
        # force=True (bench.py --force-exchange): the collective itself runs in the one-rank group and returns a copy
        y = TD.all_gather_rows(x, 5, force=True)
        assert y is not x and torch.equal(y, x)
        marks = []
        cells, queries = _embeddings(40, 1), _embeddings(6, 2)
        idx, sc = TD.sharded_retrieval(lambda lo, hi: cells[lo:hi], lambda lo, hi: queries[lo:hi], _topk, 40, 6, 5,
                                       around_exchange=marks.append, force_exchange=True)
        assert marks == ["begin", "end"]
        want_idx, want_sc = _topk(queries, cells, 5)
        assert torch.equal(idx, want_idx) and torch.equal(sc, want_sc)
        marks.clear()
        TD.sharded_retrieval(lambda lo, hi: cells[lo:hi], lambda lo, hi: queries[lo:hi], _topk, 40, 6, 5, around_exchange=marks.append)
        assert marks == []              # world 1 without the switch: no collective, no events
    finally:
        dist.destroy_process_group()


# ---- BASELINE configs[4]: the evaluation pipeline (coarse retrieval + fine localisation) sharded over ranks -------------
def _toy_scene(seed=11, n_cells=13, n_poses=21):
    """A small synthetic scene in the package's own data model (odd sizes: uneven blocks on 2 ranks)."""
    from text2pos_amd import data as D, synthetic as S
    rng = np.random.default_rng(seed)
    cells, poses = [], []
    dirs = ["north", "south", "east", "west", "on-top"]
    for i in range(n_cells):
        objs = []
        for j in range(int(rng.integers(6, 12))):
            c = rng.random(3) * np.array([1.0, 1.0, 0.3])
            n = int(rng.integers(30, 90))
            objs.append(D.Object3d(j, 1000 * i + j, c + 0.05 * rng.standard_normal((n, 3)),
                                   np.repeat(np.clip(rng.random((1, 3)), 0, 1), n, axis=0),
                                   S.LABELS[int(rng.integers(0, len(S.LABELS)))]))
        x, y = 30.0 * (i % 4), 30.0 * (i // 4)
        cells.append(D.Cell(i, "toy1", objs, 30.0, np.array([x, y, 0.0, x + 30.0, y + 30.0, 10.0])))
    for q in range(n_poses):
        c = cells[int(rng.integers(0, n_cells))]
        descs = [D.DescriptionBestCell(dirs[int(rng.integers(0, 5))], o.get_color_text(), o.label, o.id, True)
                 for o in [c.objects[int(k)] for k in rng.integers(0, len(c.objects), 6)]]
        poses.append(D.Pose(rng.random(3), c.bbox_w[0:3] + rng.random(3) * 30.0, c.id, "toy1", descs))
    return cells, poses


class _StubCoarse:
    """CPU stand-in with the CellRetrievalNetwork surface run_coarse uses; its outputs depend on the SAMPLED points, so a
    rank-dependent FixedPoints draw would show up as a different retrieval."""
    embed_dim, device = 16, torch.device("cpu")

    def encode_objects(self, objects, object_points):
        rows = []
        for objs, pts in zip(objects, object_points):
            pos = pts.pos.double().view(len(objs), -1, 3)
            f = torch.cat([pos.mean(1).sum(0), pos.abs().amax(1).sum(0), pts.x.double().mean(0),
                           torch.tensor([len(objs) / 10.0], dtype=torch.float64)])
            rows.append(torch.cat([torch.sin(f * (i + 1)) for i in range(2)])[: self.embed_dim])
        return torch.nn.functional.normalize(torch.stack(rows).float(), dim=-1)

    def encode_text(self, texts):
        rows = [torch.tensor([(sum(map(ord, t)) * (i + 3)) % 97 / 97.0 - 0.5 for i in range(self.embed_dim)]) for t in texts]
        return torch.nn.functional.normalize(torch.stack(rows).float(), dim=-1)


class _StubFine:
    device = torch.device("cpu")

    def __call__(self, objects, hints, object_points):
        from types import SimpleNamespace
        b, pad = len(objects), len(objects[0])
        key = torch.stack([p.pos.double().sum() for p in object_points])            # depends on the sampled points
        m0 = ((torch.arange(pad)[None] + (key * 1000).long()[:, None]) % 9) - 2      # some -1 / -2 = unmatched
        m0 = torch.where(m0 < 6, m0, torch.full_like(m0, -1)).clamp(min=-1)
        off = torch.sin(key)[:, None, None] * torch.ones(b, 6, 2, dtype=torch.float64) * 0.1
        return SimpleNamespace(matches0=m0, offsets=off)


def _pipeline_tables(group_world):
    import text2pos_amd  # noqa: F401
    from text2pos_amd import io as IO, pipeline as PL
    cells, poses = _toy_scene()
    sc = IO.Scenes(cells, poses)
    return PL.evaluate(_StubCoarse(), _StubFine(), sc, PL.PerCellTransform(64, 3), top_k=(1, 3, 5), threshs=(5, 10, 15),
                       pad_size=8, queries_per_call=4, topk_fn=_topk)


def _worker_pipeline(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = _pipeline_tables(world)
    torch.save(out, os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_pipeline_equals_single_process(tmp_path, world):
    """pipeline.evaluate under a process group (cells + queries of the coarse stage through distributed.sharded_retrieval,
    queries of the fine stage in blocks) returns, on every rank, exactly what the single-process run returns: retrieval lists,
    hit / close-by tables, coarse and fine localisation tables."""
    want = _pipeline_tables(1)
    mp.spawn(_worker_pipeline, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert len(want["retrievals"]) == 21 and all(len(r) == 5 for r in want["retrievals"])
    for r in range(world):
        got = torch.load(os.path.join(str(tmp_path), f"r{r}.pt"), weights_only=False)
        assert got["retrievals"] == want["retrievals"], f"rank {r}"
        for name in ("hit", "close", "localisation", "fine_mean", "fine_offset", "fine_mean_conf"):
            assert got[name] == want[name], (r, name)

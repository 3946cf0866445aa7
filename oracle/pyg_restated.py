"""oracle/pyg_restated.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  "parity unpinned".

CPU restatement of the torch_geometric operators the reference calls on the hot
path.  torch_geometric / torch_cluster are third-party dependencies, un-pinned in
/root/reference/requirements.txt:9-10 and absent from this image, and the
reference holds no test for them, so these follow the published upstream
algorithms (PyG 1.7/2.0 era, torch_cluster 1.5) with the nondeterministic
choices pinned as documented in oracle/primitives.c.

Call sites restated:
  fps               models/pointcloud/pointnet2.py:26
  radius            models/pointcloud/pointnet2.py:28-30
  PointConv         models/pointcloud/pointnet2.py:23,35
  global_max_pool   models/pointcloud/pointnet2.py:48 ; models/cell_retrieval.py:98
  global_mean_pool  models/cell_retrieval.py:102
  DynamicEdgeConv   models/cell_retrieval.py:46-48,51-53,97
  Data/Batch        dataloading/kitti360pose/utils.py:99-109
  FixedPoints/NormalizeScale  training/coarse.py:189-199 ; evaluation/pipeline.py:290-293

The signatures mirror torch_geometric so that tests/golden/make_golden.py can bind this
module as the `torch_geometric.nn` the reference's glue code imports.
"""
import ctypes
import math

import numpy as np
import torch
import torch.nn as nn

from . import lib as _lib


def _segments(batch: torch.Tensor):
    """(ptr) of a sorted batch vector."""
    batch = batch.cpu()
    assert bool((batch[1:] >= batch[:-1]).all()), "batch vector must be sorted"
    nb = int(batch.max().item()) + 1 if batch.numel() else 0
    counts = torch.bincount(batch, minlength=nb)
    ptr = torch.zeros(nb + 1, dtype=torch.long)
    ptr[1:] = torch.cumsum(counts, 0)
    return ptr


def _p(a: np.ndarray, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def fps(x: torch.Tensor, batch: torch.Tensor = None, ratio: float = 0.5, random_start: bool = False):
    """torch_cluster.fps: per sub-graph farthest point sampling, ceil(ratio*N) samples.
    Pinned: random_start=False (upstream default True picks a random first point)."""
    assert not random_start, "oracle pins random_start=False"
    if batch is None:
        batch = torch.zeros(x.shape[0], dtype=torch.long)
    ptr = _segments(batch)
    out = []
    xn = np.ascontiguousarray(x.detach().cpu().numpy().astype(np.float32))
    L = _lib()
    for b in range(len(ptr) - 1):
        lo, hi = int(ptr[b]), int(ptr[b + 1])
        n = hi - lo
        m = int(math.ceil(ratio * n))
        seg = np.ascontiguousarray(xn[lo:hi])
        idx = np.zeros(m, dtype=np.int32)
        L.t2p_oracle_fps(_p(seg, ctypes.c_float), ctypes.c_int64(1), ctypes.c_int32(n), ctypes.c_int32(m),
                         _p(idx, ctypes.c_int32))
        out.append(torch.from_numpy(idx.astype(np.int64)) + lo)
    return torch.cat(out) if out else torch.zeros(0, dtype=torch.long)


def radius(x, y, r, batch_x=None, batch_y=None, max_num_neighbors: int = 32):
    """torch_cluster.radius: returns [2, E] = (row = y index, col = x index); for each y the first
    <= max_num_neighbors x of the same sub-graph (ascending x index) with |x - y|^2 < r^2."""
    if batch_x is None:
        batch_x = torch.zeros(x.shape[0], dtype=torch.long)
    if batch_y is None:
        batch_y = torch.zeros(y.shape[0], dtype=torch.long)
    px, py = _segments(batch_x), _segments(batch_y)
    xn = x.detach().cpu().numpy().astype(np.float32)
    yn = y.detach().cpu().numpy().astype(np.float32)
    r2 = np.float32(np.float32(r) * np.float32(r))
    rows, cols = [], []
    for b in range(len(py) - 1):
        xs = xn[int(px[b]):int(px[b + 1])]
        for j in range(int(py[b]), int(py[b + 1])):
            d = xs - yn[j]
            d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
            hit = np.nonzero(d2 < r2)[0][:max_num_neighbors]
            rows.append(np.full(hit.shape[0], j, dtype=np.int64))
            cols.append(hit.astype(np.int64) + int(px[b]))
    row = torch.from_numpy(np.concatenate(rows)) if rows else torch.zeros(0, dtype=torch.long)
    col = torch.from_numpy(np.concatenate(cols)) if cols else torch.zeros(0, dtype=torch.long)
    return torch.stack([row, col], dim=0)


def knn(x, y, k, batch_x=None, batch_y=None):
    """torch_cluster.knn for x is y (the only use on the path): [2, E] = (row = y index, col = x index)."""
    assert x is y or torch.equal(x, y)
    if batch_x is None:
        batch_x = torch.zeros(x.shape[0], dtype=torch.long)
    ptr = _segments(batch_x).numpy().astype(np.int32)
    xn = np.ascontiguousarray(x.detach().cpu().numpy().astype(np.float32))
    n, dim = xn.shape
    out = np.zeros((n, k), dtype=np.int32)
    _lib().t2p_oracle_knn(_p(xn, ctypes.c_float), _p(ptr, ctypes.c_int32), ctypes.c_int32(len(ptr) - 1),
                          ctypes.c_int32(dim), ctypes.c_int32(k), _p(out, ctypes.c_int32))
    row = np.repeat(np.arange(n, dtype=np.int64), k)
    col = out.reshape(-1).astype(np.int64)
    keep = col >= 0
    return torch.from_numpy(np.stack([row[keep], col[keep]], 0))


def _scatter(msg, index, n_out, aggr):
    c = msg.shape[1]
    if aggr == "max":
        out = torch.full((n_out, c), float("-inf"), dtype=msg.dtype)
        out = out.scatter_reduce(0, index[:, None].expand(-1, c), msg, reduce="amax", include_self=True)
        # PyG fills empty segments with 0 (out of place: the training-mode tests differentiate through this)
        return torch.where(out == float("-inf"), torch.zeros_like(out), out)
    if aggr == "mean":
        out = torch.zeros((n_out, c), dtype=msg.dtype)
        out.index_add_(0, index, msg)
        cnt = torch.bincount(index, minlength=n_out).clamp(min=1).to(msg.dtype)
        return out / cnt[:, None]
    raise ValueError(aggr)


def global_max_pool(x, batch):
    return _scatter(x, batch.cpu(), int(batch.max().item()) + 1, "max")


def global_mean_pool(x, batch):
    return _scatter(x, batch.cpu(), int(batch.max().item()) + 1, "mean")


class PointConv(nn.Module):
    """torch_geometric.nn.PointConv (a.k.a. PointNetConv), aggr='max'.

    forward(x, (pos_dense, pos_centroid), edge_index[2,E] = (source dense j, target centroid i)):
      add_self_loops=True (upstream default, which the reference uses): remove edges with j == i
      (index equality), then append (i, i) for i < min(N_dense, N_centroid)  -- in this bipartite call
      that links *dense point i* to *centroid i*;
      message = local_nn(cat([x_j, pos_j - pos_i])); out_i = max over incoming edges.
    """

    def __init__(self, local_nn=None, global_nn=None, add_self_loops: bool = True):
        super().__init__()
        self.local_nn = local_nn
        self.global_nn = global_nn
        self.add_self_loops = add_self_loops

    def forward(self, x, pos, edge_index):
        if not isinstance(x, tuple):
            x = (x, None)
        if isinstance(pos, torch.Tensor):
            pos = (pos, pos)
        src, dst = edge_index[0], edge_index[1]
        if self.add_self_loops:
            keep = src != dst
            src, dst = src[keep], dst[keep]
            n = min(pos[0].shape[0], pos[1].shape[0])
            loop = torch.arange(n, dtype=torch.long)
            src, dst = torch.cat([src, loop]), torch.cat([dst, loop])
        msg = pos[0][src] - pos[1][dst]
        if x[0] is not None:
            msg = torch.cat([x[0][src], msg], dim=1)
        if self.local_nn is not None:
            msg = self.local_nn(msg)
        out = _scatter(msg, dst, pos[1].shape[0], "max")
        if self.global_nn is not None:
            out = self.global_nn(out)
        return out


class DynamicEdgeConv(nn.Module):
    """torch_geometric.nn.DynamicEdgeConv: knn graph (k, self included) inside each batch segment,
    message = nn(cat([x_i, x_j - x_i])), aggregated (max | mean) over the neighbours j of i."""

    def __init__(self, nn, k, aggr="max"):
        super().__init__()
        self.nn = nn
        self.k = k
        self.aggr = aggr

    def forward(self, x, batch=None):
        e = knn(x, x, self.k, batch, batch)  # (row = target i, col = source j)
        i, j = e[0], e[1]
        msg = self.nn(torch.cat([x[i], x[j] - x[i]], dim=-1))
        return _scatter(msg, i, x.shape[0], self.aggr)


# ---- data containers / transforms (torch_geometric.data, torch_geometric.transforms) -------------------------
class Data:
    def __init__(self, x=None, pos=None, batch=None):
        self.x, self.pos, self.batch = x, pos, batch

    @property
    def num_nodes(self):
        return self.pos.shape[0]

    def to(self, device):
        return self


class Batch(Data):
    @staticmethod
    def from_data_list(data_list):
        x = torch.cat([d.x for d in data_list], 0)
        pos = torch.cat([d.pos for d in data_list], 0)
        batch = torch.cat([torch.full((d.pos.shape[0],), i, dtype=torch.long) for i, d in enumerate(data_list)])
        return Batch(x=x, pos=pos, batch=batch)


class FixedPoints:
    """T.FixedPoints(num, replace=True): random choice with replacement (seeded generator here)."""

    def __init__(self, num, replace=True, generator: np.random.Generator = None):
        self.num, self.replace = num, replace
        self.gen = generator if generator is not None else np.random.default_rng(0)

    def __call__(self, data):
        n = data.pos.shape[0]
        choice = torch.from_numpy(self.gen.choice(n, self.num, replace=True))
        data.x, data.pos = data.x[choice], data.pos[choice]
        return data


class NormalizeScale:
    """T.NormalizeScale: centre to the mean, scale by 0.999999 / max|pos|."""

    def __call__(self, data):
        data.pos = data.pos - data.pos.mean(dim=-2, keepdim=True)
        scale = (1 / data.pos.abs().max()) * 0.999999
        data.pos = data.pos * scale
        return data


class RotateZ:
    """T.RandomRotate(degrees, axis=2) with the drawn angle made explicit (cos, sin given as fp32 like the matrix tensor
    PyG builds): pos <- pos @ [[c, s, 0], [-s, c, 0], [0, 0, 1]] (LinearTransformation of PyG 1.7 - 2.0; unverifiable
    here -- later releases use the transposed matrix, the same distribution for the symmetric range the reference uses)."""

    def __init__(self, cos: float, sin: float):
        self.matrix = torch.tensor([[cos, sin, 0.0], [-sin, cos, 0.0], [0.0, 0.0, 1.0]], dtype=torch.float)

    def __call__(self, data):
        data.pos = torch.matmul(data.pos, self.matrix)
        return data


class Compose:
    def __init__(self, ts):
        self.ts = ts

    def __call__(self, data):
        for t in self.ts:
            data = t(data)
        return data

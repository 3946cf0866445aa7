"""A scene resident in HBM: the input side of evaluation/pipeline.py on the GPU.

The reference's evaluation builds every model input on the host, per batch: `Kitti360CoarseDatasetMulti.__getitem__` /
`Kitti360TopKDataset.__getitem__` run `batch_object_points` (dataloading/kitti360pose/utils.py:89-110: one
Data -> T.FixedPoints -> T.NormalizeScale chain per object) for every cell of the database (evaluation/pipeline.py:303-308) and
again for every (query, retrieved cell) pair of the fine stage (dataloading/kitti360pose/eval.py:117-189), and the model then takes
a float64 NumPy mean over every object's raw points (models/object_encoder.py:121-131).  Here the raw points of all objects are
converted to fp32 and uploaded ONCE (one pass of the C helper csrc/host_ext.c over the float64 arrays, which also yields the
per-object means - exactly the reference's float32 values); after that a batch of cells, or of (query, candidate) samples, is
packed by one kernel launch (t2p_pack_scene_objects) from a list of object ids and sampling keys.
"""
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import data as D
from . import ops


class DeviceScene:
    """raw_xyz / raw_rgb [R, 3] fp32, obj_ptr int32 [M + 1], center / color [M, 3] fp32 on `device`; on the host: `rows` (points
    per object), `cell_ptr` (objects per cell, CSR over the scene's objects, int64 [n_cells + 1]), `center64` (float64 object
    centres: the fine stage's pose estimate adds offsets to them, models/superglue_matcher.py:139-161), `labels`.
    n_pad > 0 appends that many padding objects (Object3d.create_padding: dataloading/kitti360pose/eval.py:147-149 fills a
    sample's object list up to pad_size with them); their scene ids are `pad_ids`.  The padding objects' eight points (within 1 mm of
    the origin) are drawn from `pad_seed`, not from the process-global np.random the reference uses: every rank of a process group
    builds its own scene, and the fine stage must see the same padding on each of them (and in the single-process run)."""

    def __init__(self, cells: Sequence, device, n_pad: int = 0, threads: Optional[int] = None, pad_seed: int = 0):
        self.device = torch.device(device)
        groups = [c.objects if isinstance(c.objects, list) else list(c.objects) for c in cells]
        counts = np.fromiter((len(g) for g in groups), dtype=np.int64, count=len(groups))
        if len(groups) and counts.min() < 1:
            raise RuntimeError(f"DeviceScene: cell {int(np.argmin(counts))} has no objects")
        self.cell_ptr = np.zeros(len(groups) + 1, dtype=np.int64)
        np.cumsum(counts, out=self.cell_ptr[1:])
        self.cell_ids = [getattr(c, "id", i) for i, c in enumerate(cells)]
        self.row_of = {cid: i for i, cid in enumerate(self.cell_ids)}
        n_real = int(self.cell_ptr[-1])
        if n_pad > 0:
            pad_rng = np.random.default_rng([int(pad_seed), 0x7061_6400])
            groups = groups + [[D.Object3d.create_padding(rng=pad_rng) for _ in range(n_pad)]]
        self.pad_ids = np.arange(n_real, n_real + max(n_pad, 0), dtype=np.int64)
        flat = [o for g in groups for o in g]
        self.n_objects = len(flat)
        self.labels = [o.label for o in flat]
        self._flat = flat     # (kept for the --class_embed / --color_embed index tables, built on demand)
        raw_xyz, raw_rgb, rows, c32, k32, c64 = _flatten(groups, flat, threads)
        if int(rows.sum()) >= 2 ** 31:
            raise RuntimeError("DeviceScene: more than 2^31 raw points; build one scene per block of cells")
        self.rows = rows
        obj_ptr = np.zeros(len(flat) + 1, dtype=np.int32)
        np.cumsum(rows, out=obj_ptr[1:])
        self.center64 = c64
        to = lambda a: torch.from_numpy(a).to(self.device)
        self.raw_xyz, self.raw_rgb, self.obj_ptr = to(raw_xyz), to(raw_rgb), to(obj_ptr)
        self.center, self.color = to(c32), to(k32)
        self._idx = {}
        self._padded = {}

    @property
    def n_cells(self) -> int:
        return self.cell_ptr.shape[0] - 1

    def feature_indices(self, model):
        """(class_idx, color_idx) int32 device tables over the scene's objects for the --class_embed / --color_embed ablations
        (models/object_encoder.py:74-84), else (None, None)."""
        a, oe = model.args, model.object_encoder
        out = []
        for flag, name, fn in (("class_embed", "class", lambda o: oe.known_classes.get(o.label, 0)),
                               ("color_embed", "color", lambda o: oe.known_colors[o.get_color_text()])):
            if not getattr(a, flag, False):
                out.append(None)
                continue
            key = (name, id(oe))
            if key not in self._idx:
                self._idx[key] = torch.tensor([fn(o) for o in self._flat], dtype=torch.int32).to(self.device)
            out.append(self._idx[key])
        return tuple(out)

    def padded_object_ids(self, pad_size: int) -> np.ndarray:
        """int64 [n_cells, pad_size]: the first pad_size objects of every cell, missing slots filled with the scene's padding
        objects (slot j takes pad_ids[j]) - the object list of dataloading/kitti360pose/eval.py:141-149."""
        if pad_size not in self._padded:
            if len(self.pad_ids) < pad_size:
                raise RuntimeError(f"DeviceScene was built with n_pad={len(self.pad_ids)} < pad_size={pad_size}")
            slot = np.arange(pad_size, dtype=np.int64)[None, :]
            n = (self.cell_ptr[1:] - self.cell_ptr[:-1])[:, None]
            self._padded[pad_size] = np.where(slot < n, self.cell_ptr[:-1, None] + slot, self.pad_ids[:pad_size][None, :])
        return self._padded[pad_size]

    def pack(self, obj_ids: np.ndarray, keys: np.ndarray, n_pts: int = 256, want_rgb: bool = True, want_idx: bool = False):
        """Packed encoder inputs (xyz, rgb | None, center, mean_rgb[, sample_idx]) of the object slots (obj_ids int [n], keys
        uint64 [n]); one kernel launch on the current stream."""
        ids = torch.from_numpy(np.ascontiguousarray(obj_ids, dtype=np.int32)).to(self.device, non_blocking=True)
        k = torch.from_numpy(np.ascontiguousarray(keys, dtype=np.uint64).view(np.int64)).to(self.device, non_blocking=True)
        return ops.pack_scene_objects(self.raw_xyz, self.raw_rgb, self.obj_ptr, ids, k, self.center, self.color, n_pts,
                                      want_rgb, want_idx)

    def pack_cells(self, transform, lo: int, hi: int, cell_offset: int = 0, **kw):
        """Cells [lo, hi) of this scene under `transform` (pipeline.PerCellTransform; global index of scene cell i =
        cell_offset + i).  Returns (packed tuple, cell_ptr int32 [hi - lo + 1], obj_ids)."""
        o0, o1 = int(self.cell_ptr[lo]), int(self.cell_ptr[hi])
        cp = (self.cell_ptr[lo: hi + 1] - o0).astype(np.int32)
        n = (cp[1:] - cp[:-1]).astype(np.int64)
        cell_of = np.repeat(np.arange(lo, hi, dtype=np.int64), n)
        slot = np.arange(o1 - o0, dtype=np.int64) - np.repeat(cp[:-1].astype(np.int64), n)
        ids = np.arange(o0, o1, dtype=np.int64)
        return self.pack(ids, transform.keys(cell_of + cell_offset, slot), transform.n_pts, **kw), cp, ids


def _flatten(groups: List[list], flat: list, threads: Optional[int]):
    """(raw_xyz f32 [R, 3], raw_rgb f32 [R, 3], rows int64 [M], centre f32 [M, 3], colour f32 [M, 3], centre f64 [M, 3]).
    The float32 means are the reference's (float32 of NumPy's float64 mean), bit for bit: data._means_from_sums."""
    ext = D.host_ext()
    m = len(flat)
    if ext is not None and m and all(D._plain_objects(g) for g in groups):
        rows = np.empty(m, dtype=np.int64)
        if ext.point_rows(groups, rows) == m:
            total = int(rows.sum())
            xyz, rgb = np.empty((total, 3), dtype=np.float32), np.empty((total, 3), dtype=np.float32)
            sums, asums = np.empty((2, m, 3), dtype=np.float64), np.empty((2, m, 3), dtype=np.float64)
            rows2 = np.empty((2, m), dtype=np.int64)
            if ext.object_sums(groups, sums, asums, rows2, int(threads or D.host_threads()), xyz, rgb) == m:
                c32 = D._means_from_sums(sums[0], asums[0], rows2[0], lambda i: flat[i].xyz)
                k32 = D._means_from_sums(sums[1], asums[1], rows2[1], lambda i: flat[i].rgb)
                return xyz, rgb, rows, c32, k32, sums[0] / rows[:, None].astype(np.float64)
    # NumPy route: objects whose arrays are not C-contiguous float64 [m, 3], or the helper is not built
    xyz = np.concatenate([np.asarray(o.xyz, dtype=np.float32).reshape(-1, 3) for o in flat], 0) if m else np.zeros((0, 3), np.float32)
    rgb = np.concatenate([np.asarray(o.rgb, dtype=np.float32).reshape(-1, 3) for o in flat], 0) if m else np.zeros((0, 3), np.float32)
    rows = np.array([len(o.xyz) for o in flat], dtype=np.int64)
    if any(len(o.rgb) != len(o.xyz) for o in flat):
        raise RuntimeError("DeviceScene: an object's colours and points differ in number")
    if m and rows.min() < 1:
        raise RuntimeError("DeviceScene: an object without points cannot be resampled")
    c64 = np.stack([np.asarray(o.get_center(), dtype=np.float64) for o in flat]) if m else np.zeros((0, 3))
    k64 = np.stack([np.asarray(o.get_color_rgb(), dtype=np.float64) for o in flat]) if m else np.zeros((0, 3))
    return xyz, rgb, rows, c64.astype(np.float32), k64.astype(np.float32), c64

"""Host-side input types of the path (mirrors of the reference's data model) and their packing into the flat
structure-of-arrays layout the kernels read.

Reference: Object3d / Cell  datapreparation/kitti360pose/imports.py:8-41,221-247;
batch_object_points  dataloading/kitti360pose/utils.py:89-110 (per object Data(x=rgb, pos=xyz) -> transform ->
Batch.from_data_list); transforms T.FixedPoints / T.NormalizeScale used at training/coarse.py:189-199 and
evaluation/pipeline.py:290-293.  torch_geometric is not required: any object with `.x`, `.pos`, `.batch` works.
"""
from typing import List, Sequence

import numpy as np
import torch

COLORS = np.array([[47.2579917, 49.75368454, 42.4153065], [136.32696657, 136.95241796, 126.02741229],
                   [87.49822126, 91.69058836, 80.14558512], [213.91030679, 216.25033052, 207.24611073],
                   [110.39218852, 112.91977458, 103.68638249], [27.47505158, 28.43996795, 25.16840296],
                   [66.65951839, 70.22342483, 60.20395996], [171.00852191, 170.05737735, 155.00130334]]) / 255.0
# class names of KITTI360Pose in CLASS_TO_INDEX order (datapreparation/kitti360pose/utils.py:48-71)
KNOWN_CLASSES = ["building", "pole", "traffic light", "traffic sign", "garage", "stop", "smallpole", "lamp", "trash bin",
                 "vending machine", "box", "road", "sidewalk", "parking", "wall", "fence", "guard rail", "bridge", "tunnel",
                 "vegetation", "terrain", "pad"]
COLOR_NAMES = ["dark-green", "gray", "gray-green", "bright-gray", "gray", "black", "green", "beige"]
CLASS_NAMES = ["building", "pole", "traffic light", "traffic sign", "garage", "stop", "smallpole", "lamp", "trash bin",
               "vending machine", "box", "road", "sidewalk", "parking", "wall", "fence", "guard rail", "bridge", "tunnel",
               "vegetation", "terrain", "pad"]


class Object3d:
    def __init__(self, id: int, instance_id: int, xyz: np.ndarray, rgb: np.ndarray, label: str):
        self.id, self.instance_id, self.xyz, self.rgb, self.label = id, instance_id, xyz, rgb, label

    def get_color_rgb(self):
        return np.mean(self.rgb, axis=0)

    def get_center(self):
        return np.mean(self.xyz, axis=0)

    def get_color_text(self):
        return COLOR_NAMES[int(np.argmin(np.linalg.norm(np.mean(self.rgb, axis=0) - COLORS, axis=1)))]

    def __repr__(self):
        return f"Object3d: {self.label}"

    @classmethod
    def create_padding(cls, rng=np.random):
        """Padding object of the fine stage (datapreparation/kitti360pose/imports.py:74-83): 8 points within 1 mm of
        the origin, black, label "pad"."""
        return cls(-1, -1, rng.rand(8, 3) * 0.001 if rng is np.random else rng.random((8, 3)) * 0.001,
                   np.zeros((8, 3)), "pad")


class Cell:
    def __init__(self, idx, scene_name, objects: List[Object3d], cell_size, bbox_w):
        self.scene_name, self.objects, self.cell_size, self.bbox_w = scene_name, objects, cell_size, np.asarray(bbox_w)
        self.id = f"{scene_name}_{idx:05.0f}"   # "00XX_XXXXX" (datapreparation/kitti360pose/imports.py:189)

    def get_center(self):
        return 1 / 2 * (self.bbox_w[0:3] + self.bbox_w[3:6])


class Pose:
    def __init__(self, pose_in_cell, pose_w, cell_id, scene_name, descriptions=None, described_by=None):
        self.pose, self.pose_w, self.cell_id = pose_in_cell, np.asarray(pose_w), cell_id
        self.scene_name, self.descriptions, self.described_by = scene_name, descriptions, described_by

    def __repr__(self):
        return f"Pose at {self.pose_w} in {self.cell_id}"


class DescriptionPoseCell:
    """Attribute container of datapreparation/kitti360pose/imports.py:86-115 (unpickled as is)."""

    def __repr__(self):
        return f"Pose is {self.direction} of a {self.object_color_text} {self.object_label}"


class DescriptionBestCell:
    """One hint of a pose in the context of its best cell (datapreparation/kitti360pose/imports.py:119-176): the
    coarse / fine stages read `direction`, `object_color_text`, `object_label` (hint sentence) and `object_id`,
    `best_offset_center`, `is_matched` (training only)."""

    def __init__(self, direction=None, object_color_text=None, object_label=None, object_id=-1, is_matched=False):
        self.direction, self.object_color_text, self.object_label = direction, object_color_text, object_label
        self.object_id, self.is_matched = object_id, is_matched

    def __repr__(self):
        return f"Pose is {self.direction} of a {self.object_color_text} {self.object_label}"


class Data:
    def __init__(self, x=None, pos=None, batch=None):
        self.x, self.pos, self.batch = x, pos, batch

    @property
    def num_nodes(self):
        return self.pos.shape[0]

    def to(self, device):
        self.x, self.pos = self.x.to(device), self.pos.to(device)
        if self.batch is not None:
            self.batch = self.batch.to(device)
        return self


class Batch(Data):
    @staticmethod
    def from_data_list(data_list: Sequence[Data]):
        sizes = [d.pos.shape[0] for d in data_list]
        batch = torch.repeat_interleave(torch.arange(len(data_list)), torch.tensor(sizes))
        return Batch(x=torch.cat([d.x for d in data_list]), pos=torch.cat([d.pos for d in data_list]), batch=batch)


class FixedPoints:
    """Resample to `num` points with replacement (the reference uses T.FixedPoints(num) with default replace=True)."""

    def __init__(self, num, generator: np.random.Generator = None):
        self.num = num
        self.gen = generator if generator is not None else np.random.default_rng()

    def __call__(self, data):
        choice = torch.from_numpy(self.gen.choice(data.pos.shape[0], self.num, replace=True))
        data.x, data.pos = data.x[choice], data.pos[choice]
        return data


# ---- counter-based T.FixedPoints draw (shared with csrc/small_kernels.hip::k_pack_scene; include/t2p.h) ---------------------
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_KEY_MUL = np.uint64(0xD6E8FEB86659FD93)


def mix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser, vectorised with wrap-around arithmetic."""
    with np.errstate(over="ignore"):
        x = (np.asarray(x, dtype=np.uint64) + np.uint64(0x9E3779B97F4A7C15)) & _M64
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return x ^ (x >> np.uint64(31))


def sample_keys(seed: int, sample_index, slot) -> np.ndarray:
    """uint64 key of object slot `slot` of sample (cell) `sample_index` under `seed` (arrays broadcast)."""
    with np.errstate(over="ignore"):
        k = mix64(np.uint64(int(seed) & 0xFFFFFFFFFFFFFFFF))
        k = mix64(k ^ ((np.asarray(sample_index).astype(np.uint64) * _KEY_MUL) & _M64))
        return mix64(k ^ ((np.asarray(slot).astype(np.uint64) * _KEY_MUL) & _M64))


def keyed_draws(keys, sizes, n_pts: int) -> np.ndarray:
    """The n_pts indices (with replacement) each key draws from an object of sizes[i] points: int64 [n, n_pts].
    Point p of key k: ((mix64(k ^ p * 0xD6E8FEB86659FD93) >> 32) * m) >> 32 - the statement the pack kernel executes."""
    keys = np.atleast_1d(np.asarray(keys, dtype=np.uint64))
    m = np.atleast_1d(np.asarray(sizes)).astype(np.uint64)
    with np.errstate(over="ignore"):
        r = mix64(keys[:, None] ^ ((np.arange(n_pts, dtype=np.uint64) * _KEY_MUL) & _M64)[None, :])
        return (((r >> np.uint64(32)) * m[:, None]) >> np.uint64(32)).astype(np.int64)


class KeyedFixedPoints:
    """T.FixedPoints(num) whose draw is the counter-based one above: the k-th object this transform is applied to is slot k of
    sample `sample_index` (batch_object_points applies a cell's transform to its objects in order)."""

    def __init__(self, num: int, seed: int, sample_index: int):
        self.num, self.seed, self.sample_index, self.slot = num, seed, sample_index, 0

    def __call__(self, data):
        key = sample_keys(self.seed, self.sample_index, self.slot)
        self.slot += 1
        choice = torch.from_numpy(keyed_draws(key, data.pos.shape[0], self.num)[0])
        data.x, data.pos = data.x[choice], data.pos[choice]
        return data


class NormalizeScale:
    def __call__(self, data):
        data.pos = data.pos - data.pos.mean(dim=-2, keepdim=True)
        data.pos = data.pos * ((1 / data.pos.abs().max()) * 0.999999)
        return data


class Compose:
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, data):
        for t in self.transforms:
            data = t(data)
        return data


def batch_object_points(objects: List[Object3d], transform):
    data_list = [transform(Data(x=torch.tensor(o.rgb, dtype=torch.float), pos=torch.tensor(o.xyz, dtype=torch.float)))
                 for o in objects]
    assert len(data_list) >= 1
    return Batch.from_data_list(data_list)


class HostStaging:
    """Pinned host buffers that outlive a call (the first version of pack_cells allocated four pinned tensors per call:
    a pinned allocation is a driver call that costs more than the copy it serves).  Two sets, used in turn, each with the
    event that marks the end of the last host-to-device copy out of it: a set is not rewritten before that copy is done."""

    def __init__(self):
        self.sets = [dict(), dict()]
        self.events = [None, None]
        self.turn = 0

    def next_set(self):
        self.turn ^= 1
        ev = self.events[self.turn]
        if ev is not None:
            ev.synchronize()
        return self.turn

    def buffer(self, which: int, name: str, numel: int) -> torch.Tensor:
        buf = self.sets[which].get(name)
        if buf is None or buf.numel() < numel:
            cap = max(int(numel * 1.25), 1 << 16)
            buf = torch.empty(cap, dtype=torch.float32, pin_memory=torch.cuda.is_available())
            self.sets[which][name] = buf
        return buf[:numel]

    def mark_copied(self, which: int, device=None, stream=None):
        """Records the end of the host-to-device copies out of set `which` - on the current stream of the device the copies
        went to (t.to(device, non_blocking=True) runs there, which need not be the current device)."""
        if torch.cuda.is_available():
            ev = torch.cuda.Event()
            ev.record(stream if stream is not None else torch.cuda.current_stream(device))
            self.events[which] = ev


class ObjectMeansCache:
    """Per-cell memo of the objects' (centre, mean colour) rows - models/object_encoder.py:121-131 recomputes
    `obj.get_center()` / `obj.get_color_rgb()` (a float64 NumPy mean over the RAW points of every object) in every call, which
    at 10-20 us per object is 20x the GPU time of the cell.  Keyed by the identity of the cell's object list; an entry keeps the
    list and its objects alive and is used only while the list still holds the very same objects.  The point arrays of an
    Object3d are treated as immutable, as the reference's dataset classes treat them (augmentation happens on the PyG batches):
    call clear() after editing `obj.xyz` / `obj.rgb` in place.  Bounded (least recently inserted entries leave first; 16,384
    cells by default: a KITTI360Pose evaluation database).  The models consult it in eval() mode only: the reference's TRAINING
    loaders hand over fresh lists every step (deep copies under flip_poses, dataloading/kitti360pose/utils.py:32-33; unpickled
    batches from DataLoader workers), which could never hit and would only be kept alive here."""

    def __init__(self, max_cells: int = 1 << 14):
        self.max_cells = max_cells
        self.d = {}

    def clear(self):
        self.d.clear()

    def get(self, objs):
        e = self.d.get(id(objs))
        if e is None:
            return None
        kept, snapshot, center, color = e
        # (tuple comparison short-cuts on identity; Object3d defines no __eq__, so equal means the very same objects)
        if kept is not objs or tuple(objs) != snapshot:
            return None
        return center, color

    def put(self, objs, center, color):
        if len(self.d) >= self.max_cells:
            for k in list(self.d.keys())[: self.max_cells // 8]:
                del self.d[k]
        self.d[id(objs)] = (objs, tuple(objs), center, color)


_U64 = 1.2e-16   # unit roundoff of float64 (rounded up)


def _means_f32(arrays) -> np.ndarray:
    """float32(np.mean(a, axis=0)) for every [m_i, 3] float64 array of a cell, without one NumPy call per object: the arrays
    are concatenated and summed by np.add.reduceat.  That sums in another ORDER than np.mean (pairwise blocks), so the float64
    results differ in their last bits; whichever order is used, a sum is within (m - 1) u sum|x| of the exact one.  Where the
    float32 roundings of (mean - tol) and (mean + tol) agree the result is the reference's bit for bit; the (rare) rows
    where they do not are recomputed with np.mean itself."""
    n = len(arrays)
    if any((not isinstance(a, np.ndarray)) or a.dtype != np.float64 or a.ndim != 2 or a.shape[0] == 0 for a in arrays):
        return np.stack([np.mean(a, axis=0) for a in arrays]).astype(np.float32)
    cnt = np.fromiter((a.shape[0] for a in arrays), dtype=np.int64, count=n)
    start = np.zeros(n, dtype=np.int64)
    np.cumsum(cnt[:-1], out=start[1:])
    cat = np.concatenate(arrays, axis=0)
    mean = np.add.reduceat(cat, start, axis=0) / cnt[:, None]
    tol = np.add.reduceat(np.abs(cat), start, axis=0) * (4.0 * _U64) + 1e-300   # both sums' bounds, divided by m, + the division
    lo, hi = (mean - tol).astype(np.float32), (mean + tol).astype(np.float32)
    out = mean.astype(np.float32)
    unsafe = np.flatnonzero((lo != hi).any(axis=1))
    for i in unsafe:
        out[i] = np.mean(arrays[i], axis=0).astype(np.float32)
    return out


_HOST_EXT = None


def host_ext():
    """The CPython helper _t2p_host (csrc/host_ext.c, built by build.py::build_host_ext), or None when it has not been built:
    the callers then take the NumPy route - the same host arithmetic, one cell at a time."""
    global _HOST_EXT
    if _HOST_EXT is None:
        import glob
        import importlib.util
        import os
        _HOST_EXT = False
        for path in glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_t2p_host*.so")):
            try:
                spec = importlib.util.spec_from_file_location("_t2p_host", path)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                _HOST_EXT = mod
                break
            except ImportError:
                continue
    return _HOST_EXT or None


def _plain_objects(objs) -> bool:
    """True if the centre / mean colour of every object is np.mean over its .xyz / .rgb (Object3d, or a class that inherits
    the reference's accessors unchanged)."""
    return all(type(o) is Object3d for o in objs) or all(
        hasattr(o, "xyz") and hasattr(o, "rgb") and type(o).get_center.__qualname__.endswith("Object3d.get_center")
        and type(o).get_color_rgb.__qualname__.endswith("Object3d.get_color_rgb") for o in objs)


def _means_from_sums(sums, abs_sums, rows, arrays_of):
    """float32 means from the helper's column sums: the same proof as _means_f32 (any summation order is within
    (m - 1) u sum|x| of the exact sum), np.mean itself for the rows where it does not decide the float32 rounding."""
    cnt = rows[:, None].astype(np.float64)
    mean = sums / cnt
    tol = abs_sums * (4.0 * _U64) + 1e-300
    lo, hi = (mean - tol).astype(np.float32), (mean + tol).astype(np.float32)
    out = mean.astype(np.float32)
    for i in np.flatnonzero((lo != hi).any(axis=1)):
        out[i] = np.mean(arrays_of(int(i)), axis=0).astype(np.float32)
    return out


_HOST_THREADS = {}


def host_threads(cap: int = 32) -> int:
    """(cached per cap: the probe opens cgroup files)"""
    n = _HOST_THREADS.get(cap)
    if n is None:
        n = _HOST_THREADS[cap] = _probe_host_threads(cap)
    return n


def _probe_host_threads(cap: int = 32) -> int:
    """Worker threads of the C helper: the CPUs this process may USE, at most `cap` (the pool's size) - the smaller of its
    affinity mask and its cgroup CPU quota (a container on a 256-thread host is typically granted a fraction of it; threads
    beyond the quota only add wake-ups and throttling)."""
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: (t.split()[0], t.split()[1])),):
        try:
            quota, period = parse(open(path).read())
            if quota != "max":
                n = min(n, max(1, int(int(quota) / int(period))))
        except (OSError, ValueError, IndexError):
            pass
    try:    # cgroup v1
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and p > 0:
            n = min(n, max(1, q // p))
    except (OSError, ValueError):
        pass
    return max(1, min(int(cap), n))


def object_means_many(cells, cache: ObjectMeansCache = None, threads: int = None):
    """object_means for a list of cells in ONE pass of the C helper (cells that are not in the cache: the first epoch, or
    evaluation.pipeline's single pass over the database): one walk of the object list for `.xyz` and `.rgb` together, the sums on
    the helper's persistent thread pool.  Returns a list of (centre [n, 3], colour [n, 3]) fp32 pairs."""
    ext = host_ext()
    if ext is None or not cells or not all(type(objs) is list and _plain_objects(objs) for objs in cells):
        return [object_means(objs, cache) for objs in cells]
    n = sum(len(objs) for objs in cells)
    sums, asums = np.empty((2, n, 3), dtype=np.float64), np.empty((2, n, 3), dtype=np.float64)
    rows = np.empty((2, n), dtype=np.int64)
    if ext.object_sums(cells, sums, asums, rows, int(threads or host_threads())) != n:
        return [object_means(objs, cache) for objs in cells]    # some array is not float64 [m, 3]: the NumPy route decides
    flat = [o for objs in cells for o in objs]
    center = _means_from_sums(sums[0], asums[0], rows[0], lambda i: flat[i].xyz)
    color = _means_from_sums(sums[1], asums[1], rows[1], lambda i: flat[i].rgb)
    out, a = [], 0
    for objs in cells:
        b = a + len(objs)
        pair = (center[a:b], color[a:b])
        if cache is not None:
            cache.put(objs, pair[0], pair[1])
        out.append(pair)
        a = b
    return out


def object_means(objs, cache: ObjectMeansCache = None):
    """([n, 3] fp32 centres, [n, 3] fp32 mean colours) of a cell's objects = torch.tensor([o.get_center() ...], dtype=float)
    of the reference (models/object_encoder.py:121-131), bit for bit."""
    if cache is not None:
        hit = cache.get(objs)
        if hit is not None:
            return hit
    if _plain_objects(objs):
        center, color = _means_f32([o.xyz for o in objs]), _means_f32([o.rgb for o in objs])
    else:   # a subclass that overrides the accessors: ask it
        center = np.stack([o.get_center() for o in objs]).astype(np.float32)
        color = np.stack([o.get_color_rgb() for o in objs]).astype(np.float32)
    if cache is not None:
        cache.put(objs, center, color)
    return center, color


_CAT_POOL = None


def _pack_pool():
    global _CAT_POOL
    if _CAT_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _CAT_POOL = ThreadPoolExecutor(max_workers=4, thread_name_prefix="t2p-pack")
    return _CAT_POOL


def _cat_into(tensors, out: torch.Tensor, rows_per_item, threads: int = 4):
    """torch.cat(tensors, out=out) in `threads` contiguous pieces on a small thread pool (torch.cat releases the GIL; one
    thread moves ~5 GB/s, and a 512-cell call concatenates 2 x 25 MB)."""
    n = len(tensors)
    if n < 64 or threads <= 1:
        torch.cat(tensors, out=out)
        return
    pool = _pack_pool()
    ends = np.cumsum(rows_per_item)
    cuts = [0] + [int(np.searchsorted(ends, ends[-1] * (k + 1) // threads, side="left")) + 1 for k in range(threads - 1)] + [n]
    cuts = sorted(set(min(c, n) for c in cuts))
    jobs = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        r0 = int(ends[a - 1]) if a > 0 else 0
        jobs.append(pool.submit(torch.cat, tensors[a:b], out=out[r0: int(ends[b - 1])]))
    for j in jobs:
        j.result()


def _check_batch_vectors(object_points, counts, n_pts: int):
    """Every cell's batch vector must be `n` contiguous groups of n_pts (Batch.from_data_list of n resampled objects,
    dataloading/kitti360pose/utils.py:110): a permuted or interleaved vector would pair points with the wrong object without
    changing any size.  ALL cells are checked, in three large tensor operations (the GIL is released inside them, so the check
    runs beside the caller's concatenations): the vectors back to back, viewed as [objects, n_pts], must be constant along
    each row and start with the object's index in its cell."""
    sel = [i for i, p in enumerate(object_points) if p.batch is not None]
    if not sel:
        return
    vecs = [object_points[i].batch for i in sel]
    cnt = np.asarray(counts)[sel]
    for i, b, n in zip(sel, vecs, cnt):
        if b.dim() != 1 or b.shape[0] != int(n) * n_pts:
            raise RuntimeError(f"encode_objects: cell {i}: batch vector is not {int(n)} contiguous groups of {n_pts}")
    ext = host_ext()
    if ext is not None and all(b.dtype is torch.int64 and not b.is_cuda and b.is_contiguous() for b in vecs):
        # one pass in C over the vectors where they lie (GIL released; `vecs` keeps them alive)
        ptrs = np.fromiter((b.data_ptr() for b in vecs), dtype=np.int64, count=len(vecs))
        bad = ext.check_groups(ptrs, np.ascontiguousarray(cnt, dtype=np.int64), int(n_pts), 2)
        if bad >= 0:
            i = sel[bad]
            raise RuntimeError(f"encode_objects: cell {i}: batch vector is not {int(counts[i])} contiguous groups of {n_pts}")
        return
    g = torch.cat(vecs).view(-1, n_pts)
    start = np.zeros(len(sel), dtype=np.int64)
    np.cumsum(cnt[:-1], out=start[1:])
    local = torch.from_numpy(np.arange(int(cnt.sum()), dtype=np.int64) - np.repeat(start, cnt)).to(g.device)
    bad = (g != local[:, None].to(g.dtype)).any(dim=1)
    if bool(bad.any()):
        o = int(torch.nonzero(bad)[0])
        i = sel[int(np.searchsorted(start, o, side="right")) - 1]
        raise RuntimeError(f"encode_objects: cell {i}: batch vector is not {int(counts[i])} contiguous groups of {n_pts}")


def pack_cells(objects: List[List[Object3d]], object_points, n_pts: int, zero_color: bool = False,
               staging: HostStaging = None, means_cache: ObjectMeansCache = None, skip_rgb: bool = False, device=None):
    """Flatten the (objects, object_points) pair of CellRetrievalNetwork.encode_objects into
    xyz, rgb [Nobj, n_pts, 3], center, mean_rgb [Nobj, 3] (fp32), cell_ptr int32 [B+1].
    Without `device`: host tensors (pinned when CUDA is present).  With `device`: the arrays are returned ON the device - the
    point batches are concatenated straight into `staging`'s pinned buffers (one pass over the bytes, no per-call pinned
    allocation) and copied asynchronously; batches that already live on the device are concatenated there.
    skip_rgb: return rgb = None (the caller zeroes the colours on the device: models/object_encoder.py:86-90).
    Every cell's batch vector is verified (on the packing pool, beside the concatenations)."""
    if len(objects) != len(object_points):
        raise RuntimeError(f"encode_objects: {len(objects)} object lists but {len(object_points)} point batches")
    n_cells = len(objects)
    counts = np.fromiter((len(o) for o in objects), dtype=np.int64, count=n_cells)
    cell_ptr = np.zeros(n_cells + 1, dtype=np.int32)
    np.cumsum(counts, out=cell_ptr[1:])
    n_obj = int(cell_ptr[-1])
    if n_cells and counts.min() < 1:
        raise RuntimeError(f"encode_objects: cell {int(np.argmin(counts))} has no objects")
    pos_list = [p.pos for p in object_points]
    n_rows = np.fromiter((t.shape[0] for t in pos_list), dtype=np.int64, count=n_cells)
    if n_cells and (n_rows != counts * n_pts).any():
        i = int(np.flatnonzero(n_rows != counts * n_pts)[0])
        raise RuntimeError(f"encode_objects: cell {i} has {counts[i]} objects but {n_rows[i]} points; every object "
                           f"must be resampled to {n_pts} points (T.FixedPoints({n_pts}))")
    checking = _pack_pool().submit(_check_batch_vectors, object_points, counts, n_pts) if n_cells else None
    want_rgb = not (skip_rgb or zero_color)
    on_device = n_cells > 0 and pos_list[0].is_cuda
    pin = torch.cuda.is_available()

    def flat(which):      # float32 tensors without a grad_fn pass as they are (the common case: one cheap scan)
        ts = pos_list if which == "pos" else [p.x for p in object_points]
        if any(t.dtype is not torch.float32 or t.requires_grad for t in ts):
            ts = [(t.detach() if t.requires_grad else t).float() for t in ts]
        return ts

    if on_device:
        xyz = torch.cat(flat("pos")).reshape(n_obj, n_pts, 3)
        rgb = torch.cat(flat("x")).reshape(n_obj, n_pts, 3) if want_rgb else None
        which = None
    else:
        which = staging.next_set() if staging is not None else None

        def host(name, numel):
            if staging is not None:
                return staging.buffer(which, name, numel)
            return torch.empty(numel, dtype=torch.float32, pin_memory=pin)
        xyz = host("xyz", n_obj * n_pts * 3).view(n_obj * n_pts, 3)
        rows = counts * n_pts
        if n_cells:
            _cat_into(flat("pos"), xyz, rows)
        xyz = xyz.view(n_obj, n_pts, 3)
        rgb = None
        if want_rgb:
            rgb = host("rgb", n_obj * n_pts * 3).view(n_obj * n_pts, 3)
            if n_cells:
                _cat_into(flat("x"), rgb, rows)
            rgb = rgb.view(n_obj, n_pts, 3)
    if zero_color and not skip_rgb:
        rgb = torch.zeros((n_obj, n_pts, 3), dtype=torch.float32, device=xyz.device)
    small = (staging.buffer(which, "means", n_obj * 6) if (staging is not None and which is not None)
             else torch.empty(n_obj * 6, dtype=torch.float32, pin_memory=pin)).view(2, n_obj, 3)
    small_np = small.numpy()
    if n_cells:     # (one concatenation per array: a NumPy slice assignment per cell costs more than the cache lookup it follows)
        # (inlined ObjectMeansCache.get: a function call per cell costs as much as the look-up; threads do not help, GIL-bound)
        memo = means_cache.d if means_cache is not None else {}
        rows, miss = [], []
        for i, objs in enumerate(objects):
            e = memo.get(id(objs))
            if e is not None and e[0] is objs and tuple(objs) == e[1]:
                rows.append((e[2], e[3]))
            else:
                rows.append(None)
                miss.append(i)
        if miss:    # (first epoch / a single evaluation pass: all of them at once through the C helper)
            for i, pair in zip(miss, object_means_many([objects[i] for i in miss], means_cache)):
                rows[i] = pair
        np.concatenate([r[0] for r in rows], axis=0, out=small_np[0])
        np.concatenate([r[1] for r in rows], axis=0, out=small_np[1])
    center, mean_rgb = small[0], small[1]
    if checking is not None:
        checking.result()      # (raises here what the check found)
    if device is not None:
        to = lambda t: None if t is None else t.to(device, non_blocking=True)
        xyz, rgb, center, mean_rgb = to(xyz), to(rgb), to(center), to(mean_rgb)
        if staging is not None and which is not None:
            staging.mark_copied(which, device)
    return xyz, rgb, center, mean_rgb, cell_ptr


def flatten_raw_objects(objects: List[List[Object3d]], n_pts: int, generator: np.random.Generator):
    """Host side of the on-device packing (ops.pack_objects): raw points of all objects back to back + CSR pointers +
    the T.FixedPoints draw (indices with replacement from the seeded generator, one row per object)."""
    flat = [o for objs in objects for o in objs]
    sizes = np.array([len(o.xyz) for o in flat], dtype=np.int64)
    if (sizes < 1).any():
        raise RuntimeError("an object without points cannot be resampled")
    obj_ptr = np.zeros(len(flat) + 1, dtype=np.int32)
    obj_ptr[1:] = np.cumsum(sizes)
    raw_xyz = np.concatenate([np.asarray(o.xyz, dtype=np.float32) for o in flat], 0)
    raw_rgb = np.concatenate([np.asarray(o.rgb, dtype=np.float32) for o in flat], 0)
    sample_idx = np.stack([generator.choice(int(m), n_pts, replace=True) for m in sizes]).astype(np.int32)
    cell_ptr = np.zeros(len(objects) + 1, dtype=np.int32)
    cell_ptr[1:] = np.cumsum([len(o) for o in objects])
    return raw_xyz, raw_rgb, obj_ptr, sample_idx, cell_ptr


def draw_rotations(n_obj: int, degrees: float, generator: np.random.Generator) -> np.ndarray:
    """Host draw of T.RandomRotate(degrees, axis=2) (training/coarse.py:196): one angle ~ U(-degrees, degrees) per
    object, returned as [n_obj, 2] fp32 (cos, sin) -- the values PyG puts into its fp32 rotation matrix."""
    ang = np.pi * generator.uniform(-abs(degrees), abs(degrees), n_obj) / 180.0
    return np.stack([np.cos(ang), np.sin(ang)], 1).astype(np.float32)

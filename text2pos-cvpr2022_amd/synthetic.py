"""Synthetic KITTI360Pose-shaped inputs (the dataset itself is not available): SURVEY.md section 8(d).

Everything is a pure function of (seed, global object / cell / query index) through a counter-based hash, so any
rank can generate exactly its shard and every run of bench.py / the tests sees identical values.

Object: shape in {planar patch 45 %, pole 25 %, box surface 30 %}; m = round(exp(U[ln 25, ln 4000])) base points
(25 = smallest CLASS_TO_MINPOINTS, datapreparation/kitti360pose/utils.py:122-145); 256 indices drawn with replacement
(T.FixedPoints(256): duplicates are present); NormalizeScale; rgb = clip(COLORS[c] + N(0, 0.05^2), 0, 1);
centre ~ U[0,1]^2 x U[0,0.3]; mean_rgb = mean of the drawn rgb.
Cell: n ~ U{6..26} objects (or a fixed n).  Text: 6 hints "The pose is {dir} of a {color} {label}."
(dataloading/kitti360pose/base.py:63-65) joined by a space (dataloading/kitti360pose/cells.py:82).

Two kinds of text: `make_texts` draws every hint at random (the benchmark's queries: the encoders' work does not depend on what
a sentence says); `make_paired_texts` DESCRIBES cell i - six of its objects, each by the direction of its centre from the cell's
middle, its colour name and a label that follows from its shape and height - so that a (text, cell) pair carries the signal the
reference trains on (training/coarse.py:31-62) and a model trained on the pairs retrieves above chance.
"""
from types import SimpleNamespace

import numpy as np

from .data import CLASS_NAMES, COLOR_NAMES, COLORS

DIRECTIONS = ["north", "south", "east", "west", "on-top"]  # datapreparation/kitti360pose/select.py:13-27
LABELS = [c for c in CLASS_NAMES if c != "pad"]
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser (vectorised, wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return x ^ (x >> np.uint64(31))


def _key(seed, *parts) -> np.ndarray:
    k = _mix(np.asarray(seed, dtype=np.uint64))
    for p in parts:
        with np.errstate(over="ignore"):
            k = _mix(k ^ (np.asarray(p).astype(np.uint64) * np.uint64(0xD6E8FEB86659FD93) & _M64))
    return k


def _u01(k: np.ndarray) -> np.ndarray:
    return ((k >> np.uint64(11)).astype(np.float64) + 0.5) / float(1 << 53)


def _normal(k: np.ndarray) -> np.ndarray:
    u1, u2 = _u01(k), _u01(_mix(k ^ np.uint64(0xA5A5A5A5A5A5A5A5)))
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def default_args(**kw):
    a = dict(embed_dim=256, use_features=["class", "color", "position"], variation=0, class_embed=False,
             color_embed=False, pointnet_layers=3, pointnet_variation=0, pointnet_numpoints=256, pointnet_path=None,
             pointnet_freeze=False, pointnet_features=2)
    a.update(kw)
    return SimpleNamespace(**a)


def known_words():
    words = set("the pose is of a".split())
    for group in (DIRECTIONS, COLOR_NAMES, LABELS):
        for item in group:
            words.update(item.lower().split())
    return sorted(words)


def make_objects(seed: int, obj_lo: int, obj_hi: int, n_pts: int = 256):
    """Objects [obj_lo, obj_hi) of the stream `seed`: xyz, rgb [n, n_pts, 3], center, mean_rgb [n, 3] (fp32)."""
    n = obj_hi - obj_lo
    oid = np.arange(obj_lo, obj_hi, dtype=np.uint64)
    shape_u = _u01(_key(seed, oid, 1))
    m = np.rint(np.exp(np.log(25.0) + _u01(_key(seed, oid, 2)) * (np.log(4000.0) - np.log(25.0)))).astype(np.int64)
    ext = 0.3 + 0.7 * _u01(_key(seed, oid[:, None], 3, np.arange(3)[None, :]))          # [n,3] extents
    draw = (_u01(_key(seed, oid[:, None], 4, np.arange(n_pts)[None, :])) * m[:, None]).astype(np.int64)  # [n,P]
    draw = np.minimum(draw, m[:, None] - 1)
    # base point `draw` of object `oid`: three uniforms + a face selector, all functions of (oid, draw)
    pk = _key(seed, oid[:, None, None], 5, draw[:, :, None], np.arange(4)[None, None, :])   # [n,P,4]
    u = _u01(pk[..., :3]) * 2.0 - 1.0
    face = (_u01(pk[..., 3]) * 6.0).astype(np.int64)
    plane = shape_u < 0.45
    pole = (shape_u >= 0.45) & (shape_u < 0.70)
    pts = np.empty((n, n_pts, 3), dtype=np.float64)
    # planar patch: thin in z
    pts[...] = u * ext[:, None, :]
    pts[plane, :, 2] = u[plane, :, 2] * 0.02
    # pole: thin in x, y
    pts[pole, :, 0] = u[pole, :, 0] * 0.03
    pts[pole, :, 1] = u[pole, :, 1] * 0.03
    # box surface: clamp one coordinate to a face
    box = ~(plane | pole)
    if box.any():
        b = pts[box]
        f = face[box]
        for axis in range(3):
            for sign, fid in ((-1.0, 2 * axis), (1.0, 2 * axis + 1)):
                sel = f == fid
                col = b[..., axis]
                col[sel] = sign * np.broadcast_to(ext[box][:, None, axis], col.shape)[sel]
        pts[box] = b
    pts32 = pts.astype(np.float32)
    # NormalizeScale in fp32, as the PyG transform does on float tensors
    pts32 = pts32 - pts32.mean(axis=1, keepdims=True)
    scale = (np.float32(1.0) / np.abs(pts32).reshape(n, -1).max(axis=1)) * np.float32(0.999999)
    xyz = (pts32 * scale[:, None, None]).astype(np.float32)
    color_id = (_u01(_key(seed, oid, 6)) * 8.0).astype(np.int64)
    noise = _normal(_key(seed, oid[:, None, None], 7, draw[:, :, None], np.arange(3)[None, None, :])) * 0.05
    rgb = np.clip(COLORS[color_id][:, None, :] + noise, 0.0, 1.0).astype(np.float32)
    cu = _u01(_key(seed, oid[:, None], 8, np.arange(3)[None, :]))
    center = (cu * np.array([1.0, 1.0, 0.3])).astype(np.float32)
    mean_rgb = rgb.astype(np.float64).mean(axis=1).astype(np.float32)
    return xyz, rgb, center, mean_rgb


# labels a paired text can use for an object: by the generator's shape class (planar patch / pole / box surface), seven each;
# which of the seven follows from the height of the object's centre, a feature the model sees (models/object_encoder.py:127-131)
LABEL_GROUPS = (("road", "sidewalk", "parking", "terrain", "wall", "fence", "guard rail"),
                ("pole", "traffic light", "traffic sign", "stop", "smallpole", "lamp", "trash bin"),
                ("building", "garage", "vending machine", "box", "bridge", "tunnel", "vegetation"))


def object_attributes(seed: int, obj_lo: int, obj_hi: int):
    """What make_objects draws for objects [obj_lo, obj_hi) besides their points: shape class (0 planar patch, 1 pole, 2 box
    surface), colour index into COLORS / COLOR_NAMES, centre [n, 3] (float64, before the cast to fp32)."""
    oid = np.arange(obj_lo, obj_hi, dtype=np.uint64)
    shape_u = _u01(_key(seed, oid, 1))
    shape = np.where(shape_u < 0.45, 0, np.where(shape_u < 0.70, 1, 2)).astype(np.int64)
    color_id = (_u01(_key(seed, oid, 6)) * 8.0).astype(np.int64)
    center = _u01(_key(seed, oid[:, None], 8, np.arange(3)[None, :])) * np.array([1.0, 1.0, 0.3])
    return shape, color_id, center


def describe_object(shape: int, color_id: int, center) -> str:
    """One hint in the reference's template (dataloading/kitti360pose/base.py:63-65) for an object seen from the middle of its
    cell: direction as datapreparation/kitti360pose/descriptions.py words it (on-top when close, else the dominant axis)."""
    dx, dy = float(center[0]) - 0.5, float(center[1]) - 0.5
    if max(abs(dx), abs(dy)) < 0.1:
        direction = "on-top"
    elif abs(dx) >= abs(dy):
        direction = "east" if dx > 0 else "west"
    else:
        direction = "north" if dy > 0 else "south"
    label = LABEL_GROUPS[int(shape)][min(6, int(float(center[2]) / 0.3 * 7.0))]
    return f"The pose is {direction} of a {COLOR_NAMES[int(color_id)]} {label}."


def make_paired_texts(seed: int, n_cells: int, cell_lo: int = 0, cell_hi: int = None, n_hints: int = 6, fixed_n: int = 0):
    """Text i describes cell i of the `seed` database (cells [cell_lo, cell_hi)): `n_hints` of its objects, chosen by a hash of
    (seed, cell, object) without repetition (a cell with fewer objects repeats them in order)."""
    cell_hi = n_cells if cell_hi is None else cell_hi
    sizes = cell_sizes(seed, n_cells, fixed_n)
    ptr = np.zeros(n_cells + 1, dtype=np.int64)
    ptr[1:] = np.cumsum(sizes)
    o_lo, o_hi = int(ptr[cell_lo]), int(ptr[cell_hi])
    shape, color_id, center = object_attributes(seed, o_lo, o_hi)
    order_key = _key(seed, np.arange(o_lo, o_hi, dtype=np.uint64), 13)
    texts = []
    for c in range(cell_lo, cell_hi):
        a, b = int(ptr[c]) - o_lo, int(ptr[c + 1]) - o_lo
        order = a + np.argsort(order_key[a:b], kind="stable")
        picks = [int(order[j % (b - a)]) for j in range(n_hints)]
        texts.append(" ".join(describe_object(shape[i], color_id[i], center[i]) for i in picks))
    return texts


def cell_sizes(seed: int, n_cells: int, fixed_n: int = 0) -> np.ndarray:
    if fixed_n > 0:
        return np.full(n_cells, fixed_n, dtype=np.int32)
    cid = np.arange(n_cells, dtype=np.uint64)
    return (6 + (_u01(_key(seed, cid, 9)) * 21.0).astype(np.int64)).astype(np.int32)  # U{6..26}


def make_cells(seed: int, n_cells: int, cell_lo: int = 0, cell_hi: int = None, fixed_n: int = 0, n_pts: int = 256):
    """Cells [cell_lo, cell_hi) of an n_cells database: (xyz, rgb, center, mean_rgb, cell_ptr[int32, local])."""
    cell_hi = n_cells if cell_hi is None else cell_hi
    sizes = cell_sizes(seed, n_cells, fixed_n)
    ptr = np.zeros(n_cells + 1, dtype=np.int64)
    ptr[1:] = np.cumsum(sizes)
    o_lo, o_hi = int(ptr[cell_lo]), int(ptr[cell_hi])
    parts = [make_objects(seed, a, min(a + 8192, o_hi), n_pts) for a in range(o_lo, o_hi, 8192)]
    if parts:
        xyz, rgb, center, mean_rgb = (np.concatenate([p[i] for p in parts], 0) for i in range(4))
    else:
        xyz = rgb = np.zeros((0, n_pts, 3), np.float32)
        center = mean_rgb = np.zeros((0, 3), np.float32)
    cell_ptr = (ptr[cell_lo: cell_hi + 1] - o_lo).astype(np.int32)
    return xyz, rgb, center, mean_rgb, cell_ptr


def make_texts(seed: int, q_lo: int, q_hi: int, n_hints: int = 6):
    qid = np.arange(q_lo, q_hi, dtype=np.uint64)
    pick = lambda salt, n: (_u01(_key(seed, qid[:, None], salt, np.arange(n_hints)[None, :])) * n).astype(np.int64)
    d, c, l = pick(10, len(DIRECTIONS)), pick(11, len(COLOR_NAMES)), pick(12, len(LABELS))
    return [" ".join(f"The pose is {DIRECTIONS[d[i, j]]} of a {COLOR_NAMES[c[i, j]]} {LABELS[l[i, j]]}."
                     for j in range(n_hints)) for i in range(q_hi - q_lo)]

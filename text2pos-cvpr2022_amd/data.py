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


def pack_cells(objects: List[List[Object3d]], object_points, n_pts: int, zero_color: bool = False):
    """Flatten the (objects, object_points) pair of CellRetrievalNetwork.encode_objects into host arrays:
    xyz, rgb [Nobj, n_pts, 3], center, mean_rgb [Nobj, 3] (fp32, pinned when CUDA is present), cell_ptr int32 [B+1]."""
    if len(objects) != len(object_points):
        raise RuntimeError(f"encode_objects: {len(objects)} object lists but {len(object_points)} point batches")
    counts = [len(o) for o in objects]
    cell_ptr = np.zeros(len(objects) + 1, dtype=np.int32)
    cell_ptr[1:] = np.cumsum(counts)
    n_obj = int(cell_ptr[-1])
    pin = torch.cuda.is_available()
    xyz = torch.empty((n_obj, n_pts, 3), dtype=torch.float32, pin_memory=pin)
    rgb = torch.empty((n_obj, n_pts, 3), dtype=torch.float32, pin_memory=pin)
    center = torch.empty((n_obj, 3), dtype=torch.float32, pin_memory=pin)
    mean_rgb = torch.empty((n_obj, 3), dtype=torch.float32, pin_memory=pin)
    for i, (objs, pts) in enumerate(zip(objects, object_points)):
        lo, hi = int(cell_ptr[i]), int(cell_ptr[i + 1])
        n = hi - lo
        if n < 1:
            raise RuntimeError(f"encode_objects: cell {i} has no objects")
        if pts.pos.shape[0] != n * n_pts:
            raise RuntimeError(f"encode_objects: cell {i} has {n} objects but {pts.pos.shape[0]} points; every object "
                               f"must be resampled to {n_pts} points (T.FixedPoints({n_pts}))")
        if pts.batch is not None:
            expect = torch.arange(n).repeat_interleave(n_pts)
            if not torch.equal(pts.batch.cpu().long(), expect):
                raise RuntimeError(f"encode_objects: cell {i}: batch vector is not {n} contiguous groups of {n_pts}")
        xyz[lo:hi] = pts.pos.detach().cpu().float().reshape(n, n_pts, 3)
        if zero_color:
            rgb[lo:hi] = 0.0
        else:
            rgb[lo:hi] = pts.x.detach().cpu().float().reshape(n, n_pts, 3)
        # same conversion as the reference: torch.tensor(list of float64 means, dtype=torch.float)
        center[lo:hi] = torch.tensor(np.stack([o.get_center() for o in objs]), dtype=torch.float)
        mean_rgb[lo:hi] = torch.tensor(np.stack([o.get_color_rgb() for o in objs]), dtype=torch.float)
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

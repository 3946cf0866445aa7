"""On-disk formats of the reference: KITTI360Pose scenes and whole-module checkpoints.

  * `<base_path>/cells/<scene>.pkl`, `<base_path>/poses/<scene>.pkl`: pickled lists of the reference's `Cell` / `Pose`
    objects (dataloading/kitti360pose/base.py:37-45), written under the module path
    `datapreparation.kitti360pose.imports` or, for older data, `datapreparation.kitti360.imports`
    (dataloading/__init__.py:8-10).  `load_scene` unpickles them onto this package's classes (data.py) without the
    reference on the path: instances are restored by attribute dict, so every field the reference stored survives.
  * checkpoints are whole pickled modules (`torch.save(model, path)`, training/coarse.py:330-331, loaded at
    evaluation/pipeline.py:313-314); `load_reference_checkpoint` turns one into a plain state_dict by unpickling every
    class it does not know (the reference's `models.*`, `torch_geometric.*`, `easydict`) as an empty `nn.Module` /
    dict shell -- `nn.Module.state_dict()` only needs `_parameters`, `_buffers` and `_modules`, which the pickle carries.
"""
import io
import os
import pickle
from typing import Dict, List, Sequence

import numpy as np
import torch
import torch.nn as nn

from . import data as D

_DATA_MODULES = ("datapreparation.kitti360pose.imports", "datapreparation.kitti360.imports")
_DATA_CLASSES = {"Object3d": D.Object3d, "Cell": D.Cell, "Pose": D.Pose, "DescriptionPoseCell": D.DescriptionPoseCell,
                 "DescriptionBestCell": D.DescriptionBestCell}


class _SceneUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module in _DATA_MODULES:
            if name not in _DATA_CLASSES:
                raise pickle.UnpicklingError(f"{module}.{name} is not a KITTI360Pose data class this package knows")
            return _DATA_CLASSES[name]
        return super().find_class(module, name)


def load_pickle(path: str):
    with open(path, "rb") as f:
        return _SceneUnpickler(f).load()


class Scenes:
    """What Kitti360CoarseDatasetMulti exposes to the evaluation (dataloading/kitti360pose/cells.py:113-187)."""

    def __init__(self, cells: List[D.Cell], poses: List[D.Pose]):
        ids = [c.id for c in cells]
        if len(set(ids)) != len(ids):
            raise RuntimeError("cell ids repeat")  # cells.py:149-150
        self.all_cells, self.all_poses = cells, poses
        self.cells_dict = {c.id: c for c in cells}
        self.hint_descriptions = [[f"The pose is {d.direction} of a {d.object_color_text} {d.object_label}."
                                   for d in p.descriptions] for p in poses]   # base.py:57-66

    @property
    def texts(self) -> List[str]:
        return [" ".join(h) for h in self.hint_descriptions]  # cells.py:82

    def get_known_words(self) -> List[str]:
        words = [w for hints in self.hint_descriptions for h in hints
                 for w in h.replace(".", "").replace(",", "").lower().split()]
        return list(np.unique(words))  # base.py:71-76

    def get_known_classes(self) -> List[str]:
        return list(D.KNOWN_CLASSES)


def load_scenes(base_path: str, scene_names: Sequence[str]) -> Scenes:
    cells, poses = [], []
    for s in scene_names:
        cells += load_pickle(os.path.join(base_path, "cells", f"{s}.pkl"))
        poses += load_pickle(os.path.join(base_path, "poses", f"{s}.pkl"))
    return Scenes(cells, poses)


def save_scene(base_path: str, scene_name: str, cells: List[D.Cell], poses: List[D.Pose]):
    """Writes the two pickles in the reference's directory layout (with this package's classes)."""
    for sub, obj in (("cells", cells), ("poses", poses)):
        os.makedirs(os.path.join(base_path, sub), exist_ok=True)
        with open(os.path.join(base_path, sub, f"{scene_name}.pkl"), "wb") as f:
            pickle.dump(obj, f)


# ---- checkpoints --------------------------------------------------------------------------------------------------------
class _Shell(nn.Module):
    """Stand-in for a module class that is not importable here; keeps whatever state the pickle assigns."""

    def __init__(self, *a, **k):
        nn.Module.__init__(self)


class _DictShell(dict):
    def __init__(self, *a, **k):
        dict.__init__(self)

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {})


_shells: Dict[str, type] = {}


class _CheckpointUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        try:
            return super().find_class(module, name)
        except (ImportError, AttributeError):
            key = f"{module}.{name}"
            if key not in _shells:
                base = _DictShell if module.startswith("easydict") or name in ("Namespace", "EasyDict") else _Shell
                _shells[key] = type(name, (base,), {"__module__": module})
            return _shells[key]


class _CheckpointPickle:
    """`pickle_module` for torch.load."""
    __name__ = "t2p_checkpoint_pickle"
    Unpickler = _CheckpointUnpickler

    @staticmethod
    def load(f, **kw):
        return _CheckpointUnpickler(f, **kw).load()


def load_reference_checkpoint(path: str) -> Dict[str, torch.Tensor]:
    """Whole-module `.pth` of the reference (or a plain state_dict file) -> state_dict on the CPU."""
    obj = torch.load(path, map_location="cpu", pickle_module=_CheckpointPickle, weights_only=False)
    if isinstance(obj, nn.Module):
        return obj.state_dict()
    if isinstance(obj, dict):
        return obj
    raise RuntimeError(f"{path}: neither a module nor a state_dict ({type(obj)})")

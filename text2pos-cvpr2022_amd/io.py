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


# Globals a pickle may resolve here.  Unpickling executes whatever callable a file names, so neither loader resolves
# anything outside this allowlist: containers and scalars of builtins, NumPy's array reconstruction, torch's tensor /
# parameter rebuild helpers and its nn module classes.  (What remains trusted: torch's own rebuild functions and the
# `__setstate__` of torch.nn modules; do not load checkpoints from sources you would not run code from.)
_SAFE_BUILTINS = {"dict", "list", "tuple", "set", "frozenset", "int", "float", "bool", "str", "bytes", "bytearray", "complex",
                  "slice", "range", "object"}
_NUMPY_MODULES = ("numpy", "numpy.core.multiarray", "numpy._core.multiarray", "numpy.core.numeric", "numpy._core.numeric")
_NUMPY_NAMES = {"_reconstruct", "ndarray", "dtype", "scalar", "_frombuffer"}


_TORCH_REBUILD = {"_rebuild_tensor", "_rebuild_tensor_v2", "_rebuild_tensor_v3", "_rebuild_parameter",
                  "_rebuild_parameter_with_state", "_rebuild_qtensor", "_rebuild_device_tensor_from_numpy",
                  "_rebuild_device_tensor_from_cpu_tensor", "_rebuild_wrapper_subclass", "_rebuild_sparse_tensor",
                  "_rebuild_nested_tensor", "_rebuild_meta_tensor_no_storage"}


def _allowed_global(module: str, name: str, torch_ok: bool) -> bool:
    module = {"__builtin__": "builtins", "copy_reg": "copyreg"}.get(module, module)   # protocol <= 2 spells them the py2 way
    if "." in name:   # protocol 4 resolves dotted names attribute by attribute: "torch.os.system" under an allowed module
        return False  # would reach any callable.  Nothing legitimate here is a nested attribute.
    if module == "builtins":
        return name in _SAFE_BUILTINS
    if module == "collections":
        return name in ("OrderedDict", "defaultdict")
    if module == "copyreg":          # object reconstruction helpers of pickle protocols 0-2 (they instantiate a resolved class)
        return name in ("_reconstructor", "__newobj__", "__newobj_ex__")
    if module in _NUMPY_MODULES:
        return name in _NUMPY_NAMES
    if not torch_ok:
        return False
    if module == "torch._utils":
        return name in _TORCH_REBUILD
    if module == "torch":
        return name.endswith("Storage") or name in ("Size", "device", "Tensor", "dtype", "float32", "float64", "int64")
    if module == "torch.nn.parameter":
        return name in ("Parameter", "Buffer")
    return module.startswith("torch.nn.modules.") or module == "torch.storage" and name in ("TypedStorage", "UntypedStorage")


class _SceneUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module in _DATA_MODULES:
            if name not in _DATA_CLASSES:
                raise pickle.UnpicklingError(f"{module}.{name} is not a KITTI360Pose data class this package knows")
            return _DATA_CLASSES[name]
        # scenes written by save_scene carry this package's own module path (under either of its import names)
        if module in (D.__name__, "text2pos_amd.data", "text2pos-cvpr2022_amd.data") and name in _DATA_CLASSES:
            return _DATA_CLASSES[name]
        if not _allowed_global(module, name, torch_ok=False):
            raise pickle.UnpicklingError(f"scene pickle names {module}.{name}, which is not on the allowlist of data classes, "
                                         "containers and NumPy array helpers")
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
        if _allowed_global(module, name, torch_ok=True):
            try:
                obj = super().find_class(module, name)
            except (ImportError, AttributeError):
                obj = None
            # torch.nn.modules.*: only nn.Module classes DEFINED in that module (not what the module happens to import)
            if obj is not None and module.startswith("torch.nn.modules."):
                if not (isinstance(obj, type) and issubclass(obj, nn.Module) and obj.__module__ == module):
                    raise pickle.UnpicklingError(f"checkpoint names {module}.{name}, which is not an nn.Module class of that module")
            if obj is not None:
                return obj
        # everything else (the reference's models.*, torch_geometric.*, easydict, argparse.Namespace, ... - importable or
        # not) becomes an inert shell: a bare nn.Module or a dict that keeps the state the pickle assigns
        key = f"{module}.{name}"
        if key not in _shells:
            base = _DictShell if module.startswith("easydict") or name in ("Namespace", "SimpleNamespace", "EasyDict") else _Shell
            _shells[key] = type(name, (base,), {"__module__": module})
        return _shells[key]


class _CheckpointPickle:
    """`pickle_module` for torch.load."""
    __name__ = "t2p_checkpoint_pickle"
    Unpickler = _CheckpointUnpickler

    @staticmethod
    def load(f, **kw):
        return _CheckpointUnpickler(f, **kw).load()


def load_reference_checkpoint(path: str, return_args: bool = False):
    """Whole-module `.pth` of the reference (or a plain state_dict file) -> state_dict on the CPU.
    return_args: also return the training arguments pickled inside a whole-module checkpoint (`model.args`, an EasyDict in
    the reference: embed_dim, use_features, variation, pointnet_features, class_embed, color_embed, num_layers, ...) as a
    plain dict ({} for a bare state_dict).  The reference evaluates the pickled module itself, so it honours them
    (evaluation/pipeline.py:313-314); a state_dict alone does not say whether e.g. --variation 1 was trained."""
    obj = torch.load(path, map_location="cpu", pickle_module=_CheckpointPickle, weights_only=False)
    if isinstance(obj, nn.Module):
        sd, a = obj.state_dict(), getattr(obj, "args", None)
        args = dict(a) if isinstance(a, dict) else dict(getattr(a, "__dict__", {}) or {})
        if isinstance(a, dict):
            args.update({k: v for k, v in getattr(a, "__dict__", {}).items() if not k.startswith("_")})
    elif isinstance(obj, dict):
        sd, args = obj, {}
    else:
        raise RuntimeError(f"{path}: neither a module nor a state_dict ({type(obj)})")
    return (sd, args) if return_args else sd

"""Drop-in for the reference's fine model: same constructor, `forward(objects, hints, object_points)` and state_dict
layout as models/superglue_matcher.py::SuperGlueMatch (incl. the un-used `superglue.kenc.*` parameters, so that whole
checkpoints load with strict=True), with the arithmetic executed by libt2p_hip.so on an MI355X:

  ObjectEncoder.forward (models/object_encoder.py:61-142)   -> t2p_encode_cells(objects_only)   (csrc/api.hip)
  LanguageEncoder per hint sentence (models/modules.py:59-92) -> t2p_encode_text                (csrc/lstm.hip)
  SuperGlue.forward + mlp_offsets (models/superglue.py:239-330, models/superglue_matcher.py:116) -> t2p_match (csrc/match.hip)

Callers that drop in unchanged: evaluation/pipeline.py:189-191 (run_fine) and training/fine.py's eval loop.
Forward-only, eval mode (SURVEY.md 8(f) #4 is the training row).
"""
from copy import deepcopy
from typing import List

import numpy as np
import torch
import torch.nn as nn

from . import ops, packing
from .data import HostStaging, ObjectMeansCache, pack_cells
from .modules import LanguageEncoder, PicklableModule, tokenize
from .object_encoder import ObjectEncoder

MATCH_THRESHOLD = 0.2  # models/superglue_matcher.py:80


def _conv_mlp(channels: List[int]) -> nn.Sequential:
    """Parameter layout of models/superglue.py::MLP (Conv1d k=1 [+ BatchNorm1d + ReLU])."""
    layers = []
    for i in range(1, len(channels)):
        layers.append(nn.Conv1d(channels[i - 1], channels[i], kernel_size=1, bias=True))
        if i < len(channels) - 1:
            layers += [nn.BatchNorm1d(channels[i]), nn.ReLU()]
    return nn.Sequential(*layers)


class _Holder(nn.Module):
    """Parameter container: the arithmetic of these sub-modules runs inside t2p_match."""

    def forward(self, *a, **k):
        raise NotImplementedError("runs fused inside SuperGlueMatch.forward (HIP)")


class SuperGlue(_Holder):
    """Parameters of models/superglue.py::SuperGlue under the same keys."""

    def __init__(self, config: dict):
        super().__init__()
        d = config["descriptor_dim"]
        self.config = dict(config)
        self.kenc = _Holder()
        self.kenc.encoder = _conv_mlp([3, 32, 64, 128, 256, d])  # KeypointEncoder: constructed, never used (:234)
        self.gnn = _Holder()
        self.gnn.names = list(config["GNN_layers"])
        layers = []
        for _ in self.gnn.names:
            layer = _Holder()
            layer.attn = _Holder()
            layer.attn.merge = nn.Conv1d(d, d, kernel_size=1)
            layer.attn.proj = nn.ModuleList([deepcopy(layer.attn.merge) for _ in range(3)])
            layer.mlp = _conv_mlp([2 * d, 2 * d, d])
            nn.init.constant_(layer.mlp[-1].bias, 0.0)
            layers.append(layer)
        self.gnn.layers = nn.ModuleList(layers)
        self.final_proj = nn.Conv1d(d, d, kernel_size=1, bias=True)
        self.register_parameter("bin_score", nn.Parameter(torch.tensor(1.0)))


class MatchOutputs(dict):
    """Attribute access like the reference's EasyDict (models/superglue_matcher.py:118-127)."""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


class SuperGlueMatch(PicklableModule):
    _TRANSIENT = {"_opack": None, "_mpack": None, "_side": None, "_overflow": None, "_staging": None, "object_means_cache": None}

    def __init__(self, known_classes: List[str], known_colors: List[str], known_words: List[str], args,
                 add_self_loops: bool = True, precision: str = "f16x3"):
        super().__init__()
        self.embed_dim = args.embed_dim
        self.num_layers = args.num_layers
        self.sinkhorn_iters = args.sinkhorn_iters
        self.use_features = args.use_features
        self.args = args
        self.add_self_loops = add_self_loops
        self.precision = precision
        d = self.embed_dim
        self.object_encoder = ObjectEncoder(d, known_classes, known_colors, args)
        self.language_encoder = LanguageEncoder(known_words, d, bi_dir=True)
        self.language_encoder.precision = precision
        self.mlp_offsets = nn.Sequential(nn.Linear(d, d // 2), nn.ReLU(), nn.Linear(d // 2, 2))  # get_mlp_offset (:29-48)
        self.superglue = SuperGlue({"descriptor_dim": d, "GNN_layers": ["self", "cross"] * self.num_layers,
                                    "sinkhorn_iterations": self.sinkhorn_iters, "match_threshold": MATCH_THRESHOLD})
        self._opack = None
        self._mpack = None
        self._side = None
        self._overflow = None   # fp16-range guard word of the object encoder's f16x3 calls (include/t2p.h)

    # ---- cached weight images -------------------------------------------------------------------------------------

    # `precision` also selects the text branch's recurrence: one switch for the whole model, also when it is flipped after
    # construction (bench.py's fp32 pass and the on_overflow="fp32" recomputation do exactly that)
    @property
    def precision(self):
        return self._precision

    @precision.setter
    def precision(self, value):
        if value not in ("f16x3", "fp32"):
            raise ValueError("precision must be 'f16x3' or 'fp32'")
        self._precision = value
        lang = self._modules.get("language_encoder") if "_modules" in self.__dict__ else None
        if lang is not None:
            lang.precision = value
    def _object_pack(self):
        ver = (packing.params_version(self.object_encoder), str(self.device), self.precision)   # (a flipped precision repacks)
        if self._opack is None or self._opack[0] != ver:
            tensors = packing.pack_cell_weights(self, self.device, x3=self.precision == "f16x3")
            self._opack = (ver, tensors, ops.make_cell_weights(tensors))
        return self._opack[2]

    def overflow_detected(self) -> int:
        """Reads (synchronises) and clears the fp16-range guard word of the object encoder (f16x3 only); non-zero = the
        outputs since the last check must not be used: construct the model with precision="fp32"."""
        if self._overflow is None:
            return 0
        code = int(self._overflow.item())
        if code:
            self._overflow.zero_()
        return code

    def _match_pack(self):
        ver = (packing.params_version(self.superglue), packing.params_version(self.mlp_offsets), str(self.device), self.precision)
        if self._mpack is None or self._mpack[0] != ver:
            tensors = packing.pack_match_weights(self, self.device, self.precision)
            self._mpack = (ver, tensors, ops.make_match_weights(tensors))
        return self._mpack[2]

    def _check_forward_only(self):
        if self.training:
            raise NotImplementedError("training-mode forward (batch-statistics BatchNorm) is not built; call .eval()")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError("the HIP path is forward-only; call it under torch.no_grad()")

    # ---- forward ----------------------------------------------------------------------------------------------------
    def forward_packed(self, xyz, rgb, center, mean_rgb, cell_ptr, hints, class_idx=None, color_idx=None,
                       check_overflow=True):
        """Device-resident packed objects (see CellRetrievalNetwork.encode_objects_packed); every sample must hold the
        same number of objects (the dataset pads to args.pad_size, dataloading/kitti360pose/eval.py:147-149) and the
        same number of hints.  hints: List[List[str]], or pre-tokenised (tokens int32 [B * num_hints, T],
        lengths int32 [B * num_hints]) device tensors (modules.tokenize) to keep the host out of the call, or the hint
        encodings themselves ([B, num_hints, D] fp32 from encode_hints: a query's sentences are encoded once although it is
        matched against every one of its retrieved cells)."""
        self._check_forward_only()
        cp = np.ascontiguousarray(np.asarray(cell_ptr), dtype=np.int32)
        b = cp.shape[0] - 1
        sizes = cp[1:] - cp[:-1]
        encoded = isinstance(hints, torch.Tensor)
        tokenised = isinstance(hints, tuple)
        if encoded:
            if hints.dim() != 3 or hints.shape[0] != b or hints.shape[2] != self.embed_dim:
                raise RuntimeError("SuperGlueMatch: hint encodings must be [samples, hints, embed_dim]")
        elif tokenised:
            if hints[0].shape[0] % b != 0:
                raise RuntimeError("SuperGlueMatch: token rows must be a multiple of the number of samples")
        elif len(hints) != b or any(len(h) != len(hints[0]) for h in hints):
            raise RuntimeError("SuperGlueMatch: every sample needs the same number of hints")
        if b < 1 or (sizes != sizes[0]).any():
            raise RuntimeError("SuperGlueMatch: samples must agree in their number of objects and of hints "
                               "(torch.stack / reshape in models/superglue_matcher.py:94-102 need it too)")
        n_obj, d = int(sizes[0]), self.embed_dim
        n_hints = hints.shape[1] if encoded else (hints[0].shape[0] // b if tokenised else len(hints[0]))
        a = self.args
        if bool(getattr(a, "class_embed", False)) != (class_idx is not None) or \
                bool(getattr(a, "color_embed", False)) != (color_idx is not None):
            raise RuntimeError("args.class_embed / args.color_embed need class_idx / color_idx")
        if "color" not in a.use_features and not getattr(a, "class_embed", False):
            rgb = torch.zeros_like(rgb)   # models/object_encoder.py:86-90 (same rule as CellRetrievalNetwork.encode_objects_packed)
        cfg = ops.make_cell_config(n_pts=xyz.shape[1], embed_dim=d, pointnet_features=a.pointnet_features,
                                   use_features=tuple(a.use_features), self_loops=self.add_self_loops,
                                   radius=self.object_encoder.pointnet.radii, precision=self.precision,
                                   class_idx=class_idx, color_idx=color_idx, objects_only=True)
        dev = self.device
        if self.precision == "f16x3":
            if self._overflow is None or self._overflow.device != dev:
                self._overflow = torch.zeros(1, dtype=torch.int32, device=dev)
            cfg.overflow_flag = self._overflow.data_ptr()
        # the hint sentences do not depend on the objects: their (latency-bound) biLSTM runs on a second HIP stream
        # underneath the object encoder
        main = torch.cuda.current_stream(dev)
        if self._side is None:
            self._side = ops.concurrent_stream(dev)
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            if encoded:
                hint = hints.contiguous()
            elif tokenised:
                hint = self.language_encoder.encode_tokens(hints[0], hints[1], normalize=True).view(b, n_hints, d)
            else:
                flat = [s for h in hints for s in h]
                hint = self.language_encoder(flat, normalize=True).view(b, n_hints, d)  # :94-97
        obj = ops.encode_cells(xyz, rgb, center, mean_rgb, cp, torch.from_numpy(cp).to(dev), self._object_pack(), cfg)
        obj = ops.rownorm(obj).view(b, n_obj, d)                                    # F.normalize (:103)
        main.wait_stream(self._side)
        hint.record_stream(main)
        out = ops.match(obj.contiguous(), hint.contiguous(), self._match_pack(), self.sinkhorn_iters, MATCH_THRESHOLD)
        if check_overflow and self.overflow_detected():
            raise FloatingPointError("f16x3 path: an object-encoder activation left fp16's range; construct the model "
                                     "with precision=\"fp32\"")
        return MatchOutputs(P=out["P"], matches0=out["matches0"], matches1=out["matches1"], offsets=out["offsets"],
                            matching_scores0=out["matching_scores0"], matching_scores1=out["matching_scores1"],
                            object_encodings=obj, hint_encodings=hint)

    def encode_hints(self, hints: List[List[str]]) -> torch.Tensor:
        """List[List[str]] (samples x num_hints sentences) -> [samples, num_hints, D] L2-normalised hint encodings
        (models/superglue_matcher.py:94-97)."""
        if any(len(h) != len(hints[0]) for h in hints):
            raise RuntimeError("SuperGlueMatch: every sample needs the same number of hints")
        flat = [s for h in hints for s in h]
        return self.language_encoder(flat, normalize=True).view(len(hints), len(hints[0]), self.embed_dim)

    def forward(self, objects, hints, object_points):
        """objects: List[List[Object3d]] (B samples x pad_size objects), hints: List[List[str]] (B x num_hints),
        object_points: List[Batch] -> outputs with P, matches0, matches1, offsets, matching_scores0/1
        (models/superglue_matcher.py:87-128)."""
        self._check_forward_only()
        n_pts = int(getattr(self.args, "pointnet_numpoints", 256))
        dev = self.device
        if getattr(self, "_staging", None) is None:
            self._staging, self.object_means_cache = HostStaging(), ObjectMeansCache()
        xyz, rgb, center, mean_rgb, cell_ptr = pack_cells(objects, object_points, n_pts, staging=self._staging,
                                                          means_cache=self.object_means_cache,
                                                          skip_rgb="color" not in self.args.use_features, device=dev)
        if rgb is None:
            rgb = torch.zeros_like(xyz)
        to = lambda t: t.to(dev, non_blocking=True)
        oe, class_idx, color_idx = self.object_encoder, None, None
        if getattr(self.args, "class_embed", False):
            class_idx = to(torch.tensor([oe.known_classes.get(o.label, 0) for objs in objects for o in objs],
                                        dtype=torch.int32))
        if getattr(self.args, "color_embed", False):
            color_idx = to(torch.tensor([oe.known_colors[o.get_color_text()] for objs in objects for o in objs],
                                        dtype=torch.int32))
        return self.forward_packed(xyz, rgb, center, mean_rgb, cell_ptr, hints, class_idx, color_idx)

    @property
    def device(self):
        return next(self.mlp_offsets.parameters()).device

    def get_device(self):
        return self.device


def get_pos_in_cell(objects, matches0, offsets) -> np.ndarray:
    """Pose estimate relative to the cell: mean over the matched objects of (object centre xy + offset of its hint);
    the cell centre without matches (models/superglue_matcher.py:139-161)."""
    preds = [objects[o].get_center()[0:2] + offsets[h] for o, h in enumerate(matches0) if h != -1]
    return np.mean(preds, axis=0) if len(preds) > 0 else np.array((0.5, 0.5))

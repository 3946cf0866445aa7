"""Parameter container mirroring models/object_encoder.py::ObjectEncoder (same constructor, same state_dict keys).
Its forward runs inside CellRetrievalNetwork.encode_objects / SuperGlueMatch.forward on the HIP path (t2p_encode_cells in
eval mode, train_cell.py in train mode), including the ground-truth class / colour embedding ablations (`--class_embed`,
`--color_embed`, models/object_encoder.py:74-84,103-120); calling this module directly raises.
"""
from typing import List

import torch
import torch.nn as nn

from .data import COLOR_NAMES
from .modules import get_mlp
from .pointnet2 import PointNet2


class ObjectEncoder(nn.Module):
    def __init__(self, embed_dim: int, known_classes: List[str], known_colors: List[str], args):
        super().__init__()
        self.embed_dim = embed_dim
        self.args = args
        self.known_classes = {c: (i + 1) for i, c in enumerate(known_classes)}
        self.known_classes["<unk>"] = 0
        self.class_embedding = nn.Embedding(len(self.known_classes), embed_dim, padding_idx=0)
        self.known_colors = {c: i for i, c in enumerate(COLOR_NAMES)}
        self.known_colors["<unk>"] = 0
        self.color_embedding = nn.Embedding(len(self.known_colors), embed_dim, padding_idx=0)
        self.pos_encoder = get_mlp([3, 64, embed_dim])
        self.color_encoder = get_mlp([3, 64, embed_dim])
        self.pointnet = PointNet2(len(known_classes), len(known_colors), args)
        path = getattr(args, "pointnet_path", None)
        if path is not None:  # the reference always loads a pre-trained PointNet++ here (object_encoder.py:46)
            self.pointnet.load_state_dict(torch.load(path, map_location="cpu"))
        if getattr(args, "pointnet_freeze", False):
            self.pointnet.requires_grad_(False)
        dim = {0: self.pointnet.dim0, 1: self.pointnet.dim1, 2: self.pointnet.dim2}[args.pointnet_features]
        self.mlp_pointnet = get_mlp([dim, embed_dim])
        self.mlp_merge = get_mlp([len(args.use_features) * embed_dim, embed_dim])

    def forward(self, objects, object_points):
        raise NotImplementedError(
            "ObjectEncoder runs fused inside CellRetrievalNetwork.encode_objects (HIP); it has no stand-alone forward")

    @property
    def device(self):
        return self.class_embedding.weight.device

    def get_device(self):
        return self.device

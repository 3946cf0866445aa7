"""oracle/model.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Eager fp32 CPU restatement of the coarse cell-retrieval forward path, keeping the
reference's execution shape (one PointNet++ forward per cell, BatchNorm un-folded,
torch.nn.LSTM on packed sequences) and its state_dict key layout so that weights
interchange with the product module and with the reference.

Follows (reference file:line):
  mlp()                      models/modules.py:11-36        (Linear, BatchNorm1d, ReLU) per layer, trailing ReLU
  OracleLanguageEncoder      models/modules.py:39-92
  OraclePointNet2            models/pointcloud/pointnet2.py:18-100
  OracleObjectEncoder        models/object_encoder.py:16-142  (default path: class_embed/color_embed False)
  OracleCellRetrieval        models/cell_retrieval.py:23-107
  retrieve_topk_f64          training/coarse.py:100-104,134-140

See oracle/__init__.py for what is pinned by reference execution and what is "parity unpinned".
"""
from types import SimpleNamespace
from typing import List

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import pyg_restated as gnn


def default_args(**kw):
    a = dict(embed_dim=256, use_features=["class", "color", "position"], variation=0, class_embed=False,
             color_embed=False, pointnet_layers=3, pointnet_variation=0, pointnet_numpoints=256,
             pointnet_path=None, pointnet_freeze=False, pointnet_features=2)
    a.update(kw)
    return SimpleNamespace(**a)


def mlp(channels: List[int], add_batchnorm: bool = True) -> nn.Sequential:
    layers = []
    for cin, cout in zip(channels[:-1], channels[1:]):
        if add_batchnorm:
            layers.append(nn.Sequential(nn.Linear(cin, cout), nn.BatchNorm1d(cout), nn.ReLU()))
        else:
            layers.append(nn.Sequential(nn.Linear(cin, cout), nn.ReLU()))
    return nn.Sequential(*layers)


def tokenize(descriptions: List[str], vocab: dict):
    """models/modules.py:60-66 -- strip '.' and ',', lower, split; unknown word -> 0."""
    return [[vocab.get(w, 0) for w in d.replace(".", "").replace(",", "").lower().split()] for d in descriptions]


class OracleLanguageEncoder(nn.Module):
    def __init__(self, known_words, embedding_dim):
        super().__init__()
        self.known_words = {w: i + 1 for i, w in enumerate(known_words)}
        self.known_words["<unk>"] = 0
        self.word_embedding = nn.Embedding(len(self.known_words), embedding_dim, padding_idx=0)
        self.lstm = nn.LSTM(input_size=embedding_dim, hidden_size=embedding_dim, bidirectional=True, num_layers=1)

    def forward(self, descriptions):
        toks = tokenize(descriptions, self.known_words)
        lens = [len(t) for t in toks]
        b, tmax = len(toks), max(lens)
        padded = np.zeros((b, tmax), dtype=np.int64)
        for i, t in enumerate(toks):
            padded[i, : len(t)] = t
        emb = self.word_embedding(torch.from_numpy(padded))
        packed = nn.utils.rnn.pack_padded_sequence(emb, torch.tensor(lens), batch_first=True, enforce_sorted=False)
        d = self.word_embedding.embedding_dim
        h0 = torch.zeros(2, b, d, dtype=emb.dtype)
        c0 = torch.zeros(2, b, d, dtype=emb.dtype)
        _, (h, _) = self.lstm(packed, (h0, c0))
        return torch.mean(h, dim=0)


class _SA(nn.Module):
    def __init__(self, ratio, radius, local_nn, add_self_loops=True):
        super().__init__()
        self.ratio, self.radius = ratio, radius
        self.point_conv = gnn.PointConv(local_nn=local_nn, add_self_loops=add_self_loops)

    def forward(self, x, pos, batch, trace=None):
        idx = gnn.fps(pos, batch, self.ratio)
        cent, dense = gnn.radius(pos, pos[idx], self.radius, batch_x=batch, batch_y=batch[idx])
        edge_index = torch.stack((dense, cent), dim=0)
        out = self.point_conv(x, (pos, pos[idx]), edge_index)
        if trace is not None:
            trace.append(dict(fps=idx.clone(), edge=edge_index.clone(), out=out.detach().clone()))
        return out, pos[idx], batch[idx]


class _GA(nn.Module):
    def __init__(self, net):
        super().__init__()
        self.mlp = net

    def forward(self, x, pos, batch):
        return gnn.global_max_pool(self.mlp(torch.cat((x, pos), dim=1)), batch)


class OraclePointNet2(nn.Module):
    def __init__(self, num_classes, num_colors, add_self_loops=True):
        super().__init__()
        self.sa1 = _SA(0.5, 0.2, mlp([3 + 3, 32, 64]), add_self_loops)
        self.sa2 = _SA(0.5, 0.3, mlp([64 + 3, 128, 128]), add_self_loops)
        self.sa3 = _SA(0.5, 0.4, mlp([128 + 3, 256, 256]), add_self_loops)
        self.ga = _GA(mlp([256 + 3, 512, 1024]))
        self.lin1 = nn.Linear(1024, 512)
        self.lin2 = nn.Linear(512, 256)
        self.class_classifier = nn.Linear(256, num_classes)
        self.color_classifier = nn.Linear(256, num_colors)
        self.dim0, self.dim1, self.dim2 = 1024, 512, 256

    def forward(self, data, trace=None):
        t = [] if trace is not None else None
        x, pos, batch = self.sa1(data.x, data.pos, data.batch, t)
        x, pos, batch = self.sa2(x, pos, batch, t)
        x, pos, batch = self.sa3(x, pos, batch, t)
        f0 = self.ga(x, pos, batch)
        f1 = F.relu(self.lin1(f0))
        f2 = F.relu(self.lin2(f1))
        if trace is not None:
            trace.append(dict(sa=t, features0=f0.detach().clone(), features1=f1.detach().clone(),
                              features2=f2.detach().clone()))
        return SimpleNamespace(features0=f0, features1=f1, features2=f2)


class OracleObjectEncoder(nn.Module):
    def __init__(self, embed_dim, known_classes, known_colors, args, add_self_loops=True):
        super().__init__()
        self.embed_dim, self.args = embed_dim, args
        self.class_embedding = nn.Embedding(len(known_classes) + 1, embed_dim, padding_idx=0)
        # models/object_encoder.py:36-38 -- 8 colour names with "gray" duplicated -> 7 names + "<unk>" = 8 rows
        n_colors = len(set(known_colors)) + 1
        self.color_embedding = nn.Embedding(n_colors, embed_dim, padding_idx=0)
        self.pos_encoder = mlp([3, 64, embed_dim])
        self.color_encoder = mlp([3, 64, embed_dim])
        self.pointnet = OraclePointNet2(len(known_classes), len(known_colors), add_self_loops)
        dim = {0: 1024, 1: 512, 2: 256}[args.pointnet_features]
        self.mlp_pointnet = mlp([dim, embed_dim])
        self.mlp_merge = mlp([len(args.use_features) * embed_dim, embed_dim])

    def forward(self, object_points, mean_rgb, center, trace=None, class_idx=None, color_idx=None):
        """object_points: list (one per cell) of Batch(x=rgb, pos=xyz, batch); mean_rgb/center [Nobj,3]
        (= obj.get_color_rgb() / obj.get_center(), models/object_encoder.py:121-131); class_idx / color_idx [Nobj]
        for the --class_embed / --color_embed ablations (models/object_encoder.py:74-84,103-120)."""
        key = "features%d" % self.args.pointnet_features
        class_embed = bool(getattr(self.args, "class_embed", False))
        color_embed = bool(getattr(self.args, "color_embed", False))
        if not class_embed:  # models/object_encoder.py:86
            if "color" not in self.args.use_features:
                for b in object_points:
                    b.x[:] = 0.0
            feats = [getattr(self.pointnet(b, trace), key) for b in object_points]
            feats = self.mlp_pointnet(torch.cat(feats, dim=0))
        parts = []
        if "class" in self.args.use_features:
            if class_embed:
                parts.append(F.normalize(self.class_embedding(torch.as_tensor(class_idx).long()), dim=-1))
            else:
                parts.append(F.normalize(feats, dim=-1))
        if "color" in self.args.use_features:
            if color_embed:
                parts.append(F.normalize(self.color_embedding(torch.as_tensor(color_idx).long()), dim=-1))
            else:
                parts.append(F.normalize(self.color_encoder(mean_rgb.float()), dim=-1))
        if "position" in self.args.use_features:
            parts.append(F.normalize(self.pos_encoder(center.float()), dim=-1))
        if len(parts) > 1:
            return self.mlp_merge(torch.cat(parts, dim=-1))
        return parts[0]


class OracleCellRetrieval(nn.Module):
    def __init__(self, known_classes, known_colors, known_words, args, add_self_loops=True):
        super().__init__()
        self.embed_dim, self.args, self.variation = args.embed_dim, args, args.variation
        d = args.embed_dim
        aggr = "max" if args.variation == 0 else "mean"
        self.graph1 = gnn.DynamicEdgeConv(mlp([2 * d, d, d]), k=8, aggr=aggr)
        self.lin = mlp([d, d, d])
        self.object_encoder = OracleObjectEncoder(d, known_classes, known_colors, args, add_self_loops)
        self.language_encoder = OracleLanguageEncoder(known_words, d)

    @torch.no_grad()
    def encode_text(self, descriptions):
        return F.normalize(self.language_encoder(descriptions))

    @torch.no_grad()
    def encode_objects_packed(self, xyz, rgb, center, mean_rgb, cell_ptr, trace=None, class_idx=None, color_idx=None):
        return self.encode_objects_packed_grad(xyz, rgb, center, mean_rgb, cell_ptr, trace, class_idx, color_idx)

    def encode_objects_packed_grad(self, xyz, rgb, center, mean_rgb, cell_ptr, trace=None, class_idx=None, color_idx=None):
        """xyz/rgb [Nobj,P,3] fp32 (already FixedPoints+NormalizeScale'd), center/mean_rgb [Nobj,3], cell_ptr [B+1].
        (No no_grad: the training-mode tests differentiate this with torch.autograd, in train() mode.)"""
        xyz, rgb = torch.as_tensor(xyz).float(), torch.as_tensor(rgb).float()
        cell_ptr = [int(v) for v in cell_ptr]
        p = xyz.shape[1]
        batches, batch = [], []
        for c in range(len(cell_ptr) - 1):
            lo, hi = cell_ptr[c], cell_ptr[c + 1]
            n = hi - lo
            batches.append(gnn.Batch(x=rgb[lo:hi].reshape(n * p, 3).clone(), pos=xyz[lo:hi].reshape(n * p, 3).clone(),
                                     batch=torch.arange(n).repeat_interleave(p)))
            batch += [c] * n
        batch = torch.tensor(batch, dtype=torch.long)
        emb = self.object_encoder(batches, torch.as_tensor(mean_rgb), torch.as_tensor(center), trace, class_idx, color_idx)
        if trace is not None:
            trace.append(dict(object_embeddings=emb.detach().clone()))
        emb = F.normalize(emb, dim=-1)
        x = self.graph1(emb, batch)
        x = gnn.global_max_pool(x, batch) if self.variation == 0 else gnn.global_mean_pool(x, batch)
        x = self.lin(x)
        return F.normalize(x)


def randomize_bn_stats(model: nn.Module, seed: int = 4321):
    """randomise BN running stats / affine so that BN folding is exercised."""
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, nn.BatchNorm1d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.weight.data.copy_(1.0 + 0.2 * torch.randn(m.num_features, generator=g))
            m.bias.data.copy_(0.1 * torch.randn(m.num_features, generator=g))


def retrieve_topk_f64(cell_enc: np.ndarray, text_enc: np.ndarray, k: int):
    """training/coarse.py:100-104,134-140: float64 arrays, per query `scores = C @ q`,
    `argsort(-scores)[:k]`; pinned: stable sort, i.e. ties -> lower cell index."""
    c = np.zeros(cell_enc.shape, dtype=np.float64)
    c[:] = cell_enc
    q = np.zeros(text_enc.shape, dtype=np.float64)
    q[:] = text_enc
    idx = np.zeros((q.shape[0], k), dtype=np.int64)
    sc = np.zeros((q.shape[0], k), dtype=np.float64)
    for i in range(q.shape[0]):
        scores = c[:] @ q[i]
        order = np.argsort(-1.0 * scores, kind="stable")[:k]
        idx[i, : len(order)] = order
        sc[i, : len(order)] = scores[order]
    return idx, sc

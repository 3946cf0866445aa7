"""TEST INFRASTRUCTURE -- CPU restatement of the reference's fine stage (hint <-> object matching + offset regression).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this package; the product path
(text2pos-cvpr2022_amd/) never does.

Follows (reference file:line, relative to /root/reference):
  SuperGlueMatch.forward            models/superglue_matcher.py:87-128
  get_mlp_offset                    models/superglue_matcher.py:29-48   (Linear, ReLU, Linear: no BN, no trailing ReLU)
  attention / MultiHeadedAttention  models/superglue.py:90-116          (4 heads, channel c = d * heads + h)
  AttentionalPropagation / GNN      models/superglue.py:119-146         (Conv1d k=1 == per-token Linear; BN in eval mode)
  log_sinkhorn_iterations / OT      models/superglue.py:149-177
  SuperGlue.forward                 models/superglue.py:239-330         (final_proj, scores / sqrt(D), OT, mutual NN + 0.2)
  get_pos_in_cell                   models/superglue_matcher.py:139-161

Pinned against the reference itself: tests/golden/make_golden.py executes the reference's own models.superglue.SuperGlue
(pure torch, importable here) and SuperGlueMatch.forward glue and stores inputs/outputs in tests/golden/fine.npz;
tests/test_oracle.py checks this restatement against that fixture.  Tokens-major layout [B, N, D] (the reference keeps
[B, D, N]); parameters are nn.Linear, convertible from the reference's Conv1d state_dict with `load_reference_state`.
"""
from typing import List

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .model import OracleLanguageEncoder, OracleObjectEncoder
from . import pyg_restated as gnn

NUM_HEADS = 4
MATCH_THRESHOLD = 0.2


class Propagation(nn.Module):
    def __init__(self, d: int):
        super().__init__()
        self.proj = nn.ModuleList([nn.Linear(d, d) for _ in range(3)])  # query, key, value
        self.merge = nn.Linear(d, d)
        self.mlp0 = nn.Linear(2 * d, 2 * d)
        self.bn = nn.BatchNorm1d(2 * d)
        self.mlp3 = nn.Linear(2 * d, d)

    def forward(self, x, source):
        """x [B, N, D], source [B, M, D] -> delta [B, N, D]"""
        b, n, d = x.shape
        m = source.shape[1]
        dh = d // NUM_HEADS
        # channel c of a projection belongs to head c % heads, position c // heads inside the head
        q = self.proj[0](x).view(b, n, dh, NUM_HEADS)
        k = self.proj[1](source).view(b, m, dh, NUM_HEADS)
        v = self.proj[2](source).view(b, m, dh, NUM_HEADS)
        scores = torch.einsum("bndh,bmdh->bhnm", q, k) / dh ** 0.5
        prob = F.softmax(scores, dim=-1)
        msg = torch.einsum("bhnm,bmdh->bndh", prob, v).reshape(b, n, d)
        msg = self.merge(msg)
        h = self.mlp0(torch.cat([x, msg], dim=-1))
        h = F.relu(self.bn(h.transpose(1, 2)).transpose(1, 2))
        return self.mlp3(h)


def log_optimal_transport(scores, alpha, iters: int):
    """scores [B, M, N] -> log couplings [B, M+1, N+1] (models/superglue.py:149-177)"""
    b, m, n = scores.shape
    one = scores.new_tensor(1.0)
    ms, ns = m * one, n * one
    z = torch.cat([torch.cat([scores, alpha.expand(b, m, 1)], -1), torch.cat([alpha.expand(b, 1, n), alpha.expand(b, 1, 1)], -1)], 1)
    norm = -(ms + ns).log()
    log_mu = torch.cat([norm.expand(m), ns.log()[None] + norm])[None].expand(b, -1)
    log_nu = torch.cat([norm.expand(n), ms.log()[None] + norm])[None].expand(b, -1)
    u, v = torch.zeros_like(log_mu), torch.zeros_like(log_nu)
    for _ in range(iters):
        u = log_mu - torch.logsumexp(z + v.unsqueeze(1), dim=2)
        v = log_nu - torch.logsumexp(z + u.unsqueeze(2), dim=1)
    return z + u.unsqueeze(2) + v.unsqueeze(1) - norm


class OracleSuperGlue(nn.Module):
    def __init__(self, d: int, num_layers: int, sinkhorn_iters: int):
        super().__init__()
        self.d, self.iters = d, sinkhorn_iters
        self.names = ["self", "cross"] * num_layers
        self.layers = nn.ModuleList([Propagation(d) for _ in self.names])
        self.final_proj = nn.Linear(d, d)
        self.bin_score = nn.Parameter(torch.tensor(1.0))

    def forward(self, desc0, desc1):
        """desc0 [B, M, D] (objects), desc1 [B, N, D] (hints)"""
        for layer, name in zip(self.layers, self.names):
            s0, s1 = (desc1, desc0) if name == "cross" else (desc0, desc1)
            d0, d1 = layer(desc0, s0), layer(desc1, s1)
            desc0, desc1 = desc0 + d0, desc1 + d1
        m0, m1 = self.final_proj(desc0), self.final_proj(desc1)
        scores = torch.einsum("bnd,bmd->bnm", m0, m1) / self.d ** 0.5
        z = log_optimal_transport(scores, self.bin_score, self.iters)
        p = torch.exp(z)
        inner = z[:, :-1, :-1]
        max0, max1 = inner.max(2), inner.max(1)
        i0, i1 = max0.indices, max1.indices
        ar0 = torch.arange(i0.shape[1])[None]
        ar1 = torch.arange(i1.shape[1])[None]
        mutual0 = ar0 == i1.gather(1, i0)
        mutual1 = ar1 == i0.gather(1, i1)
        zero = z.new_tensor(0.0)
        ms0 = torch.where(mutual0, max0.values.exp(), zero)
        ms1 = torch.where(mutual1, ms0.gather(1, i1), zero)
        valid0 = mutual0 & (ms0 > MATCH_THRESHOLD)
        valid1 = mutual1 & valid0.gather(1, i1)
        return dict(matches0=torch.where(valid0, i0, i0.new_tensor(-1)), matches1=torch.where(valid1, i1, i1.new_tensor(-1)),
                    matching_scores0=ms0, matching_scores1=ms1, P=p)

    def load_reference_state(self, sd: dict, prefix: str = "superglue."):
        """Copy a reference state_dict (Conv1d weights [O, I, 1]) into this module."""
        def lin(dst, key):
            dst.weight.data.copy_(sd[prefix + key + ".weight"].squeeze(-1))
            dst.bias.data.copy_(sd[prefix + key + ".bias"])
        for i, layer in enumerate(self.layers):
            base = f"gnn.layers.{i}."
            for j in range(3):
                lin(layer.proj[j], base + f"attn.proj.{j}")
            lin(layer.merge, base + "attn.merge")
            lin(layer.mlp0, base + "mlp.0")
            for name in ("weight", "bias", "running_mean", "running_var"):
                getattr(layer.bn, name).data.copy_(sd[prefix + base + "mlp.1." + name])
            lin(layer.mlp3, base + "mlp.3")
        lin(self.final_proj, "final_proj")
        self.bin_score.data.copy_(sd[prefix + "bin_score"])


class OracleSuperGlueMatch(nn.Module):
    def __init__(self, known_classes, known_colors, known_words, args, add_self_loops=True):
        super().__init__()
        d = args.embed_dim
        self.embed_dim, self.args = d, args
        self.object_encoder = OracleObjectEncoder(d, known_classes, known_colors, args, add_self_loops)
        self.language_encoder = OracleLanguageEncoder(known_words, d)
        self.mlp_offsets = nn.Sequential(nn.Linear(d, d // 2), nn.ReLU(), nn.Linear(d // 2, 2))
        self.superglue = OracleSuperGlue(d, args.num_layers, args.sinkhorn_iters)

    @torch.no_grad()
    def forward_packed(self, xyz, rgb, center, mean_rgb, cell_ptr, hints: List[List[str]]):
        """xyz/rgb [B*n, P, 3], center/mean_rgb [B*n, 3], cell_ptr [B+1] (n objects per sample, all samples alike),
        hints: B lists of hint sentences."""
        xyz, rgb = torch.as_tensor(xyz).float(), torch.as_tensor(rgb).float()
        cell_ptr = [int(v) for v in cell_ptr]
        b, p = len(cell_ptr) - 1, xyz.shape[1]
        batches = []
        for c in range(b):
            lo, hi = cell_ptr[c], cell_ptr[c + 1]
            n = hi - lo
            batches.append(gnn.Batch(x=rgb[lo:hi].reshape(n * p, 3).clone(), pos=xyz[lo:hi].reshape(n * p, 3).clone(),
                                     batch=torch.arange(n).repeat_interleave(p)))
        hint_enc = F.normalize(torch.stack([self.language_encoder(h) for h in hints]), dim=-1)      # [B, H, D]
        obj = self.object_encoder(batches, torch.as_tensor(mean_rgb), torch.as_tensor(center))
        obj = F.normalize(obj.reshape(b, -1, self.embed_dim), dim=-1)                               # [B, n, D]
        out = self.superglue(obj, hint_enc)
        out["offsets"] = self.mlp_offsets(hint_enc)
        out["object_encodings"], out["hint_encodings"] = obj, hint_enc
        return out


def get_pos_in_cell(centers_xy: np.ndarray, matches0: np.ndarray, offsets: np.ndarray) -> np.ndarray:
    """models/superglue_matcher.py:139-161: mean over matched objects of (object centre xy + offset of its hint);
    (0.5, 0.5) without matches.  centers_xy [n, 2] = obj.get_center()[0:2]."""
    preds = [centers_xy[o] + offsets[h] for o, h in enumerate(matches0) if h != -1]
    return np.mean(preds, axis=0) if len(preds) > 0 else np.array((0.5, 0.5))

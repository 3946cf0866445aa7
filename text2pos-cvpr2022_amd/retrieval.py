"""Top-k cell retrieval: the replacement for the per-query NumPy loop of training/coarse.py:134-140.

`retrieve_topk(cell_encodings, text_encodings, k)` returns what the reference computes per query as
`np.argsort(-1.0 * (cell_encodings @ text_encodings[q]))[0:k]` -- for all queries at once, ranked in float64 on the
GPU (csrc/sim_topk.hip), ties resolved to the lower cell index.
"""
import numpy as np
import torch

from . import ops


def _as_device_f32(x, device):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    x = x.to(device=device)
    if x.dtype != torch.float32:
        # the reference stores fp32 model outputs in float64 arrays (training/coarse.py:100-116); the values are
        # still exactly representable in fp32, anything else would silently lose precision here
        x32 = x.to(torch.float32)
        if not torch.equal(x32.to(x.dtype), x):
            raise RuntimeError("retrieve_topk: encodings are not exactly representable in float32")
        x = x32
    return x.contiguous()


def retrieve_topk(cell_encodings, text_encodings, k: int, device=None, index_offset: int = 0):
    """cell_encodings [Nc, D], text_encodings [Nq, D] (torch or numpy) -> (indices int64 [Nq, k], scores f64 [Nq, k])
    as torch tensors on the GPU."""
    if device is None:
        device = cell_encodings.device if isinstance(cell_encodings, torch.Tensor) and cell_encodings.is_cuda else \
            torch.device("cuda", torch.cuda.current_device())
    c = _as_device_f32(cell_encodings, device)
    q = _as_device_f32(text_encodings, device)
    d = c.shape[1]
    if d not in (128, 256, 384):     # e.g. embed_dim 300: zero columns up to the kernel's width add exact zeros to every score
        from .packing import kernel_embed_dim
        pad = kernel_embed_dim(d) - d
        c, q = torch.nn.functional.pad(c, (0, pad)).contiguous(), torch.nn.functional.pad(q, (0, pad)).contiguous()
    return ops.sim_topk(q, c, int(k), index_offset)


def top_retrievals(cell_encodings, text_encodings, db_cell_ids, top_k):
    """{query_idx: retrieved cell ids} exactly as eval_epoch builds it (training/coarse.py:133-147)."""
    idx, _ = retrieve_topk(cell_encodings, text_encodings, int(np.max(top_k)))
    idx = idx.cpu().numpy()
    ids = np.asarray(db_cell_ids)
    return {q: ids[idx[q]] for q in range(idx.shape[0])}

"""Tensor-level wrappers over the C ABI (include/t2p.h): argument validation, workspace handling, stream plumbing.

PyTorch is used for device memory, streams and torch.distributed only; all arithmetic happens in libt2p_hip.so.
Shape / dtype / device / contiguity violations raise RuntimeError before anything is launched.
"""
import ctypes as C
import os
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib as L

_workspaces: Dict[tuple, torch.Tensor] = {}


def _stream(device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def _need(t: torch.Tensor, name: str, dtype, ndim=None, device=None):
    if not isinstance(t, torch.Tensor):
        raise RuntimeError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: must live on the GPU (got device {t.device}); there is no CPU path")
    if t.dtype != dtype:
        raise RuntimeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: must be contiguous")
    if ndim is not None and t.dim() != ndim:
        raise RuntimeError(f"{name}: expected {ndim} dimensions, got shape {tuple(t.shape)}")
    if device is not None and t.device != device:
        raise RuntimeError(f"{name}: on {t.device}, expected {device}")
    return t


DEFAULT_CHUNK_OBJECTS = 65000   # include/t2p.h: T2P_DEFAULT_CHUNK_OBJECTS


def workspace(device, nbytes: int, tag: str) -> torch.Tensor:
    """Grow-only scratch buffer per (device, tag), obtained from torch's caching allocator."""
    key = (str(device), tag)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        _workspaces[key] = None
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def release_workspaces():
    _workspaces.clear()


# ---- streams that really run side by side ----------------------------------------------------------------------
def _spin_cycles(device) -> int:
    """Argument of torch.cuda._sleep for a ~0.4 ms single-thread spin on this device (its unit is device dependent)."""
    key = str(device)
    if key not in _spin_memo:
        st = torch.cuda.current_stream(device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(1000)             # (first launch: module load)
        torch.cuda.synchronize(device)
        cycles = 20000
        for _ in range(8):
            e0.record(st)
            torch.cuda._sleep(cycles)
            e1.record(st)
            e1.synchronize()
            ms = e0.elapsed_time(e1)
            if 0.3 <= ms <= 0.8:
                break
            cycles = max(1000, int(cycles * min(8.0, 0.45 / max(ms, 1e-3))))
        _spin_memo[key] = cycles
    return _spin_memo[key]


def _measure_overlap(a: "torch.cuda.Stream", b: "torch.cuda.Stream") -> bool:
    dev = a.device
    cycles = _spin_cycles(dev)
    torch.cuda.synchronize(dev)
    start, ea, eb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    start.record(a)
    b.wait_event(start)
    with torch.cuda.stream(a):
        torch.cuda._sleep(cycles)
        ea.record(a)
    with torch.cuda.stream(b):
        torch.cuda._sleep(cycles)
        eb.record(b)
    ea.synchronize()
    eb.synchronize()
    solo, both = start.elapsed_time(ea), max(start.elapsed_time(ea), start.elapsed_time(eb))
    return both < 1.6 * solo


def streams_overlap(a: "torch.cuda.Stream", b: "torch.cuda.Stream") -> bool:
    """True when a kernel on stream `b` runs BESIDE a kernel on stream `a`.  HIP maps a process's streams onto a handful of hardware
    queues (4 by default) in the order of their first use; two streams that share a queue serialise, and nothing in the API says which
    do.  Measured with a single-thread spin kernel (~0.45 ms) per stream behind a common start event: side by side the pair takes one
    spin time, in one queue two.  A stream keeps its queue, so a pair is measured once per process.  (A 12,000-cell encode in two parts
    loses its whole two-stream gain, ~3 %, when the second part's stream shares the text stream's queue - which depends on what the
    process did before: docs/notebook.md, round 6.)"""
    if a.cuda_stream == b.cuda_stream:
        return False
    key = (str(a.device),) + tuple(sorted((a.cuda_stream, b.cuda_stream)))
    if key not in _overlap_memo:
        _overlap_memo[key] = _measure_overlap(a, b)
    return _overlap_memo[key]


def concurrent_stream(device, beside=()) -> "torch.cuda.Stream":
    """A HIP stream that runs beside the device's current stream and beside every stream in `beside` (streams_overlap), and - as far as
    the hardware queues allow - beside the streams this function handed out before.  Streams come from a per-device set of at most
    eight that lives as long as the process (a stream's queue is fixed at its first use; a destroyed stream's slot would be handed
    out again), so two callers may get the same stream - which orders their work, never breaks it.  Cost: ~1 ms per pair of streams
    never measured before.  Plain `torch.cuda.Stream()` when the spin kernel is unavailable, under a graph capture (the probe
    synchronises the device) or with T2P_NO_STREAM_PROBE set (the switch exists for A/B measurements of this function)."""
    device = torch.device(device)
    if not hasattr(torch.cuda, "_sleep") or os.environ.get("T2P_NO_STREAM_PROBE") or torch.cuda.is_current_stream_capturing():
        return torch.cuda.Stream(device=device)
    must = [torch.cuda.current_stream(device)] + [b for b in beside if b is not None]
    kept = _stream_sets.setdefault(str(device), [])
    best, best_score, i = None, None, 0
    while i < _MAX_KEPT_STREAMS:
        if i == len(kept):
            c = torch.cuda.Stream(device=device)
            if any(c.cuda_stream == o.cuda_stream for o in kept):
                break                      # (torch's stream pool came round)
            kept.append(c)
        c, i = kept[i], i + 1
        if any(c.cuda_stream == m.cuda_stream for m in must) or not all(streams_overlap(m, c) for m in must):
            continue
        out = [o for o in kept if _handed_out.get((str(device), o.cuda_stream), 0) > 0 and all(o.cuda_stream != m.cuda_stream for m in must)]
        score = sum(_handed_out[(str(device), o.cuda_stream)] for o in out if not streams_overlap(o, c))   # (c itself included)
        if best is None or score < best_score:
            best, best_score = c, score
        if score == 0:
            break
    if best is None:
        return torch.cuda.Stream(device=device)
    _handed_out[(str(device), best.cuda_stream)] = _handed_out.get((str(device), best.cuda_stream), 0) + 1
    return best


def pick_concurrent_streams(device, beside, n: int):
    """`n` streams from concurrent_stream, each beside `beside` and beside the ones picked before it."""
    picked = []
    for _ in range(n):
        picked.append(concurrent_stream(device, list(beside) + picked))
    return picked


_MAX_KEPT_STREAMS = 8
_overlap_memo, _stream_sets, _handed_out, _spin_memo = {}, {}, {}, {}


# ---------------------------------------------------------------------------------------------------------------
def sample_group(xyz: torch.Tensor, radius=(0.2, 0.3, 0.4)):
    """Fused FPS + ball query of the three SA levels.  xyz [n_obj, n_pts, 3] fp32.
    Returns dict(fps_idx=[3 x uint8 [n_obj, n_c]], nbr=[3 x uint8 [n_obj, n_c, 32]], cnt=[3 x uint8 [n_obj, n_c]])."""
    _need(xyz, "xyz", torch.float32, 3)
    n_obj, n_pts, three = xyz.shape
    if three != 3:
        raise RuntimeError(f"xyz: last dimension must be 3, got {three}")
    nd, out = n_pts, dict(fps_idx=[], nbr=[], cnt=[])
    for _ in range(3):
        nc = (nd + 1) // 2
        out["fps_idx"].append(torch.empty((n_obj, nc), dtype=torch.uint8, device=xyz.device))
        out["nbr"].append(torch.empty((n_obj, nc, 32), dtype=torch.uint8, device=xyz.device))
        out["cnt"].append(torch.empty((n_obj, nc), dtype=torch.uint8, device=xyz.device))
        nd = nc
    arr = lambda ts: (C.c_void_p * 3)(*[t.data_ptr() for t in ts])
    r = (C.c_float * 3)(*[float(v) for v in radius])
    L.check(L.lib().t2p_sample_group(_ptr(xyz), n_obj, n_pts, r, arr(out["fps_idx"]), arr(out["nbr"]), arr(out["cnt"]),
                                     _stream(xyz.device)), "t2p_sample_group")
    return out


def group_edges(xyz: torch.Tensor, first_obj: torch.Tensor, radius=(0.2, 0.3, 0.4), self_loops: bool = True):
    """FPS + ball query of the three SA levels as EDGE LISTS for the training-mode path, built on the device:
    t2p_group_rows (k_sample_group's compact row lists) -> t2p_edge_counts -> prefix sum -> t2p_edge_expand.  One host
    read-back (the three edge totals) instead of the size read-backs of a nonzero / sort / bincount chain per level.
    xyz [n_obj, n_pts, 3] fp32; first_obj [n_obj] int32 (device): first object of each object's cell.
    Returns a list of three dicts: fps_idx uint8 [n_obj, n_c], src / dst int32 [E] (dense row, centroid row; sorted by dst,
    torch_geometric's self-loop rewrite applied when self_loops), cent_ptr int32 [n_obj * n_c + 1]."""
    _need(xyz, "xyz", torch.float32, 3)
    dev = xyz.device
    _need(first_obj, "first_obj", torch.int32, 1, dev)
    n_obj, n_pts, three = xyz.shape
    if three != 3 or first_obj.numel() != n_obj:
        raise RuntimeError("group_edges: xyz must be [n_obj, n_pts, 3] and first_obj [n_obj]")
    nds, ncs, fps, rows, n_rows = [], [], [], [], []
    nd = n_pts
    for _ in range(3):
        nc = (nd + 1) // 2
        nds.append(nd)
        ncs.append(nc)
        fps.append(torch.empty((n_obj, nc), dtype=torch.uint8, device=dev))
        rows.append(torch.empty((n_obj, nc * 33), dtype=torch.int16, device=dev))
        n_rows.append(torch.empty((n_obj,), dtype=torch.int16, device=dev))
        nd = nc
    arr = lambda ts: (C.c_void_p * 3)(*[t.data_ptr() for t in ts])
    r = (C.c_float * 3)(*[float(v) for v in radius])
    st = _stream(dev)
    L.check(L.lib().t2p_group_rows(_ptr(xyz), n_obj, n_pts, r, int(bool(self_loops)), arr(fps), arr(rows), arr(n_rows), st),
            "t2p_group_rows")
    cent_ptrs = []
    for l in range(3):
        counts = torch.empty((n_obj * ncs[l],), dtype=torch.int32, device=dev)
        L.check(L.lib().t2p_edge_counts(_ptr(rows[l]), _ptr(n_rows[l]), _ptr(first_obj), n_obj, nds[l], ncs[l],
                                        int(bool(self_loops)), _ptr(counts), st), "t2p_edge_counts")
        cp = torch.zeros((n_obj * ncs[l] + 1,), dtype=torch.int32, device=dev)
        torch.cumsum(counts, 0, dtype=torch.int32, out=cp[1:])
        cent_ptrs.append(cp)
    totals = torch.stack([cp[-1] for cp in cent_ptrs]).cpu().tolist()      # the one read-back
    out = []
    for l in range(3):
        e = int(totals[l])
        src = torch.empty((e,), dtype=torch.int32, device=dev)
        dst = torch.empty((e,), dtype=torch.int32, device=dev)
        L.check(L.lib().t2p_edge_expand(_ptr(rows[l]), _ptr(n_rows[l]), _ptr(first_obj), _ptr(cent_ptrs[l]), n_obj, nds[l], ncs[l],
                                        int(bool(self_loops)), _ptr(src), _ptr(dst), st), "t2p_edge_expand")
        out.append(dict(fps_idx=fps[l], src=src, dst=dst, cent_ptr=cent_ptrs[l], n_dense=nds[l], n_cent=ncs[l]))
    return out


def dedup_rows(xyz: torch.Tensor, rgb: torch.Tensor, rows: torch.Tensor, n_rows: torch.Tensor):
    """In-place t2p_dedup_rows: rows uint16 [n_obj, (n_pts / 2) * 33] (as int16 storage), n_rows uint16 [n_obj] (int16)."""
    _need(xyz, "xyz", torch.float32, 3)
    _need(rgb, "rgb", torch.float32, 3, xyz.device)
    _need(rows, "rows", torch.int16, 2, xyz.device)
    _need(n_rows, "n_rows", torch.int16, 1, xyz.device)
    n_obj, n_pts = xyz.shape[0], xyz.shape[1]
    if rows.shape[0] != n_obj or rows.shape[1] != ((n_pts + 1) // 2) * 33 or n_rows.numel() != n_obj:
        raise RuntimeError("dedup_rows: inconsistent shapes")
    L.check(L.lib().t2p_dedup_rows(_ptr(xyz), _ptr(rgb), n_obj, n_pts, _ptr(rows), _ptr(n_rows), _stream(xyz.device)),
            "t2p_dedup_rows")


def knn(x: torch.Tensor, seg_ptr: torch.Tensor, k: int, max_seg_rows: Optional[int] = None) -> torch.Tensor:
    _need(x, "x", torch.float32, 2)
    _need(seg_ptr, "seg_ptr", torch.int32, 1, x.device)
    if max_seg_rows is None:
        sp = seg_ptr.cpu()
        max_seg_rows = int((sp[1:] - sp[:-1]).max().item()) if sp.numel() > 1 else 0
    out = torch.empty((x.shape[0], k), dtype=torch.int32, device=x.device)
    L.check(L.lib().t2p_knn(_ptr(x), x.shape[1], _ptr(seg_ptr), seg_ptr.numel() - 1, int(max_seg_rows), k, _ptr(out),
                            _stream(x.device)), "t2p_knn")
    return out


def gemm(a: torch.Tensor, w_kmajor: torch.Tensor, bias: Optional[torch.Tensor] = None, relu: bool = False):
    """act(a @ w_kmajor + bias);  a [M, K], w_kmajor [K, N]."""
    _need(a, "a", torch.float32, 2)
    _need(w_kmajor, "w", torch.float32, 2, a.device)
    if bias is not None:
        _need(bias, "bias", torch.float32, 1, a.device)
    m, k = a.shape
    if w_kmajor.shape[0] != k:
        raise RuntimeError(f"gemm: a is [{m},{k}] but w is {tuple(w_kmajor.shape)}")
    n = w_kmajor.shape[1]
    out = torch.empty((m, n), dtype=torch.float32, device=a.device)
    L.check(L.lib().t2p_gemm(_ptr(a), k, _ptr(w_kmajor), _ptr(bias), _ptr(out), n, 0, m, k, n, int(relu),
                             _stream(a.device)), "t2p_gemm")
    return out


def matmul(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a [M, K] @ b [K, N] on the tiled fp32-MFMA GEMM (t2p_gemm) for any sizes: K is zero-padded to a multiple of 4 and N
    to a multiple of 8 (the kernel's granules).  The training path's weight-gradient / score products go through here:
    a transposed operand is a torch copy (`x.t().contiguous()`), the arithmetic is this library's."""
    _need(a.contiguous(), "a", torch.float32, 2)
    m, k = a.shape
    if b.shape[0] != k:
        raise RuntimeError(f"matmul: a is [{m},{k}] but b is {tuple(b.shape)}")
    n = b.shape[1]
    kp, np_ = (k + 3) // 4 * 4, (n + 7) // 8 * 8
    a = a.contiguous()
    b = b.contiguous()
    if kp != k:
        a = torch.nn.functional.pad(a, (0, kp - k))
    if kp != k or np_ != n:
        b = torch.nn.functional.pad(b, (0, np_ - n, 0, kp - k))
    if m == 0 or n == 0:
        return torch.zeros((m, n), dtype=torch.float32, device=a.device)
    out = gemm(a.contiguous(), b.contiguous())
    return out if np_ == n else out[:, :n].contiguous()


def gemm_tn(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a[M, K1]^T @ b[M, N] -> [K1, N] (t2p_gemm_tn): rows split over the grid, partials added in a fixed order.  The
    weight-gradient products of the training-mode path (dW = dY^T X, ...): no transposed copy, no library GEMM."""
    _need(a, "a", torch.float32, 2)
    _need(b, "b", torch.float32, 2, a.device)
    m, k1 = a.shape
    if b.shape[0] != m:
        raise RuntimeError(f"gemm_tn: a is [{m},{k1}] but b is {tuple(b.shape)}")
    n = b.shape[1]
    out = torch.empty((k1, n), dtype=torch.float32, device=a.device)
    if m == 0:
        return out.zero_()
    ws = torch.empty((L.lib().t2p_gemm_tn_workspace_bytes(m, k1, n),), dtype=torch.uint8, device=a.device)
    L.check(L.lib().t2p_gemm_tn(_ptr(a), k1, _ptr(b), n, _ptr(out), n, m, k1, n, _ptr(ws), ws.numel(), _stream(a.device)),
            "t2p_gemm_tn")
    return out


def linear_wgrad(dy: torch.Tensor, x: torch.Tensor, want_colsum: bool = True):
    """(dy[M, K1]^T @ x[M, N] -> [K1, N], column sums of dy [K1] or None) in one pass over the rows (t2p_linear_wgrad_f32): the
    weight and bias gradients of a Linear.  Row pitches must be multiples of 4 floats (the layers of the path are)."""
    _need(dy, "dy", torch.float32, 2)
    _need(x, "x", torch.float32, 2, dy.device)
    m, k1 = dy.shape
    if x.shape[0] != m:
        raise RuntimeError(f"linear_wgrad: dy is [{m},{k1}] but x is {tuple(x.shape)}")
    n = x.shape[1]
    if k1 % 4:
        dy = torch.nn.functional.pad(dy, (0, (-k1) % 4)).contiguous()
    if n % 4:
        x = torch.nn.functional.pad(x, (0, (-n) % 4)).contiguous()
    out = torch.empty((k1, n), dtype=torch.float32, device=dy.device)
    colsum = torch.empty((k1,), dtype=torch.float32, device=dy.device) if want_colsum else None
    ws = torch.empty((L.lib().t2p_linear_wgrad_workspace_bytes(m, k1, n),), dtype=torch.uint8, device=dy.device)
    L.check(L.lib().t2p_linear_wgrad_f32(_ptr(dy), dy.shape[1], _ptr(x), x.shape[1], _ptr(out), n, _ptr(colsum), m, k1, n, _ptr(ws),
                                         ws.numel(), _stream(dy.device)), "t2p_linear_wgrad_f32")
    return out, colsum


def rownorm(x: torch.Tensor) -> torch.Tensor:
    _need(x, "x", torch.float32, 2)
    out = torch.empty_like(x)
    L.check(L.lib().t2p_rownorm(_ptr(x), x.shape[0], x.shape[1], _ptr(out), _stream(x.device)), "t2p_rownorm")
    return out


def sim_topk(queries: torch.Tensor, cells: torch.Tensor, k: int, index_offset: int = 0):
    """Float64 cosine scores + ordered top-k (ties -> lower index).  Returns (idx int64 [nq,k], score f64 [nq,k])."""
    _need(queries, "queries", torch.float32, 2)
    _need(cells, "cells", torch.float32, 2, queries.device)
    nq, dim = queries.shape
    nc = cells.shape[0]
    if cells.shape[1] != dim:
        raise RuntimeError(f"sim_topk: queries have dim {dim}, cells {cells.shape[1]}")
    dev = queries.device
    idx = torch.empty((nq, k), dtype=torch.int64, device=dev)
    score = torch.empty((nq, k), dtype=torch.float64, device=dev)
    nbytes = L.lib().t2p_sim_topk_workspace_bytes(nq, nc, k)
    ws = workspace(dev, nbytes, "sim_topk")
    L.check(L.lib().t2p_sim_topk(_ptr(queries), _ptr(cells), nq, nc, dim, k, int(index_offset), _ptr(idx), _ptr(score),
                                 _ptr(ws), ws.numel(), _stream(dev)), "t2p_sim_topk")
    return idx, score


# ---------------------------------------------------------------------------------------------------------------
def make_cell_config(n_pts=256, embed_dim=256, pointnet_features=2, use_features=("class", "color", "position"),
                     self_loops=True, knn_k=8, variation=0, radius=(0.2, 0.3, 0.4), chunk_objects=0,
                     precision="f16x3", class_idx=None, color_idx=None, objects_only=False,
                     overflow_flag=None, tuning=0) -> L.CellConfig:
    """class_idx / color_idx: int32 device tensors [n_obj] enabling the --class_embed / --color_embed ablations.
    overflow_flag: int32 device tensor [1], the sticky fp16-range guard word of the f16x3 path (include/t2p.h)."""
    cfg = L.CellConfig()
    cfg.n_pts, cfg.embed_dim, cfg.pointnet_features = int(n_pts), int(embed_dim), int(pointnet_features)
    cfg.use_class = int("class" in use_features)
    cfg.use_color = int("color" in use_features)
    cfg.use_position = int("position" in use_features)
    cfg.self_loops, cfg.knn_k, cfg.variation = int(bool(self_loops)), int(knn_k), int(variation)
    cfg.radius = (C.c_float * 3)(*[float(r) for r in radius])
    cfg.chunk_objects = int(chunk_objects)
    cfg.objects_only = int(bool(objects_only))
    if precision not in ("fp32", "f16x3"):
        raise RuntimeError(f"precision must be 'fp32' or 'f16x3', got {precision!r}")
    cfg.precision = 1 if precision == "f16x3" else 0
    for name, t in (("class", class_idx), ("color", color_idx)):
        if t is not None:
            _need(t, name + "_idx", torch.int32, 1)
            setattr(cfg, name + "_embed", 1)
            setattr(cfg, name + "_idx", t.data_ptr())
    cfg.tuning = int(tuning)
    if overflow_flag is not None:
        _need(overflow_flag, "overflow_flag", torch.int32, 1)
        cfg.overflow_flag = overflow_flag.data_ptr()
    cfg._keepalive = (class_idx, color_idx, overflow_flag)
    return cfg


def make_cell_weights(packed: Dict[str, object]) -> L.CellWeights:
    """packed: name -> tensor (or list of 3 tensors for the sa_* members), fp32 contiguous on the GPU."""
    w = L.CellWeights()
    for name in L.CellWeights._names[0]:
        ts = packed[name]
        for t in ts:
            _need(t, name, torch.float32)
        setattr(w, name, (C.c_void_p * 3)(*[t.data_ptr() for t in ts]))
    for name in L.CellWeights._names[1]:
        t = packed.get(name)
        if t is not None:
            _need(t, name, torch.float32)
        setattr(w, name, 0 if t is None else t.data_ptr())
    for name in ("sa_w2_x3", "sa_w1_x3"):
        x3 = packed.get(name)
        if x3 is not None:
            for t in x3:
                if t is not None:
                    _need(t, name, torch.int16)
            setattr(w, name, (C.c_void_p * 3)(*[0 if t is None else t.data_ptr() for t in x3]))
    if packed.get("sa_b2_x3") is not None:
        for t in packed["sa_b2_x3"]:
            _need(t, "sa_b2_x3", torch.float32)
        w.sa_b2_x3 = (C.c_void_p * 3)(*[t.data_ptr() for t in packed["sa_b2_x3"]])
        w.sa_w2_scale = (C.c_float * 3)(*[float(v) for v in packed["sa_w2_scale"]])
    for name in ("lin1", "lin2", "merge", "pn", "g_wp", "g_wq"):
        img = packed.get(name + "_x3")
        if img is not None:
            _need(img, name + "_x3", torch.int16)
            setattr(w, name + "_x3", img.data_ptr())
            setattr(w, name + "_scale", float(packed[name + "_scale"]))
    for name in ("ga_w1_x3", "ga_w2_x3", "g_w2_x3"):
        g = packed.get(name)
        if g is not None:
            _need(g, name, torch.int16)
            setattr(w, name, g.data_ptr())
    if packed.get("ga_w1_l1") is not None:
        w.ga_w1_l1, w.ga_b1_absmax = float(packed["ga_w1_l1"]), float(packed["ga_b1_absmax"])
        w.sa_wp_l1 = (C.c_float * 3)(*[float(v) for v in packed["sa_wp_l1"]])
        w.sa_a1_l1, w.sa_b1_absmax = float(packed["sa_a1_l1"]), float(packed["sa_b1_absmax"])
    for name in ("class_embedding", "color_embedding"):
        t = packed.get(name)
        if t is not None:
            _need(t, name, torch.float32, 2)
            setattr(w, name, t.data_ptr())
    return w


def encode_cells(xyz, rgb, center, mean_rgb, cell_ptr_host: np.ndarray, cell_ptr_dev, weights: L.CellWeights,
                 cfg: L.CellConfig, want_trace=False, ws_tag: str = "encode_cells"):
    """Cell branch on packed, device-resident inputs.  Returns out [n_cells, D] (and a dict of stage outputs when
    want_trace is True or names some of them; knn_idx rows are local to the internal chunk of <= chunk_objects objects)."""
    _need(xyz, "xyz", torch.float32, 3)
    dev = xyz.device
    _need(rgb, "rgb", torch.float32, 3, dev)
    _need(center, "center", torch.float32, 2, dev)
    _need(mean_rgb, "mean_rgb", torch.float32, 2, dev)
    _need(cell_ptr_dev, "cell_ptr", torch.int32, 1, dev)
    n_obj, n_pts = xyz.shape[0], xyz.shape[1]
    if tuple(rgb.shape) != tuple(xyz.shape) or xyz.shape[2] != 3:
        raise RuntimeError(f"encode_cells: xyz {tuple(xyz.shape)} / rgb {tuple(rgb.shape)} must both be [n_obj, n_pts, 3]")
    if tuple(center.shape) != (n_obj, 3) or tuple(mean_rgb.shape) != (n_obj, 3):
        raise RuntimeError("encode_cells: center / mean_rgb must be [n_obj, 3]")
    for name in ("class", "color"):
        if getattr(cfg, name + "_embed") and cfg._keepalive[0 if name == "class" else 1].numel() != n_obj:
            raise RuntimeError(f"encode_cells: {name}_idx must have one entry per object")
    if n_pts != cfg.n_pts:
        raise RuntimeError(f"encode_cells: objects have {n_pts} points, config says {cfg.n_pts}")
    cp = np.ascontiguousarray(cell_ptr_host, dtype=np.int32)
    n_cells = cp.shape[0] - 1
    if cell_ptr_dev.numel() != n_cells + 1:
        raise RuntimeError("encode_cells: host and device cell_ptr differ in length")
    D = cfg.embed_dim
    out = None if cfg.objects_only else torch.empty((n_cells, D), dtype=torch.float32, device=dev)
    trace, tr = None, None
    if cfg.objects_only:  # ObjectEncoder.forward only: the [n_obj, D] result comes back through the trace slot
        if want_trace:
            raise RuntimeError("encode_cells: objects_only returns the object embeddings only")
        tr = L.CellTrace()
        obj_emb = torch.empty((n_obj, D), dtype=torch.float32, device=dev)
        tr.obj_emb = obj_emb.data_ptr()
    elif want_trace:
        # want_trace: True = every stage output, or an iterable of names (the full set is ~130 KB per object)
        names = set(("fps_idx", "nbr", "cnt", "sa_out", "features0", "features1", "features2", "obj_emb", "knn_idx")
                    if want_trace is True else want_trace)
        unknown = names - {"fps_idx", "nbr", "cnt", "sa_out", "features0", "features1", "features2", "obj_emb", "knn_idx"}
        if unknown:
            raise RuntimeError(f"encode_cells: unknown trace outputs {sorted(unknown)}")
        tr = L.CellTrace()
        trace = {}
        shapes, nd = [], n_pts
        for c in (64, 128, 256):
            nc = (nd + 1) // 2
            shapes.append((nc, c))
            nd = nc
        per_level = {"fps_idx": lambda nc, c: ((n_obj, nc), torch.uint8), "nbr": lambda nc, c: ((n_obj, nc, 32), torch.uint8),
                     "cnt": lambda nc, c: ((n_obj, nc), torch.uint8),
                     "sa_out": lambda nc, c: ((n_obj * nc, c + 32), torch.float32)}
        for name, fn in per_level.items():
            if name in names:
                trace[name] = [torch.zeros(fn(nc, c)[0], dtype=fn(nc, c)[1], device=dev) for nc, c in shapes]
                setattr(tr, name, (C.c_void_p * 3)(*[t.data_ptr() for t in trace[name]]))
        flat = {"features0": ((n_obj, 1024), torch.float32), "features1": ((n_obj, 512), torch.float32),
                "features2": ((n_obj, 256), torch.float32), "obj_emb": ((n_obj, D), torch.float32),
                "knn_idx": ((n_obj, cfg.knn_k), torch.int32)}
        for name, (shape, dt) in flat.items():
            if name in names:
                trace[name] = torch.empty(shape, dtype=dt, device=dev)
                setattr(tr, name, trace[name].data_ptr())
    nbytes = L.lib().t2p_encode_cells_workspace_bytes(n_obj, n_cells, C.byref(cfg))
    # a single cell larger than the chunk forms its own (bigger) chunk: size for it
    biggest = int((cp[1:] - cp[:-1]).max()) if n_cells > 0 else 0
    chunk = cfg.chunk_objects if cfg.chunk_objects > 0 else DEFAULT_CHUNK_OBJECTS
    if biggest > chunk:
        big_cfg = L.CellConfig.from_buffer_copy(cfg)
        big_cfg.chunk_objects = biggest
        nbytes = max(nbytes, L.lib().t2p_encode_cells_workspace_bytes(n_obj, n_cells, C.byref(big_cfg)))
    ws = workspace(dev, nbytes, ws_tag)   # (one scratch buffer per tag: concurrent calls on two streams need two tags)
    rc = L.lib().t2p_encode_cells(_ptr(xyz), _ptr(rgb), _ptr(center), _ptr(mean_rgb),
                                  cp.ctypes.data_as(C.c_void_p), _ptr(cell_ptr_dev), n_obj, n_cells, C.byref(weights),
                                  C.byref(cfg), _ptr(out) if out is not None else None,
                                  C.byref(tr) if tr is not None else None, _ptr(ws),
                                  ws.numel(), _stream(dev))
    L.check(rc, "t2p_encode_cells")
    if cfg.objects_only:
        return obj_emb
    return (out, trace) if want_trace else out


def pack_objects(raw_xyz, raw_rgb, obj_ptr, sample_idx, rot=None):
    """Device-side FixedPoints gather (+ RandomRotate about z) + NormalizeScale + per-object means.  raw_xyz/raw_rgb
    [Np,3] fp32, obj_ptr [Nobj+1] int32, sample_idx [Nobj,P] int32 (local indices), rot None or [Nobj,2] fp32 (cos, sin
    of each object's angle, data.draw_rotations).  Returns (xyz [Nobj,P,3], rgb [Nobj,P,3], center, mean_rgb)."""
    _need(raw_xyz, "raw_xyz", torch.float32, 2)
    dev = raw_xyz.device
    _need(raw_rgb, "raw_rgb", torch.float32, 2, dev)
    _need(obj_ptr, "obj_ptr", torch.int32, 1, dev)
    _need(sample_idx, "sample_idx", torch.int32, 2, dev)
    n_obj, n_pts = sample_idx.shape
    if obj_ptr.numel() != n_obj + 1 or tuple(raw_rgb.shape) != tuple(raw_xyz.shape) or raw_xyz.shape[1] != 3:
        raise RuntimeError("pack_objects: inconsistent shapes")
    if rot is not None:
        _need(rot, "rot", torch.float32, 2, dev)
        if tuple(rot.shape) != (n_obj, 2):
            raise RuntimeError("pack_objects: rot must be [n_obj, 2]")
    xyz = torch.empty((n_obj, n_pts, 3), dtype=torch.float32, device=dev)
    rgb = torch.empty((n_obj, n_pts, 3), dtype=torch.float32, device=dev)
    center = torch.empty((n_obj, 3), dtype=torch.float32, device=dev)
    mean_rgb = torch.empty((n_obj, 3), dtype=torch.float32, device=dev)
    L.check(L.lib().t2p_pack_objects(_ptr(raw_xyz), _ptr(raw_rgb), _ptr(obj_ptr), _ptr(sample_idx),
                                     _ptr(rot) if rot is not None else None, n_obj, n_pts,
                                     _ptr(xyz), _ptr(rgb), _ptr(center), _ptr(mean_rgb), _stream(dev)), "t2p_pack_objects")
    return xyz, rgb, center, mean_rgb


def pack_scene_objects(raw_xyz, raw_rgb, obj_ptr, obj_id, key, scene_center, scene_color, n_pts: int = 256, want_rgb: bool = True,
                       want_idx: bool = False):
    """The dataloader of a scene resident in HBM (t2p_pack_scene_objects, include/t2p.h): output slot s = scene object
    obj_id[s] (int32 [n_out]) resampled to n_pts points by the counter-based draw of key[s] (int64 / uint64 bit pattern
    [n_out]), NormalizeScale'd bit for bit like the host chain; centre / mean colour gathered from the scene's tables.
    Returns (xyz, rgb | None, center, mean_rgb[, sample_idx])."""
    _need(raw_xyz, "raw_xyz", torch.float32, 2)
    dev = raw_xyz.device
    _need(raw_rgb, "raw_rgb", torch.float32, 2, dev)
    _need(obj_ptr, "obj_ptr", torch.int32, 1, dev)
    _need(obj_id, "obj_id", torch.int32, 1, dev)
    _need(key, "key", torch.int64, 1, dev)
    _need(scene_center, "scene_center", torch.float32, 2, dev)
    _need(scene_color, "scene_color", torch.float32, 2, dev)
    n_out = obj_id.shape[0]
    if key.shape[0] != n_out or tuple(raw_rgb.shape) != tuple(raw_xyz.shape) or raw_xyz.shape[1] != 3 or \
            scene_center.shape[0] != obj_ptr.shape[0] - 1 or tuple(scene_color.shape) != tuple(scene_center.shape):
        raise RuntimeError("pack_scene_objects: inconsistent shapes")
    xyz = torch.empty((n_out, n_pts, 3), dtype=torch.float32, device=dev)
    rgb = torch.empty((n_out, n_pts, 3), dtype=torch.float32, device=dev) if want_rgb else None
    center = torch.empty((n_out, 3), dtype=torch.float32, device=dev)
    mean_rgb = torch.empty((n_out, 3), dtype=torch.float32, device=dev)
    idx = torch.empty((n_out, n_pts), dtype=torch.int32, device=dev) if want_idx else None
    L.check(L.lib().t2p_pack_scene_objects(_ptr(raw_xyz), _ptr(raw_rgb), _ptr(obj_ptr), _ptr(obj_id), _ptr(key), _ptr(scene_center),
                                           _ptr(scene_color), n_out, n_pts, _ptr(xyz), _ptr(rgb) if want_rgb else None,
                                           _ptr(center), _ptr(mean_rgb), _ptr(idx) if want_idx else None, _stream(dev)),
            "t2p_pack_scene_objects")
    return (xyz, rgb, center, mean_rgb, idx) if want_idx else (xyz, rgb, center, mean_rgb)


def make_match_weights(packed: Dict[str, object]) -> L.MatchWeights:
    """packed: packing.pack_match_weights(...)"""
    w = L.MatchWeights()
    cross = np.ascontiguousarray(packed["cross"], dtype=np.int32)
    w.n_layers = int(cross.shape[0])
    w.cross = cross.ctypes.data_as(C.c_void_p)
    for name in ("wqkv", "bqkv", "wm", "bm", "w1", "b1", "w2", "b2", "wf", "bf", "wo1", "bo1", "wo2", "bo2"):
        t = packed[name]
        _need(t, name, torch.float32)
        setattr(w, name, t.data_ptr())
    w.bin_score = float(packed["bin_score"])
    if packed.get("wqkv_x3") is not None:   # f16x3 images of the GNN matrices
        for name, sc in (("wqkv", "qkv"), ("wm", "m"), ("w1", "1"), ("w2", "2"), ("wf", "f")):
            _need(packed[name + "_x3"], name + "_x3", torch.int16)
            setattr(w, name + "_x3", packed[name + "_x3"].data_ptr())
            setattr(w, "scale_" + sc, float(packed["scale_" + sc]))
    w._keepalive = (cross, packed)
    return w


def match(desc0, desc1, weights: L.MatchWeights, sinkhorn_iters: int, threshold: float = 0.2):
    """desc0 [B, M, D] object encodings, desc1 [B, N, D] hint encodings (fp32, L2-normalised, on the GPU).
    Returns dict(P [B, M+1, N+1], matches0 [B, M] int64, matches1 [B, N] int64, matching_scores0/1, offsets [B, N, 2])."""
    _need(desc0, "desc0", torch.float32, 3)
    dev = desc0.device
    _need(desc1, "desc1", torch.float32, 3, dev)
    b, m, d = desc0.shape
    n = desc1.shape[1]
    if desc1.shape[0] != b or desc1.shape[2] != d:
        raise RuntimeError(f"match: desc0 {tuple(desc0.shape)} / desc1 {tuple(desc1.shape)} disagree")
    out = dict(P=torch.empty((b, m + 1, n + 1), dtype=torch.float32, device=dev),
               matches0=torch.empty((b, m), dtype=torch.int64, device=dev),
               matches1=torch.empty((b, n), dtype=torch.int64, device=dev),
               matching_scores0=torch.empty((b, m), dtype=torch.float32, device=dev),
               matching_scores1=torch.empty((b, n), dtype=torch.float32, device=dev),
               offsets=torch.empty((b, n, 2), dtype=torch.float32, device=dev))
    ws = workspace(dev, L.lib().t2p_match_workspace_bytes(b, m, n, d), "match")
    rc = L.lib().t2p_match(_ptr(desc0), _ptr(desc1), b, m, n, d, C.byref(weights), int(sinkhorn_iters), float(threshold),
                           _ptr(out["P"]), _ptr(out["matches0"]), _ptr(out["matches1"]), _ptr(out["matching_scores0"]),
                           _ptr(out["matching_scores1"]), _ptr(out["offsets"]), _ptr(ws), ws.numel(), _stream(dev))
    L.check(rc, "t2p_match")
    return out


def make_text_weights(embedding, w_ih, w_hh, bias, w_hh_x3=None, w_hh_scale=0.0) -> L.TextWeights:
    """w_hh_x3 (int16 device tensor, packing.pack_text_weights(x3=True)) selects the f16x3 recurrence; None = exact fp32."""
    w = L.TextWeights()
    for n, t in (("embedding", embedding), ("w_ih", w_ih), ("w_hh", w_hh), ("bias", bias)):
        _need(t, n, torch.float32)
        setattr(w, n, t.data_ptr())
    if w_hh_x3 is not None:
        _need(w_hh_x3, "w_hh_x3", torch.int16)
        w.w_hh_x3, w.w_hh_scale = w_hh_x3.data_ptr(), float(w_hh_scale)
    return w


def encode_text(tokens: torch.Tensor, lengths: torch.Tensor, weights: L.TextWeights, vocab: int, embed_dim: int,
                want_raw: bool = False):
    """tokens [B, T] int32 right-padded with 0, lengths [B] int32 -> L2-normalised [B, D] (and the raw encoder output)."""
    _need(tokens, "tokens", torch.int32, 2)
    dev = tokens.device
    _need(lengths, "lengths", torch.int32, 1, dev)
    b, t = tokens.shape
    if lengths.numel() != b:
        raise RuntimeError("encode_text: lengths must have one entry per row of tokens")
    out = torch.empty((b, embed_dim), dtype=torch.float32, device=dev)
    raw = torch.empty((b, embed_dim), dtype=torch.float32, device=dev) if want_raw else None
    nbytes = L.lib().t2p_encode_text_workspace_bytes(b, vocab, embed_dim)
    ws = workspace(dev, nbytes, "encode_text")
    L.check(L.lib().t2p_encode_text(_ptr(tokens), _ptr(lengths), b, t, vocab, embed_dim, C.byref(weights), _ptr(raw),
                                    _ptr(out), _ptr(ws), ws.numel(), _stream(dev)), "t2p_encode_text")
    return (out, raw) if want_raw else out


def pairwise_ranking(scores: torch.Tensor, margin: float):
    """scores [B, B] fp32 -> (row_loss [B], d_scores [B, B]) of PairwiseRankingLoss (t2p_pairwise_ranking)."""
    _need(scores, "scores", torch.float32, 2)
    b = scores.shape[0]
    if scores.shape[1] != b:
        raise RuntimeError("pairwise_ranking: scores must be square")
    dev = scores.device
    row_loss = torch.empty((b,), dtype=torch.float32, device=dev)
    row_cnt = torch.empty((b,), dtype=torch.float32, device=dev)
    d_scores = torch.empty_like(scores)
    L.check(L.lib().t2p_pairwise_ranking(_ptr(scores), b, float(margin), _ptr(row_loss), _ptr(d_scores), _ptr(row_cnt),
                                         _stream(dev)), "t2p_pairwise_ranking")
    return row_loss, d_scores


def hardest_ranking(scores: torch.Tensor, margin: float):
    """scores [B, B] fp32 -> (best [2B], d_scores [B, B]) of HardestRankingLoss (t2p_hardest_ranking)."""
    _need(scores, "scores", torch.float32, 2)
    b = scores.shape[0]
    if scores.shape[1] != b:
        raise RuntimeError("hardest_ranking: scores must be square")
    dev = scores.device
    best = torch.empty((2 * b,), dtype=torch.float32, device=dev)
    where = torch.empty((2 * b,), dtype=torch.int32, device=dev)
    d_scores = torch.empty_like(scores)
    L.check(L.lib().t2p_hardest_ranking(_ptr(scores), b, float(margin), _ptr(best), _ptr(where), _ptr(d_scores), _stream(dev)),
            "t2p_hardest_ranking")
    return best, d_scores


def lstm_cell_forward(pre, table, tokens, lengths, step: int, reverse: bool, c_prev, h_prev, gates, c, h):
    """One training-mode LSTM step (t2p_lstm_cell_forward): pre [B,4D] = h_prev @ W_hh^T, table [V,4D]; writes gates
    [B,4D] (i, f, g, o), c, h [B,D] in place."""
    dev = pre.device
    for name, t in (("pre", pre), ("table", table), ("gates", gates)):
        _need(t, name, torch.float32, 2, dev)
    for name, t in (("c_prev", c_prev), ("h_prev", h_prev), ("c", c), ("h", h)):
        _need(t, name, torch.float32, 2, dev)
    _need(tokens, "tokens", torch.int32, 2, dev)
    _need(lengths, "lengths", torch.int32, 1, dev)
    b, d = c.shape
    if tuple(pre.shape) != (b, 4 * d) or tuple(gates.shape) != (b, 4 * d) or table.shape[1] != 4 * d or tokens.shape[0] != b:
        raise RuntimeError("lstm_cell_forward: inconsistent shapes")
    L.check(L.lib().t2p_lstm_cell_forward(_ptr(pre), _ptr(table), _ptr(tokens), _ptr(lengths), b, tokens.shape[1], d,
                                          int(step), int(bool(reverse)), _ptr(c_prev), _ptr(h_prev), _ptr(gates), _ptr(c),
                                          _ptr(h), _stream(dev)), "t2p_lstm_cell_forward")


def lstm_train_loops_supported(d: int) -> bool:
    """Widths the in-library time loops take (16-byte loads of the state rows)."""
    return d % 4 == 0


def lstm_train_forward(table, w_hh_k, tokens, lengths, reverse: bool, gates, cs, hs):
    """All T steps of one direction in ONE call (t2p_lstm_train_forward): table [V,4D], w_hh_k [D,4D]; fills gates [T,B,4D],
    cs / hs [T+1,B,D] (slice 0 = the initial state, left as given)."""
    dev = table.device
    _need(table, "table", torch.float32, 2, dev)
    _need(w_hh_k, "w_hh_k", torch.float32, 2, dev)
    _need(tokens, "tokens", torch.int32, 2, dev)
    _need(lengths, "lengths", torch.int32, 1, dev)
    for name, t in (("gates", gates), ("cs", cs), ("hs", hs)):
        _need(t, name, torch.float32, 3, dev)
    b, t_len = tokens.shape
    d = w_hh_k.shape[0]
    if (tuple(w_hh_k.shape) != (d, 4 * d) or table.shape[1] != 4 * d or tuple(gates.shape) != (t_len, b, 4 * d)
            or tuple(cs.shape) != (t_len + 1, b, d) or tuple(hs.shape) != (t_len + 1, b, d) or lengths.shape[0] != b):
        raise RuntimeError("lstm_train_forward: inconsistent shapes")
    pre = torch.empty((b, 4 * d), dtype=torch.float32, device=dev)
    L.check(L.lib().t2p_lstm_train_forward(_ptr(table), _ptr(w_hh_k), _ptr(tokens), _ptr(lengths), b, t_len, d,
                                           int(bool(reverse)), _ptr(gates), _ptr(cs), _ptr(hs), _ptr(pre), _stream(dev)),
            "t2p_lstm_train_forward")


def lstm_train_backward(dh_last, w_hh_t, gates, cs, lengths, d_pre):
    """The whole backward recurrence of one direction in ONE call (t2p_lstm_train_backward): dh_last [B,D] = gradient of the
    final hidden state, w_hh_t [4D,D]; fills d_pre [T,B,4D]."""
    dev = gates.device
    _need(dh_last, "dh_last", torch.float32, 2, dev)
    _need(w_hh_t, "w_hh_t", torch.float32, 2, dev)
    _need(lengths, "lengths", torch.int32, 1, dev)
    for name, t in (("gates", gates), ("cs", cs), ("d_pre", d_pre)):
        _need(t, name, torch.float32, 3, dev)
    t_len, b, d4 = gates.shape
    d = d4 // 4
    if (tuple(dh_last.shape) != (b, d) or tuple(w_hh_t.shape) != (4 * d, d) or tuple(cs.shape) != (t_len + 1, b, d)
            or tuple(d_pre.shape) != (t_len, b, 4 * d) or lengths.shape[0] != b):
        raise RuntimeError("lstm_train_backward: inconsistent shapes")
    ws = torch.empty((5, b, d), dtype=torch.float32, device=dev)
    L.check(L.lib().t2p_lstm_train_backward(_ptr(dh_last), _ptr(w_hh_t), _ptr(gates), _ptr(cs), _ptr(lengths), b, t_len, d,
                                            _ptr(d_pre), _ptr(ws), _stream(dev)), "t2p_lstm_train_backward")


def lstm_cell_backward(dh_gemm, dh_carry_in, dc_in, gates, c_prev, c, lengths, step: int, d_pre, dc_out, dh_carry_out):
    """Backward of one training-mode LSTM step (t2p_lstm_cell_backward); dh_gemm may be None (last step)."""
    dev = gates.device
    for name, t in (("dh_carry_in", dh_carry_in), ("dc_in", dc_in), ("gates", gates), ("c_prev", c_prev), ("c", c),
                    ("d_pre", d_pre), ("dc_out", dc_out), ("dh_carry_out", dh_carry_out)):
        _need(t, name, torch.float32, 2, dev)
    if dh_gemm is not None:
        _need(dh_gemm, "dh_gemm", torch.float32, 2, dev)
    _need(lengths, "lengths", torch.int32, 1, dev)
    b, d = c.shape
    if tuple(gates.shape) != (b, 4 * d) or tuple(d_pre.shape) != (b, 4 * d):
        raise RuntimeError("lstm_cell_backward: inconsistent shapes")
    L.check(L.lib().t2p_lstm_cell_backward(_ptr(dh_gemm) if dh_gemm is not None else None, _ptr(dh_carry_in), _ptr(dc_in),
                                           _ptr(gates), _ptr(c_prev), _ptr(c), _ptr(lengths), b, d, int(step), _ptr(d_pre),
                                           _ptr(dc_out), _ptr(dh_carry_out), _stream(dev)), "t2p_lstm_cell_backward")


# ---------------------------------------------------------------------------------------------------------------
def profile_enable(on: bool):
    """Bracket every kernel launch with hipEvents on its launch stream (bench.py's live per-kernel timing)."""
    L.lib().t2p_profile_enable(int(bool(on)))


def profile_repeat(scope: Optional[str], reps: int = 1):
    """Measurement hook (t2p_profile_repeat): launches reported under `scope` are issued `reps` times back to back (same
    arguments, same results) - profiles/energy_table.py holds one kernel on the chip for seconds that way.  None / 1: off."""
    L.lib().t2p_profile_repeat(scope.encode() if scope else None, int(reps))


def profile_report() -> Dict[str, tuple]:
    """{kernel name: (launches, total_ms)} of the launches recorded since the last report; waits for them."""
    buf = C.create_string_buffer(1 << 16)
    L.check(L.lib().t2p_profile_report(buf, len(buf)), "t2p_profile_report")
    out = {}
    for line in buf.value.decode().splitlines():
        name, cnt, ms = line.split()
        out[name] = (int(cnt), float(ms))
    return out

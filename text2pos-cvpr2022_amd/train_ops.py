"""Autograd building blocks of the training-mode cell branch (second part), backed by
csrc/train_ops.hip through the C ABI: batch-statistics BatchNorm1d (+ReLU) over row segments, segment max with the winning
row remembered, and Linear on the tiled GEMM.  The reference layers they stand for: `get_mlp` blocks in `model.train()`
(models/modules.py:21-29; PointNet++ layers take their statistics per cell because the reference runs it once per cell,
models/object_encoder.py:92-95), `PointConv(aggr="max")`, `gnn.global_max_pool`, `DynamicEdgeConv(aggr="max")`
(models/pointcloud/pointnet2.py:31-49, models/cell_retrieval.py:46-49, :96-99)."""
import ctypes as C

import torch

from . import _lib as L
from . import ops
from .ops import _need, _ptr, _stream


class _BnReluTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, seg_ptr, gamma, beta, eps, relu):
        _need(x, "x", torch.float32, 2)
        dev = x.device
        _need(seg_ptr, "seg_ptr", torch.int32, 1, dev)
        _need(gamma, "gamma", torch.float32, 1, dev)
        _need(beta, "beta", torch.float32, 1, dev)
        m, c = x.shape
        n_seg = seg_ptr.numel() - 1
        y = torch.empty_like(x)
        mean, invstd, var_u = (torch.empty((n_seg, c), dtype=torch.float32, device=dev) for _ in range(3))
        ws = torch.empty((max(1, L.lib().t2p_bn_train_workspace_bytes(m, n_seg, c)),), dtype=torch.uint8, device=dev)
        L.check(L.lib().t2p_bn_relu_train_forward(_ptr(x), _ptr(seg_ptr), n_seg, m, c, _ptr(gamma.detach()),
                                                  _ptr(beta.detach()), float(eps), int(bool(relu)), _ptr(y), _ptr(mean),
                                                  _ptr(invstd), _ptr(var_u), _ptr(ws), ws.numel(), _stream(dev)),
                "t2p_bn_relu_train_forward")
        ctx.save_for_backward(x, seg_ptr, mean, invstd, gamma.detach(), beta.detach())   # (not y: its sign is recomputed from x)
        ctx.relu = bool(relu)
        ctx.mark_non_differentiable(mean, var_u)
        return y, mean, var_u

    @staticmethod
    def backward(ctx, dy, _dmean, _dvar):
        x, seg_ptr, mean, invstd, gamma, beta = ctx.saved_tensors
        dev = x.device
        n_seg, c = mean.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dg, db = (torch.empty((n_seg + 1, c), dtype=torch.float32, device=dev) for _ in range(2))   # row n_seg: the sum over the segments
        m = x.shape[0]
        if n_seg == 0:
            dg.zero_()
            db.zero_()
        ws = torch.empty((max(1, L.lib().t2p_bn_train_workspace_bytes(m, n_seg, c)),), dtype=torch.uint8, device=dev)
        L.check(L.lib().t2p_bn_relu_train_backward(_ptr(dy), _ptr(x), _ptr(beta.contiguous()), _ptr(seg_ptr), n_seg, m, c, _ptr(mean),
                                                   _ptr(invstd), _ptr(gamma), int(ctx.relu), _ptr(dx), _ptr(dg), _ptr(db),
                                                   _ptr(ws), ws.numel(), _stream(dev)), "t2p_bn_relu_train_backward")
        return dx, None, dg[n_seg], db[n_seg], None, None


def bn_relu_train(x, seg_ptr, bn: torch.nn.BatchNorm1d, relu: bool = True, rows_min: int = None):
    """BatchNorm1d in training mode (+ ReLU) over the row segments seg_ptr [S+1] int32 (device), statistics per segment.
    Updates bn.running_mean / running_var / num_batches_tracked once per segment, in order, exactly as S consecutive
    calls of the module would (momentum bn.momentum - None = cumulative average, as in torch -, unbiased variance).
    rows_min: the smallest segment's row count when the caller knows it on the host (otherwise it is read back from
    seg_ptr); like nn.BatchNorm1d, a segment with a single row is refused in training mode."""
    if rows_min is None:
        rows_min = int((seg_ptr[1:] - seg_ptr[:-1]).min().item()) if seg_ptr.numel() > 1 else 2
    if rows_min <= 1:
        raise ValueError(f"Expected more than 1 value per channel when training, got a segment of {rows_min} row(s) "
                         f"x {x.shape[1]} channels")     # torch.nn.functional._verify_batch_size
    y, mean, var_u = _BnReluTrainFn.apply(x.contiguous(), seg_ptr, bn.weight, bn.bias, bn.eps, relu)
    if bn.track_running_stats and bn.running_mean is not None:
        with torch.no_grad():
            s = mean.shape[0]
            if bn.momentum is None:
                # cumulative moving average: the k-th call uses factor 1 / (n0 + k); S calls in closed form
                n0 = float(bn.num_batches_tracked.item())
                bn.running_mean.mul_(n0 / (n0 + s)).add_(mean.sum(0) / (n0 + s))
                bn.running_var.mul_(n0 / (n0 + s)).add_(var_u.sum(0) / (n0 + s))
            else:
                mom = float(bn.momentum)
                # r <- (1 - m) r + m stat, S times: closed form with weights m (1 - m)^(S-1-k)
                w = mom * (1.0 - mom) ** torch.arange(s - 1, -1, -1, device=mean.device, dtype=torch.float32)
                keep = (1.0 - mom) ** s
                bn.running_mean.mul_(keep).add_((w[:, None] * mean).sum(0))
                bn.running_var.mul_(keep).add_((w[:, None] * var_u).sum(0))
            bn.num_batches_tracked += s
    return y


class _SegmentMaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, seg_ptr, covers_all_rows=False):
        _need(x, "x", torch.float32, 2)
        dev = x.device
        _need(seg_ptr, "seg_ptr", torch.int32, 1, dev)
        n_seg, c = seg_ptr.numel() - 1, x.shape[1]
        out = torch.empty((n_seg, c), dtype=torch.float32, device=dev)
        arg = torch.empty((n_seg, c), dtype=torch.int32, device=dev)
        L.check(L.lib().t2p_segment_max_forward(_ptr(x), _ptr(seg_ptr), n_seg, c, _ptr(out), _ptr(arg), _stream(dev)),
                "t2p_segment_max_forward")
        ctx.save_for_backward(arg, seg_ptr)
        ctx.rows = x.shape[0]
        ctx.tiles = bool(covers_all_rows)
        return out

    @staticmethod
    def backward(ctx, dout):
        arg, seg_ptr = ctx.saved_tensors
        n_seg, c = arg.shape
        # The kernel writes every element of every segment's rows (the winner's gradient or 0).  When the caller KNOWS that the
        # segments tile [0, rows) (the training path's seg_ptr comes from its host plan) the zero fill is skipped; otherwise rows
        # outside every segment must not leak uninitialised memory into the gradients.
        alloc = torch.empty if ctx.tiles else torch.zeros
        dx = alloc((ctx.rows, c), dtype=torch.float32, device=dout.device)
        L.check(L.lib().t2p_segment_max_backward(_ptr(dout.contiguous()), _ptr(arg), _ptr(seg_ptr), n_seg, c, _ptr(dx),
                                                 _stream(dout.device)), "t2p_segment_max_backward")
        return dx, None, None


def segment_max(x, seg_ptr, covers_all_rows: bool = False):
    """Row-segment maximum [S, C] of x [M, C] (rows sorted by destination); gradient flows to the winning rows.
    covers_all_rows: the caller guarantees seg_ptr[0] == 0 and seg_ptr[-1] == M (the backward then skips zero-filling dx)."""
    return _SegmentMaxFn.apply(x.contiguous(), seg_ptr, covers_all_rows)


class _SegmentMeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, seg_ptr):
        n_seg, c = seg_ptr.numel() - 1, x.shape[1]
        out = torch.empty((n_seg, c), dtype=torch.float32, device=x.device)
        L.check(L.lib().t2p_segment_mean_forward(_ptr(x), _ptr(seg_ptr), n_seg, c, _ptr(out), _stream(x.device)),
                "t2p_segment_mean_forward")
        ctx.save_for_backward(seg_ptr)
        ctx.rows = x.shape[0]
        return out

    @staticmethod
    def backward(ctx, dout):
        (seg_ptr,) = ctx.saved_tensors
        n_seg, c = dout.shape
        dx = torch.zeros((ctx.rows, c), dtype=torch.float32, device=dout.device)
        L.check(L.lib().t2p_segment_mean_backward(_ptr(dout.contiguous()), _ptr(seg_ptr), n_seg, c, _ptr(dx),
                                                  _stream(dout.device)), "t2p_segment_mean_backward")
        return dx, None


def segment_mean(x, seg_ptr):
    """Row-segment mean [S, C] of x [M, C] (variation 1: aggr="mean" / global_mean_pool)."""
    return _SegmentMeanFn.apply(x.contiguous(), seg_ptr)


class _EdgeFeaturesFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pos, pos_c, src, dst):
        dev = pos.device
        e, c = src.numel(), x.shape[1]
        w = (c + 3 + 7) // 8 * 8                           # row pitch: the GEMM's granule, zero columns behind the message
        out = torch.empty((e, w), dtype=torch.float32, device=dev)
        L.check(L.lib().t2p_edge_features_forward(_ptr(x), _ptr(pos), _ptr(pos_c), _ptr(src), _ptr(dst), e, c, w, _ptr(out),
                                                  _stream(dev)), "t2p_edge_features_forward")
        ctx.save_for_backward(src)
        ctx.shape = tuple(x.shape)
        ctx.width = w
        return out

    @staticmethod
    def backward(ctx, dout):
        (src,) = ctx.saved_tensors
        dx = torch.zeros(ctx.shape, dtype=torch.float32, device=dout.device)
        L.check(L.lib().t2p_edge_features_backward(_ptr(dout.contiguous()), _ptr(src), src.numel(), ctx.shape[1], ctx.width, _ptr(dx),
                                                   _stream(dout.device)), "t2p_edge_features_backward")
        return dx, None, None, None, None


def edge_features(x, pos, pos_c, src, dst):
    """[x[src] | pos[src] - pos_c[dst] | 0 ...] per edge, C + 3 columns zero-padded to a multiple of 8 (linear() pairs the pad
    columns with zero weight columns); src / dst int32 (device); gradient to x only (positions are inputs)."""
    return _EdgeFeaturesFn.apply(x.contiguous(), pos.contiguous(), pos_c.contiguous(), src, dst)


class _PairFeaturesFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, tgt, src):
        e, d = tgt.numel(), x.shape[1]
        out = torch.empty((e, 2 * d), dtype=torch.float32, device=x.device)
        L.check(L.lib().t2p_pair_features_forward(_ptr(x), _ptr(tgt), _ptr(src), e, d, _ptr(out), _stream(x.device)),
                "t2p_pair_features_forward")
        ctx.save_for_backward(tgt, src)
        ctx.shape = tuple(x.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        tgt, src = ctx.saved_tensors
        dx = torch.zeros(ctx.shape, dtype=torch.float32, device=dout.device)
        L.check(L.lib().t2p_pair_features_backward(_ptr(dout.contiguous()), _ptr(tgt), _ptr(src), tgt.numel(), ctx.shape[1],
                                                   _ptr(dx), _stream(dout.device)), "t2p_pair_features_backward")
        return dx, None, None


def pair_features(x, tgt, src):
    """[x[tgt] | x[src] - x[tgt]] per edge (DynamicEdgeConv message input); tgt / src int32 (device)."""
    return _PairFeaturesFn.apply(x.contiguous(), tgt, src)


class _NormalizeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.rownorm(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        L.check(L.lib().t2p_rownorm_backward(_ptr(x), _ptr(dy.contiguous()), x.shape[0], x.shape[1], _ptr(dx), _stream(x.device)),
                "t2p_rownorm_backward")
        return dx


def normalize(x):
    """F.normalize(x, dim=-1) on t2p_rownorm with its backward kernel."""
    return _NormalizeFn.apply(x.contiguous())


class _LinearFn(torch.autograd.Function):
    """x [M, K] @ weight[N, K]^T + bias on the tiled fp32-MFMA GEMM (t2p_gemm); dX on the same GEMM (weight is its k-major
    operand); dW = dY^T X and db = column sums of dY in ONE pass over the rows on t2p_linear_wgrad_f32 (csrc/train_gemm.hip:
    every row of dY and X read once per row range, fixed-order reduction of the partial blocks: deterministic)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        k, n = x.shape[1], weight.shape[0]
        kw = weight.shape[1]                               # k > kw: x arrives with zero columns behind its kw inputs (edge_features)
        pad = (-k) % 8                                     # the GEMM wants K % 4 == 0 and N % 8 == 0: zero columns
        xp = torch.nn.functional.pad(x.detach(), (0, pad)).contiguous() if pad else x.detach().contiguous()
        wp = (torch.nn.functional.pad(weight.detach(), (0, k + pad - kw)).contiguous() if k + pad != kw
              else weight.detach().contiguous())
        ctx.save_for_backward(xp, wp)
        ctx.k = k
        ctx.kw = kw
        ctx.has_bias = bias is not None
        b = bias.detach().contiguous() if bias is not None else None
        pad_n = (-n) % 8                                   # e.g. --embed_dim 300 (training/args.py:19): zero output columns, cut off
        if pad_n:
            wt = torch.nn.functional.pad(wp, (0, 0, 0, pad_n)).t().contiguous()
            bp = torch.nn.functional.pad(b, (0, pad_n)).contiguous() if b is not None else None
            return ops.gemm(xp, wt, bp)[:, :n].contiguous()
        return ops.gemm(xp, wp.t().contiguous(), b)

    @staticmethod
    def backward(ctx, dy):
        xp, wp = ctx.saved_tensors
        dy = dy.contiguous()
        need_x, need_w, need_b = ctx.needs_input_grad
        # [M, N] x [N, K]: the weight is its own k-major operand
        dx = ops.gemm(dy, wp)[:, : ctx.k] if need_x else None
        dw = db = None
        want_b = bool(need_b and ctx.has_bias)
        if need_w:                                         # frozen layers (--pointnet_freeze) skip it
            dw, db = ops.linear_wgrad(dy, xp, want_colsum=want_b)
            dw = dw[:, : ctx.kw]
        elif want_b:
            db = dy.sum(0)
        return dx, dw, db


def linear(x, lin: torch.nn.Linear):
    return _LinearFn.apply(x, lin.weight, lin.bias)

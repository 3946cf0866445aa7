"""Host-side weight preparation for the HIP kernels: eval-mode BatchNorm folding, transposition to k-major
([in][out]) and zero-padding of the K dimension, all in float64 before the single rounding to fp32.

Parameter containers keep the reference's state_dict layout (models/modules.py:21-29: each MLP layer is
Sequential(Linear, BatchNorm1d, ReLU) -> keys `<mlp>.<layer>.0.{weight,bias}` and
`<mlp>.<layer>.1.{weight,bias,running_mean,running_var,num_batches_tracked}`).
"""
from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.nn as nn


def fold_linear_bn(layer: nn.Sequential) -> Tuple[torch.Tensor, torch.Tensor]:
    """(W [out,in], b [out]) in float64 of Linear followed by eval-mode BatchNorm1d (if present)."""
    lin = layer[0]
    w = lin.weight.detach().double()
    b = lin.bias.detach().double() if lin.bias is not None else torch.zeros(w.shape[0], dtype=torch.float64,
                                                                               device=w.device)
    if len(layer) > 1 and isinstance(layer[1], nn.BatchNorm1d):
        bn = layer[1]
        s = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
        w = w * s[:, None]
        b = (b - bn.running_mean.detach().double()) * s + bn.bias.detach().double()
    return w, b


def kmajor(w: torch.Tensor, k_pad: int = None) -> torch.Tensor:
    """[out,in] float64 -> contiguous fp32 [in (zero-padded to k_pad)][out]."""
    wt = w.t().contiguous()
    if k_pad is not None and k_pad > wt.shape[0]:
        wt = torch.cat([wt, torch.zeros(k_pad - wt.shape[0], wt.shape[1], dtype=wt.dtype, device=wt.device)], 0)
    return wt.float().contiguous()


def f32(b: torch.Tensor) -> torch.Tensor:
    return b.float().contiguous()


class Fp16RangeError(ValueError):
    """A folded weight cannot be represented by the f16x3 split (|w| past fp16's range): use precision="fp32"."""


def _check_fp16_range(w: torch.Tensor, what: str):
    m = float(w.detach().abs().max()) if w.numel() else 0.0
    if not np.isfinite(m) or m > 65504.0:
        raise Fp16RangeError(f"{what}: max |w| = {m:g} does not fit the f16x3 weight image (fp16 range 65504); "
                             "construct the model with precision=\"fp32\"")


def pack_f16x3(w_kmajor: torch.Tensor) -> torch.Tensor:
    """fp32 [K][N] k-major weights -> int16 [2][N/32][K/16][2][32][8]: the f16x3 split w = hi + lo/2048 (hi, lo fp16,
    round-to-nearest) laid out exactly as the MFMA B-operand registers of csrc/ws_sa.hip / ws_gemm.hip read it:
    element (plane, tile, step, half, lane, e) = split(w[k = half*K/2 + 8*step + e][n = 32*tile + lane])."""
    k, n = w_kmajor.shape
    assert k % 16 == 0 and n % 32 == 0, (k, n)
    w = w_kmajor.detach().float().cpu()
    _check_fp16_range(w, "pack_f16x3")
    hi = w.to(torch.float16)
    lo = ((w - hi.float()) * 2048.0).to(torch.float16)
    planes = []
    for t in (hi, lo):
        # [K][N] -> [half][step][e][tile][lane] -> [tile][step][half][lane][e]
        t = t.view(2, k // 16, 8, n // 32, 32).permute(3, 1, 0, 4, 2).contiguous()
        planes.append(t)
    return torch.stack(planes).view(torch.int16).contiguous()


def f16x3_scale(w: torch.Tensor) -> float:
    """Largest power of two s with max|s w| <= 2^14 (fp16 overflows at 65504), clipped to [2^-8, 2^15]."""
    m = float(w.detach().abs().max())
    if m == 0.0:
        return 1.0
    e = int(np.floor(np.log2(2.0 ** 14 / m)))
    return float(2.0 ** max(-8, min(15, e)))


def pack_f16x3_scaled(w_kmajor: torch.Tensor, scale: float) -> torch.Tensor:
    """Single-accumulator form used by the SA edge kernel (csrc/sa3.hip, sa_rows.hip, sa_points.hip): w' = scale * w (a power of two: exact),
    hi = fp16(w'), lo = fp16(w' - hi) WITHOUT the 2048 factor (the matrix cores honour fp16 denormals, and w' uses fp16's
    range, so lo keeps 11 significant bits).  Same register-order layout as pack_f16x3."""
    k, n = w_kmajor.shape
    assert k % 32 == 0 and n % 32 == 0, (k, n)
    w = w_kmajor.detach().double().cpu() * scale
    _check_fp16_range(w, "pack_f16x3_scaled")
    hi = w.to(torch.float16)
    lo = (w - hi.double()).to(torch.float16)
    planes = [t.view(2, k // 16, 8, n // 32, 32).permute(3, 1, 0, 4, 2).contiguous() for t in (hi, lo)]
    return torch.stack(planes).view(torch.int16).contiguous()


def pack_gemm_x3(w_kmajor: torch.Tensor, scale: float) -> torch.Tensor:
    """fp32 [K][N] k-major -> int16 [2][N][Kp] (Kp = K rounded up to 32, zero padded): the scaled split w' = scale w,
    hi = fp16(w'), lo = fp16(w' - hi), n-major with k contiguous -- a 16-byte load is one MFMA B-operand fragment of
    csrc/tg_gemm_x3.hip."""
    k, n = w_kmajor.shape
    kp = (k + 31) // 32 * 32
    w = torch.zeros((n, kp), dtype=torch.float64)
    w[:, :k] = w_kmajor.detach().double().cpu().t() * scale
    _check_fp16_range(w, "pack_gemm_x3")
    hi = w.to(torch.float16)
    lo = (w - hi.double()).to(torch.float16)
    return torch.stack([hi, lo]).view(torch.int16).contiguous()


def kernel_embed_dim(d: int) -> int:
    """Width the kernels run an embed_dim at: the next multiple of 128 up to 384 (128 and 256 as they are; the reference's
    argparse default 300, training/args.py:19, zero-padded to 384; 64 to 128) - the cell head, the biLSTM and the ranking kernel
    are built for multiples of 128.  Zero weight rows / columns and zero biases keep every padded channel EXACTLY 0 through Linear + ReLU,
    max / mean aggregation, F.normalize, the kNN distances (they add +0) and the LSTM cell (i = f = o = 1/2, g = 0: c = h = 0),
    so the first `d` columns are the unpadded model's outputs."""
    if 1 <= d <= 384:
        return (d + 127) // 128 * 128
    raise NotImplementedError(f"embed_dim={d}: the kernels are built for 128, 256 and (zero-padded) anything up to 384")


def _pad_to(t: torch.Tensor, rows: int = None, cols: int = None) -> torch.Tensor:
    """Zero-pad a 1-d / 2-d tensor to [rows][cols] (None = keep)."""
    if t.dim() == 1:
        n = cols if cols is not None else rows
        if n is None or n == t.shape[0]:
            return t
        out = torch.zeros(n, dtype=t.dtype, device=t.device)
        out[: t.shape[0]] = t
        return out.contiguous()
    r = t.shape[0] if rows is None else rows
    c = t.shape[1] if cols is None else cols
    if (r, c) == tuple(t.shape):
        return t
    out = torch.zeros((r, c), dtype=t.dtype, device=t.device)
    out[: t.shape[0], : t.shape[1]] = t
    return out.contiguous()


def _pad_cell_pack(p: Dict[str, object], d: int, dp: int, cell_head: bool):
    """embed_dim d -> dp in every tensor of the fp32 pack that has an embed_dim-sized axis (k-major [in][out] matrices)."""
    p["pn_w"], p["pn_b"] = _pad_to(p["pn_w"], cols=dp), _pad_to(p["pn_b"], cols=dp)
    for pre in ("col", "pos"):
        p[pre + "_w2"], p[pre + "_b2"] = _pad_to(p[pre + "_w2"], cols=dp), _pad_to(p[pre + "_b2"], cols=dp)
    nfeat = p["merge_w"].shape[0] // d                      # concat slots [class | color | position], d rows each
    m = torch.zeros((nfeat * dp, dp), dtype=p["merge_w"].dtype, device=p["merge_w"].device)
    for f in range(nfeat):
        m[f * dp: f * dp + d, :d] = p["merge_w"][f * d: (f + 1) * d]
    p["merge_w"], p["merge_b"] = m.contiguous(), _pad_to(p["merge_b"], cols=dp)
    for name in ("class_embedding", "color_embedding"):
        p[name] = _pad_to(p[name], cols=dp)
    if cell_head:
        for name in ("g_wp", "g_wq", "g_w2", "lin_w1", "lin_w2"):
            p[name] = _pad_to(p[name], rows=dp, cols=dp)
        for name in ("g_bp", "g_b2", "lin_b1", "lin_b2"):
            p[name] = _pad_to(p[name], cols=dp)


def pack_cell_weights(model, device, x3: bool = True) -> Dict[str, object]:
    """model: CellRetrievalNetwork or SuperGlueMatch (this package; the latter has no cell head).  Returns name -> fp32
    device tensor(s) for ops.make_cell_weights.  x3=False leaves the f16x3 images out (precision="fp32": weights outside
    fp16's range are then fine)."""
    p = _pack_cell_weights_fp32(model, device)
    d, dp = int(model.embed_dim), int(getattr(model, "kernel_dim", model.embed_dim))
    if dp != d:
        _pad_cell_pack(p, d, dp, hasattr(model, "graph1"))
    if x3:
        _add_x3_images(p, device, hasattr(model, "graph1"))
    return p


def _add_x3_images(p: Dict[str, object], device, cell_head: bool):
    scales = [f16x3_scale(t) for t in p["sa_w2"]]
    p["sa_w2_scale"] = scales
    p["sa_w2_x3"] = [pack_f16x3_scaled(t, sc).to(device) for t, sc in zip(p["sa_w2"], scales)]
    p["sa_b2_x3"] = [f32(b.double() * sc).to(device) for b, sc in zip(p["sa_b2"], scales)]
    # rows of the layer-1 matrices: [features C | xyz | zero rows up to C + 32].  The f16x3 kernels step 16 k per MFMA, so
    # their images (and their K) stop at C + 16: one MFMA step of zeros less per row batch
    assert all(float(t[-16:].abs().max()) == 0.0 for t in p["sa_w1"][1:] + [p["ga_w1"]])
    p["sa_w1_x3"] = [None] + [pack_f16x3(t[:-16]).to(device) for t in p["sa_w1"][1:]]
    p.update(ga_w1_x3=pack_f16x3(p["ga_w1"][:-16]).to(device), ga_w2_x3=pack_f16x3(p["ga_w2"]).to(device))
    # fp16-range guard: GA layer 1's output is bounded by ||W||_1 (largest column sum of |w|) * max|input| + max|b|
    p["ga_w1_l1"] = float(p["ga_w1"].double().abs().sum(0).max())
    p["ga_b1_absmax"] = float(p["ga_b1"].double().abs().max())
    # ... and the layer-1 tables that depend on the inputs only (position rows = the last three real rows of each sa_w1)
    feat = (3, 64, 128)
    p["sa_wp_l1"] = [float(w[c: c + 3].double().abs().sum(0).max()) for w, c in zip(p["sa_w1"], feat)]
    p["sa_a1_l1"] = float(p["sa_w1"][0].double().abs().sum(0).max())
    p["sa_b1_absmax"] = float(p["sa_b1"][0].double().abs().max())
    for name, src in (("lin1", "lin1_w"), ("lin2", "lin2_w"), ("pn", "pn_w"), ("merge", "merge_w")) + \
            ((("g_wp", "g_wp"), ("g_wq", "g_wq")) if cell_head else ()):
        sc = f16x3_scale(p[src])
        p[name + "_scale"], p[name + "_x3"] = sc, pack_gemm_x3(p[src], sc).to(device)
    if cell_head and p["g_w2"].shape[0] % 32 == 0 and p["g_w2"].shape[1] % 32 == 0:
        p["g_w2_x3"] = pack_f16x3(p["g_w2"]).to(device)


def _pack_cell_weights_fp32(model, device) -> Dict[str, object]:
    oe, pn = model.object_encoder, model.object_encoder.pointnet
    p: Dict[str, object] = {}
    sa_w1, sa_b1, sa_w2, sa_b2 = [], [], [], []
    for sa, kpad in ((pn.sa1, 6), (pn.sa2, 96), (pn.sa3, 160)):
        nn_ = sa.point_conv.local_nn
        w1, b1 = fold_linear_bn(nn_[0])
        w2, b2 = fold_linear_bn(nn_[1])
        sa_w1.append(kmajor(w1, kpad).to(device))
        sa_b1.append(f32(b1).to(device))
        sa_w2.append(kmajor(w2).to(device))
        sa_b2.append(f32(b2).to(device))
    p.update(sa_w1=sa_w1, sa_b1=sa_b1, sa_w2=sa_w2, sa_b2=sa_b2)
    w1, b1 = fold_linear_bn(pn.ga.mlp[0])
    w2, b2 = fold_linear_bn(pn.ga.mlp[1])
    p.update(ga_w1=kmajor(w1, 288).to(device), ga_b1=f32(b1).to(device), ga_w2=kmajor(w2).to(device),
             ga_b2=f32(b2).to(device))
    for name, lin in (("lin1", pn.lin1), ("lin2", pn.lin2)):
        p[name + "_w"] = kmajor(lin.weight.detach().double()).to(device)
        p[name + "_b"] = f32(lin.bias.detach().double()).to(device)
    w, b = fold_linear_bn(oe.mlp_pointnet[0])
    p.update(pn_w=kmajor(w).to(device), pn_b=f32(b).to(device))
    for pre, enc in (("col", oe.color_encoder), ("pos", oe.pos_encoder)):
        w1, b1 = fold_linear_bn(enc[0])
        w2, b2 = fold_linear_bn(enc[1])
        p[pre + "_w1"], p[pre + "_b1"] = kmajor(w1).to(device), f32(b1).to(device)
        p[pre + "_w2"], p[pre + "_b2"] = kmajor(w2).to(device), f32(b2).to(device)
    w, b = fold_linear_bn(oe.mlp_merge[0])
    p.update(merge_w=kmajor(w).to(device), merge_b=f32(b).to(device))
    p.update(class_embedding=f32(oe.class_embedding.weight.detach()).to(device),
             color_embedding=f32(oe.color_embedding.weight.detach()).to(device))
    if not hasattr(model, "graph1"):  # fine stage: ObjectEncoder only
        return p
    # DynamicEdgeConv nn on [x_i | x_j - x_i]:  W1a x_i + W1b (x_j - x_i) = (W1a - W1b) x_i + W1b x_j
    d = model.embed_dim
    w1, b1 = fold_linear_bn(model.graph1.nn[0])
    w2, b2 = fold_linear_bn(model.graph1.nn[1])
    p.update(g_wp=kmajor(w1[:, :d] - w1[:, d:]).to(device), g_bp=f32(b1).to(device), g_wq=kmajor(w1[:, d:]).to(device),
             g_w2=kmajor(w2).to(device), g_b2=f32(b2).to(device))
    w1, b1 = fold_linear_bn(model.lin[0])
    w2, b2 = fold_linear_bn(model.lin[1])
    p.update(lin_w1=kmajor(w1).to(device), lin_b1=f32(b1).to(device), lin_w2=kmajor(w2).to(device),
             lin_b2=f32(b2).to(device))
    return p


def pack_match_weights(model, device, precision: str = "f16x3") -> Dict[str, object]:
    """model: SuperGlueMatch (this package).  Conv1d(k=1) weights [out, in, 1] -> k-major [in][out]; eval BatchNorm of
    AttentionalPropagation.mlp folded into its first conv (float64 on the host)."""
    sg = model.superglue
    cols = {k: [] for k in ("wqkv", "bqkv", "wm", "bm", "w1", "b1", "w2", "b2")}
    conv = lambda c: (c.weight.detach().double().squeeze(-1), c.bias.detach().double())
    for layer in sg.gnn.layers:
        ws, bs = zip(*[conv(c) for c in layer.attn.proj])
        cols["wqkv"].append(torch.cat([w.t() for w in ws], dim=1))   # [D][3D]: q | k | v
        cols["bqkv"].append(torch.cat(bs))
        w, b = conv(layer.attn.merge)
        cols["wm"].append(w.t())
        cols["bm"].append(b)
        w, b = conv(layer.mlp[0])
        bn = layer.mlp[1]
        sc = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
        cols["w1"].append((w * sc[:, None]).t())
        cols["b1"].append((b - bn.running_mean.detach().double()) * sc + bn.bias.detach().double())
        w, b = conv(layer.mlp[3])
        cols["w2"].append(w.t())
        cols["b2"].append(b)
    d = model.embed_dim
    empty = {"wqkv": (0, d, 3 * d), "bqkv": (0, 3 * d), "wm": (0, d, d), "bm": (0, d), "w1": (0, 2 * d, 2 * d),
             "b1": (0, 2 * d), "w2": (0, 2 * d, d), "b2": (0, d)}
    p: Dict[str, object] = {k: (f32(torch.stack(v)) if v else torch.zeros(empty[k], dtype=torch.float32)).to(device)
                            for k, v in cols.items()}
    w, b = conv(sg.final_proj)
    p.update(wf=f32(w.t()).to(device), bf=f32(b).to(device), bin_score=float(sg.bin_score.detach()))
    l1, l2 = model.mlp_offsets[0], model.mlp_offsets[2]
    p.update(wo1=f32(l1.weight.detach().double().t()).to(device), bo1=f32(l1.bias.detach().double()).to(device),
             wo2=f32(l2.weight.detach().double().t()).to(device), bo2=f32(l2.bias.detach().double()).to(device))
    p["cross"] = [1 if n == "cross" else 0 for n in sg.gnn.names]
    if precision == "f16x3":
        for name, sc in (("wqkv", "qkv"), ("wm", "m"), ("w1", "1"), ("w2", "2"), ("wf", "f")):
            mats = p[name] if p[name].dim() == 3 else p[name][None]
            scale = f16x3_scale(mats) if mats.numel() else 1.0
            p["scale_" + sc] = scale
            imgs = [pack_gemm_x3(m, scale) for m in mats.cpu()]
            p[name + "_x3"] = (torch.stack(imgs) if imgs else torch.zeros((0,), dtype=torch.int16)).contiguous().to(device)
    return p


def pack_text_weights(lang, device, x3: bool = False, pad_to: int = None) -> Dict[str, torch.Tensor]:
    """lang: LanguageEncoder (this package).  nn.LSTM parameter layout: weight_ih_l0 [4D, D] (gates i,f,g,o).
    x3: also the scaled f16x3 images of the two recurrent matrices (csrc/lstm.hip: k_bilstm_x3), one power-of-two scale.
    pad_to: hidden width the kernel runs (kernel_embed_dim): every gate block, the input axis and the embedding rows are
    zero-padded to it."""
    lstm = lang.lstm
    d = lstm.hidden_size
    dp = d if pad_to is None else int(pad_to)

    def gates_kmajor(w):           # [4D, D_in] -> k-major [Dp_in][4 Dp] with gate g's columns at g Dp ...
        w = w.detach().double()
        out = torch.zeros((dp, 4 * dp), dtype=torch.float64)
        for g in range(4):
            out[:d, g * dp: g * dp + d] = w[g * d: (g + 1) * d, :].t().cpu()
        return out

    def gates_vec(b):
        b = b.detach().double().cpu()
        out = torch.zeros(4 * dp, dtype=torch.float64)
        for g in range(4):
            out[g * dp: g * dp + d] = b[g * d: (g + 1) * d]
        return out

    w_ih = torch.stack([gates_kmajor(lstm.weight_ih_l0), gates_kmajor(lstm.weight_ih_l0_reverse)])
    w_hh = torch.stack([gates_kmajor(lstm.weight_hh_l0), gates_kmajor(lstm.weight_hh_l0_reverse)])
    bias = torch.stack([gates_vec(lstm.bias_ih_l0) + gates_vec(lstm.bias_hh_l0),
                        gates_vec(lstm.bias_ih_l0_reverse) + gates_vec(lstm.bias_hh_l0_reverse)])
    emb = lang.word_embedding.weight.detach()
    p = dict(embedding=_pad_to(f32(emb), cols=dp).to(device), w_ih=f32(w_ih).to(device),
             w_hh=f32(w_hh).to(device), bias=f32(bias).to(device))
    if x3:
        w32 = f32(w_hh)
        sc = f16x3_scale(w32)
        p["w_hh_scale"] = sc
        p["w_hh_x3"] = torch.stack([pack_f16x3_scaled(w32[0], sc), pack_f16x3_scaled(w32[1], sc)]).contiguous().to(device)
    return p


def params_version(module: nn.Module) -> Tuple:
    """Cheap change detector for cached packs: (data_ptr, _version) of every parameter and buffer.  Runs once per encode call, so the
    walk goes over the modules' own dictionaries (nn.Module.parameters() / buffers() spend ~0.5 ms per call on a coarse model in
    generator recursion and de-duplication - a quarter of a 64-cell call); a tensor registered twice is simply listed twice."""
    out = []
    stack = [module]
    while stack:
        m = stack.pop()
        for t in m._parameters.values():
            if t is not None:
                out.append((t.data_ptr(), t._version))
        for t in m._buffers.values():
            if t is not None:
                out.append((t.data_ptr(), t._version))
        for c in m._modules.values():
            if c is not None:
                stack.append(c)
    return tuple(out)

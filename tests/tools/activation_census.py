"""Per-layer activation ranges of a coarse checkpoint, read off the CPU oracle (test infrastructure: the oracle is the checker).

The f16x3 path of the HIP library splits every MFMA operand into two fp16 pieces, so a layer's activations must stay inside
what the pieces cover: below 65504 (fp16's largest value) and - for a layer's LARGEST activation - above 2^-7 (below that the
low pieces underflow and the layer keeps fewer than 18 bits of its own scale).  The library's guard word checks
exactly that on every call; this census says how FAR a checkpoint is from either edge, layer by layer: forward hooks on every
BatchNorm1d of the oracle (get_mlp blocks are Linear -> BatchNorm1d -> ReLU, models/modules.py:21-29, so a BatchNorm's output is
the pre-ReLU value the next operand is made of) over a sample of cells."""
import numpy as np
import torch

FP16_MAX = 65504.0
LOW_EDGE = 2.0 ** -7


def census(oracle_model, xyz, rgb, center, mean_rgb, cell_ptr):
    rows, handles = {}, []

    def hook(name):
        def fn(_m, _inp, out):
            r = rows.setdefault(name, {"max_abs": 0.0, "max_pos": 0.0, "rows": 0})
            r["max_abs"] = max(r["max_abs"], float(out.abs().max()))
            r["max_pos"] = max(r["max_pos"], float(out.clamp(min=0).max()))
            r["rows"] += int(out.shape[0])
        return fn
    for name, mod in oracle_model.named_modules():
        if isinstance(mod, torch.nn.BatchNorm1d):
            handles.append(mod.register_forward_hook(hook(name)))
    try:
        with torch.no_grad():
            oracle_model.encode_objects_packed(xyz, rgb, center, mean_rgb, cell_ptr)
    finally:
        for h in handles:
            h.remove()
    for r in rows.values():
        r["headroom_to_fp16_max_log2"] = float(np.log2(FP16_MAX / max(r["max_abs"], 1e-30)))
        r["largest_activation_over_low_edge_log2"] = float(np.log2(max(r["max_pos"], 1e-30) / LOW_EDGE))
    return rows


def summary(rows):
    worst_hi = min(rows.items(), key=lambda kv: kv[1]["headroom_to_fp16_max_log2"])
    worst_lo = min(rows.items(), key=lambda kv: kv[1]["largest_activation_over_low_edge_log2"])
    return {"layers": len(rows),
            "closest_to_fp16_max": {"layer": worst_hi[0], "max_abs": worst_hi[1]["max_abs"], "headroom_bits": worst_hi[1]["headroom_to_fp16_max_log2"]},
            "closest_to_low_edge": {"layer": worst_lo[0], "largest_activation": worst_lo[1]["max_pos"],
                                    "bits_above_2^-7": worst_lo[1]["largest_activation_over_low_edge_log2"]}}

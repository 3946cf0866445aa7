"""Deterministic model weights for the golden fixtures (test infrastructure).

The fixtures store only inputs and expected outputs; the weights are a pure function of (seed, parameter name, flat
index) so that the reference model (at fixture-generation time), the CPU oracle and the HIP module (at test time)
can all be filled identically without shipping multi-megabyte state_dicts or depending on torch's RNG stream.
"""
import zlib

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(x):
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return x ^ (x >> np.uint64(31))


def _uniform(seed: int, name: str, n: int) -> np.ndarray:
    """n values in (-1, 1), float64."""
    base = _mix(np.uint64(seed) ^ np.uint64(zlib.crc32(name.encode())))
    k = _mix(base ^ _mix(np.arange(n, dtype=np.uint64)))
    return ((k >> np.uint64(11)).astype(np.float64) + 0.5) / float(1 << 52) - 1.0


def fill_state_dict(module: torch.nn.Module, seed: int) -> None:
    """Overwrite every parameter and buffer of `module` in place."""
    sd = module.state_dict()
    new = {}
    for name, t in sd.items():
        n = t.numel()
        if name.endswith("num_batches_tracked"):
            new[name] = torch.zeros_like(t)
            continue
        u = _uniform(seed, name, n).reshape(tuple(t.shape))
        if t.dim() == 0:                           # scalar parameter (SuperGlue.bin_score)
            new[name] = torch.tensor(1.0 + 0.5 * float(u), dtype=t.dtype)
            continue
        if name.endswith("running_var"):
            v = 1.0 + 0.5 * u                      # (0.5, 1.5)
        elif name.endswith("running_mean"):
            v = 0.1 * u
        elif ".1.weight" in name and t.dim() == 1:  # BatchNorm gamma inside get_mlp blocks
            v = 1.0 + 0.2 * u
        elif name.endswith("bias"):
            v = 0.1 * u
        elif "embedding" in name:
            v = u.copy()
            v[0] = 0.0                              # padding_idx = 0
        else:                                       # Linear / LSTM weight [out, in]
            fan_in = t.shape[1] if t.dim() == 3 else t.shape[-1]   # Conv1d [out, in, 1] / Linear [out, in]
            v = u / np.sqrt(fan_in)
            if "lstm" in name:
                v = u / np.sqrt(t.shape[-1]) * 1.5
            if name.endswith("final_proj.weight"):  # SuperGlue: peaked scores, so that real matches (and rejections)
                v = v * 8.0                          # occur with random weights (4-5 of 6 hints matched)
        new[name] = torch.from_numpy(np.ascontiguousarray(v)).to(t.dtype)
    module.load_state_dict(new, strict=True)

"""Error study (CPU, seconds): could the two CORRECTION products of f16x3 (hi.lo + lo.hi) run on the block-scaled fp8 MFMA
(twice the f16 rate: 1.5 x on the matrix time of SA2 / SA3 / GA2)?  They need ~11 bits of relative accuracy to deliver 22 in
total; MX e4m3 (per-32-element power-of-two scale, 4 significant bits per element) gives them 4.

One layer-2-shaped product, [4096 x 256] post-ReLU activations x [256 x 256] weights, every form against float64:

    fp32 matmul                 rms err / rms out 2.2e-07   max 2.6e-06
    f16x3                       rms err / rms out 2.8e-07   max 1.4e-06
    f16 hi.hi + fp8 corrections rms err / rms out 1.5e-05   max 1.0e-04
    f16 hi.hi only              rms err / rms out 3.0e-04   max 1.7e-03

50 x the error of f16x3 per layer; over the ~10 layers in front of the object embeddings (checked at 1e-4 ABSOLUTE on O(1)
values, 2.6e-5 today) that does not hold the bar, and every extra 1e-5 multiplies the DynamicEdgeConv near-tie flips.
Not a default; at most an opt-in "fast" precision.  (DESIGN.md "Next", docs/notebook.md round 4.)"""
import torch

torch.manual_seed(0)
K, N, M = 256, 256, 4096
x = torch.relu(torch.randn(M, K, dtype=torch.float64) + 0.2).float()
w = (torch.randn(K, N, dtype=torch.float64) / 16).float()
exact = x.double() @ w.double()
rms = exact.std().item()


def split16(a):
    hi = a.half()
    return hi, (a - hi.float()).half()


def mm(a, b):
    return a.double() @ b.double()


def q8_mx(a, dim):
    """MX e4m3: per-32-block power-of-two scale along `dim`, elements rounded to float8_e4m3fn."""
    a = a.float()
    r, c = a.shape
    blk = a.reshape(r, c // 32, 32) if dim == 1 else a.t().reshape(c, r // 32, 32)
    amax = blk.abs().amax(-1, keepdim=True)
    scale = torch.exp2(torch.floor(torch.log2(amax.clamp_min(1e-30))) - 8)
    q = (blk / scale).clamp(-448, 448).to(torch.float8_e4m3fn).float() * scale
    return q.reshape(r, c) if dim == 1 else q.reshape(c, r).t()


xh, xl = split16(x)
wh, wl = split16(w)
forms = {"fp32 matmul": (x @ w).double(),
         "f16x3": mm(xh, wh) + mm(xh, wl) + mm(xl, wh),
         "f16 hi.hi + fp8 corrections": mm(xh, wh) + mm(q8_mx(xh.float(), 1), q8_mx(wl.float(), 0)) + mm(q8_mx(xl.float(), 1), q8_mx(wh.float(), 0)),
         "f16 hi.hi only": mm(xh, wh)}
for name, y in forms.items():
    e = y - exact
    print(f"{name:28s} rms err / rms out = {e.std().item() / rms:.1e}   max |err| / rms out = {e.abs().max().item() / rms:.1e}")

# fp32 products of the training path: old (t2p_gemm / t2p_gemm_tn) against new (t2p_linear_f32 / t2p_linear_wgrad_f32), 64-cell batch shapes
import sys, time, torch
sys.path.insert(0, '/root/repo')
import text2pos_amd
from text2pos_amd import ops
dev = torch.device('cuda:0')
def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
shapes = [("SA1.1", 1390000, 8, 32), ("SA1.2", 1390000, 32, 64), ("SA2.1", 1000000, 72, 128), ("SA2.2", 1000000, 128, 128),
          ("SA3.1", 390000, 136, 256), ("SA3.2", 390000, 256, 256), ("GA.1", 32768, 264, 512), ("GA.2", 32768, 512, 1024),
          ("head", 1024, 1024, 512), ("merge", 1024, 768, 256), ("knn", 8192, 512, 256)]
tot = dict(of=0, nf=0, ox=0, nx=0, ow=0, nw=0)
for name, m, k, n in shapes:
    a = torch.randn(m, k, device=dev); w = torch.randn(k, n, device=dev) / k ** 0.5; b = torch.randn(n, device=dev)
    dy = torch.randn(m, n, device=dev); wt = w.t().contiguous()
    kp, npad = (k + 3) // 4 * 4, (n + 7) // 8 * 8
    fl = 2.0 * m * k * n / 1e9
    of = timeit(lambda: ops.gemm(a, w, b)) if n % 8 == 0 else float('nan')
    nf = of
    ox = timeit(lambda: ops.gemm(dy, wt)) if k % 8 == 0 else float('nan')
    nx = ox
    ow = timeit(lambda: (ops.gemm_tn(dy, a), dy.sum(0)))
    nw = timeit(lambda: ops.linear_wgrad(dy, a))
    for key, v in zip(tot, (of, nf, ox, nx, ow, nw)):
        tot[key] += 0 if v != v else v
    print(f"{name:6s} M={m:8d} K={k:4d} N={n:4d} | fwd {of:7.3f} -> {nf:7.3f} ms ({fl/nf:6.1f} TF/s) | dx {ox:7.3f} -> {nx:7.3f} ms ({fl/nx:6.1f}) | dw+db {ow:7.3f} -> {nw:7.3f} ms ({fl/nw:6.1f})", flush=True)
print("sum", {k: round(v, 2) for k, v in tot.items()})

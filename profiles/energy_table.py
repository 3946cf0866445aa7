"""Per-kernel energy table of the cell branch (VERDICT r04 #3a): which kernels cost joules, which only milliseconds.

    python profiles/energy_table.py [--seconds 2.5] [--out gpurun_out/r05_energy.md]      (GPU box, repo root)

The step is power-limited under matrix load, so a shorter step needs fewer joules - and a kernel's share of the
step's ENERGY, not of its time, says what it is worth attacking and what is free to overlap.  A 5 ms kernel cannot be resolved
by a 20 Hz power sensor inside a 90 ms step; instead each of the large kernels is held on the chip ALONE for >= `seconds`:
t2p_profile_repeat (include/t2p.h) makes the library issue that kernel's launch N times back to back - same arguments, same
results - inside an otherwise normal single-stream pass over the benchmark's 12,000 cells, while a sampler thread reads the
package power and the shader clock from the GPU's hwmon files (power1_input, freq1_input) every 50 ms.  The mean over the
window in which the repeated kernel runs (first and last 200 ms cut) is that kernel's power; times its per-step duration
(hipEvents, same run) = joules per step.  The whole step looped the same way gives the step's power for comparison.
"""
import argparse
import glob
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

KERNELS = [  # (profile scope, kernel, what)
    ("sample_group", "k_sample_group<true>", "FPS + ball query, 3 levels"),
    ("dedup_rows", "k_dedup_rows", "SA1 repeated-point rows dropped"),
    ("ws_edge_sa_k32_n64", "k_sa_points<12>", "SA1 both layers + max"),
    ("ws_dense_k80_n128", "k_ws<80,128,...> DENSE_STORE", "SA2 layer-1 point table"),
    ("ws_edge_sa_k128_n128", "k_sa_rows<128,128,64,4>", "SA2 layer 2 + max"),
    ("ws_dense_k144_n256", "k_ws<144,256,...> DENSE_STORE", "SA3 layer-1 point table"),
    ("ws_edge_sa_k256_n256", "k_sa3", "SA3 layer 2 + max (dominant)"),
    ("ws_dense_k272_n512", "k_ws<272,256,...> DENSE_STORE split", "GA layer 1 (fp16 planes out)"),
    ("ws_groupmax_k512_n1024", "k_ga2", "GA layer 2 + max"),
]


def hwmon_of_device(torch, index=0):
    """hwmon directory of HIP device `index` (matched by PCI bus address; the node's other GPUs are visible in sysfs too)."""
    p = torch.cuda.get_device_properties(index)
    want = None
    if hasattr(p, "pci_bus_id"):
        want = "%04x:%02x:%02x" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, getattr(p, "pci_device_id", 0))
    cands = []
    for h in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        bdf = os.path.basename(os.path.realpath(os.path.join(h, "..", "..")))
        cands.append((bdf, h))
    for bdf, h in cands:
        if want and bdf.lower().startswith(want):
            return h, bdf
    return None, want


class Sampler(threading.Thread):
    def __init__(self, hwmon, period=0.05):
        super().__init__(daemon=True)
        self.hwmon, self.period, self.rows, self.stop_flag = hwmon, period, [], False

    def read(self):
        def rd(name):
            with open(os.path.join(self.hwmon, name)) as f:
                return float(f.read().strip())
        return rd("power1_input") * 1e-6, rd("freq1_input") * 1e-6

    def run(self):
        while not self.stop_flag:
            t = time.perf_counter()
            try:
                w, mhz = self.read()
                self.rows.append((t, w, mhz))
            except OSError:
                pass
            time.sleep(max(0.0, self.period - (time.perf_counter() - t)))

    def window(self, t0, t1, cut=0.2):
        r = [(w, f) for t, w, f in self.rows if t0 + cut <= t <= t1 - cut]
        if not r:
            return None
        a = np.array(r)
        return dict(samples=len(r), w_mean=float(a[:, 0].mean()), w_max=float(a[:, 0].max()), mhz_mean=float(a[:, 1].mean()),
                    mhz_min=float(a[:, 1].min()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=2.5)
    ap.add_argument("--cells", type=int, default=12000)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r05_energy.md"))
    a = ap.parse_args()
    import bench
    import torch
    import text2pos_amd as t2p
    from text2pos_amd import ops, synthetic as S
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    hwmon, bdf = hwmon_of_device(torch)
    if hwmon is None:
        raise SystemExit(f"no hwmon directory for the GPU at {bdf}")
    xyz, rgb, center, mean_rgb, cell_ptr = bench.generate_cells(S, bench.SEED, a.cells, 0, a.cells, max(1, min(64, os.cpu_count() or 1)))
    torch.manual_seed(1234)
    model = t2p.CellRetrievalNetwork(S.LABELS + ["pad"], S.COLOR_NAMES, S.known_words(), S.default_args())
    model = model.to(dev).eval()
    d = [torch.from_numpy(x).to(dev) for x in (xyz, rgb, center, mean_rgb)]
    d_ptr = torch.from_numpy(cell_ptr).to(dev)
    # BatchNorm statistics that match the data (as bench.py): realistic activation VALUES matter for MFMA power
    bns = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm1d)]
    for m in bns:
        m.reset_running_stats()
        m.momentum = None
    model.train()
    o64 = int(cell_ptr[64])
    with torch.no_grad():
        model.encode_objects_packed(*(t[:o64] for t in d), cell_ptr[:65])
    model.eval()
    for m in bns:
        m.momentum = 0.1

    def encode():
        with torch.no_grad():
            out = model.encode_objects_packed(*d, cell_ptr, d_ptr, check_overflow=False, streams=1)
        torch.cuda.synchronize()
        return out

    ref = encode()
    # per-kernel time per step (hipEvents), plain pass
    ops.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(3):
        encode()
    step_ms = (time.perf_counter() - t0) / 3 * 1e3
    ops.profile_enable(False)
    prof = ops.profile_report()
    ms = {k: v[1] / 3 for k, v in prof.items()}
    launches = {k: v[0] // 3 for k, v in prof.items()}
    smp = Sampler(hwmon)
    smp.start()
    time.sleep(1.0)
    t_idle = (time.perf_counter() - 1.0, time.perf_counter())
    rows = []
    for scope, kern, what in KERNELS:
        if scope not in ms or ms[scope] <= 0:
            print("skip", scope, "(not in the profile)", sorted(ms))
            continue
        reps = max(2, int(np.ceil(a.seconds * 1e3 / ms[scope])))
        ops.profile_repeat(scope, reps)
        t0 = time.perf_counter()
        out = encode()
        t1 = time.perf_counter()
        ops.profile_repeat(None, 1)
        assert torch.equal(out, ref), scope          # repeated launches change no result
        # the repeated kernel runs in `launches` bursts separated by the rest of the chunk's kernels: with reps x ms >> 90 ms
        # the window is > 95 % that kernel
        w = smp.window(t0, t1)
        rows.append(dict(scope=scope, kernel=kern, what=what, ms=ms[scope], launches=launches[scope], reps=reps,
                         wall=t1 - t0, **(w or {})))
        print(rows[-1], flush=True)
    # the whole step, looped
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < max(3.0, a.seconds):
        encode()
        n += 1
    t1 = time.perf_counter()
    w_step = smp.window(t0, t1)
    step_loop_ms = (t1 - t0) / n * 1e3
    smp.stop_flag = True
    smp.join()
    idle = smp.window(*t_idle, cut=0.0)
    lines = [f"# r05 per-kernel energy table: encode of {a.cells} cells ({int(cell_ptr[-1])} objects), one HIP stream, f16x3", "",
             f"GPU {bdf}; sampler: hwmon power1_input / freq1_input every 50 ms; each kernel repeated in place (t2p_profile_repeat) for "
             f">= {a.seconds} s per pass; idle {idle['w_mean']:.0f} W." if idle else "", "",
             "| kernel | role | ms / step | launches / step | W mean | W max | sclk MHz mean | J / step | share of sum J | share of sum ms |",
             "|---|---|---:|---:|---:|---:|---:|---:|---:|---:|"]
    tot_j = sum(r["ms"] * 1e-3 * r.get("w_mean", 0.0) for r in rows)
    tot_ms = sum(r["ms"] for r in rows)
    for r in rows:
        j = r["ms"] * 1e-3 * r.get("w_mean", 0.0)
        lines.append(f"| `{r['kernel']}` | {r['what']} | {r['ms']:.2f} | {r['launches']} | {r.get('w_mean', 0):.0f} | {r.get('w_max', 0):.0f} | "
                     f"{r.get('mhz_mean', 0):.0f} | {j:.2f} | {100 * j / tot_j:.1f} % | {100 * r['ms'] / tot_ms:.1f} % |")
    lines += ["", f"Sum over these kernels: {tot_ms:.1f} ms, {tot_j:.1f} J per step.  Whole cell-encoder pass looped: {step_loop_ms:.1f} ms per pass at "
              f"{w_step['w_mean']:.0f} W mean / {w_step['w_max']:.0f} W max, sclk {w_step['mhz_mean']:.0f} MHz mean = {step_loop_ms * 1e-3 * w_step['w_mean']:.1f} J per pass "
              f"(plain pass with per-kernel events: {step_ms:.1f} ms)." if w_step else ""]
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    open(a.out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()

"""Builds profiles/<tag>_pmc_traffic.json from two rocprofv3 PMC passes of the same bench command:

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <dir>/fetch -o p -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d <dir>/write -o p -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline
    python profiles/pmc_traffic.py <dir>/fetch <dir>/write profiles/r01_d_pmc_traffic.json

(separate passes: FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2 -- MI355X_MICROARCH.md, "rocprofv3 PMC slots").
Units and corrections as the guide's HBM section prescribes: both counters are in KiB; on gfx950 FETCH_SIZE reports half
of the fetched bytes (x2).  Per-launch averages per kernel.
"""
import collections
import csv
import glob
import json
import re
import sys


def short(k):
    m = re.search(r"(k_\w+(<[^>]*>)?)", k)
    return m.group(1) if m else k[:40]


def per_kernel(d, counter):
    """kernel -> list of per-dispatch byte counts, in dispatch order"""
    f = glob.glob(f"{d}/**/p_counter_collection.csv", recursive=True)[0]
    vals = collections.defaultdict(list)
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    for r in rows:
        vals[short(r["Kernel_Name"])].append(float(r["Counter_Value"]) * 1024.0)
    return vals


def main():
    fdir, wdir, out = sys.argv[1:4]
    fv, wv = per_kernel(fdir, "FETCH_SIZE"), per_kernel(wdir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fv) | set(wv)):
        f, w = fv.get(k, []), wv.get(k, [])
        n = max(len(f), len(w))
        f, w = f + [0.0] * (n - len(f)), w + [0.0] * (n - len(w))
        tot = [2 * a + b for a, b in zip(f, w)]          # the two passes run the same command: dispatch i is the same launch
        # bench.py launches every cell kernel on small batches too (BatchNorm calibration, phase timings): the per-launch figure
        # that belongs next to `roofline.avg_launch_ms` is the mean over the FULL-SIZE launches (>= 3/4 of the largest)
        full = [i for i, t in enumerate(tot) if t >= 0.75 * max(tot)] if tot and max(tot) > 0 else []
        mean = lambda xs: sum(xs) / len(xs) if xs else 0.0
        kernels[k] = {"launches": n, "full_size_launches": len(full),
                      "fetch_bytes_per_launch_raw": mean([f[i] for i in full]),
                      "fetch_bytes_per_launch_corrected_x2": 2 * mean([f[i] for i in full]),
                      "write_bytes_per_launch": mean([w[i] for i in full]),
                      "hbm_bytes_per_launch": mean([tot[i] for i in full]),
                      "hbm_bytes_per_launch_all_launches": mean(tot)}
    json.dump({"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py "
                          "--steps 1 --warmup 1 --no-cpu-baseline --no-fp32-pass --cell-streams 1 --no-extras --no-dropin; per-launch figures = mean over the full-size launches",
               "units": "bytes; FETCH_SIZE / WRITE_SIZE are reported in KiB; FETCH_SIZE doubled per the gfx950 note in "
                        "MI355X_MICROARCH.md (HBM section); WRITE_SIZE uncalibrated",
               "kernels": kernels}, open(out, "w"), indent=1)
    for k in sorted(kernels, key=lambda k: -kernels[k]["hbm_bytes_per_launch"])[:12]:
        print(f"{k:40s} {kernels[k]['launches']:4d} launches  {kernels[k]['hbm_bytes_per_launch'] / 1e9:7.2f} GB per launch")


if __name__ == "__main__":
    main()

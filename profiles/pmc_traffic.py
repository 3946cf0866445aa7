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
    f = glob.glob(f"{d}/**/p_counter_collection.csv", recursive=True)[0]
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        tot[k] += float(r["Counter_Value"]) * 1024.0
        cnt[k] += 1
    return tot, cnt


def main():
    fdir, wdir, out = sys.argv[1:4]
    ft, fc = per_kernel(fdir, "FETCH_SIZE")
    wt, wc = per_kernel(wdir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(ft) | set(wt)):
        n = fc.get(k) or wc.get(k)
        fetch = ft.get(k, 0.0) / max(fc.get(k, 1), 1)
        write = wt.get(k, 0.0) / max(wc.get(k, 1), 1)
        kernels[k] = {"launches": n, "fetch_bytes_per_launch_raw": fetch, "fetch_bytes_per_launch_corrected_x2": 2 * fetch,
                      "write_bytes_per_launch": write, "hbm_bytes_per_launch": 2 * fetch + write}
    json.dump({"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py "
                          "--steps 1 --warmup 1 --no-cpu-baseline --no-fp32-pass --no-two-stream",
               "units": "bytes; FETCH_SIZE / WRITE_SIZE are reported in KiB; FETCH_SIZE doubled per the gfx950 note in "
                        "MI355X_MICROARCH.md (HBM section); WRITE_SIZE uncalibrated",
               "kernels": kernels}, open(out, "w"), indent=1)
    for k in sorted(kernels, key=lambda k: -kernels[k]["hbm_bytes_per_launch"])[:12]:
        print(f"{k:40s} {kernels[k]['launches']:4d} launches  {kernels[k]['hbm_bytes_per_launch'] / 1e9:7.2f} GB per launch")


if __name__ == "__main__":
    main()

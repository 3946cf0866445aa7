"""Builds profiles/<tag>_evidence.json: the counter evidence bench.py quotes next to its live timings, stamped with the SOURCES it
was taken on.  bench.py cannot host rocprofv3, so `roofline.traffic` (HBM bytes per launch of the dominant kernel) and
`roofline.mfma_busy` come from separate rocprofv3 --pmc passes of the same command (profiles/collect.sh); this file ties them
to the kernel sources by content hash, and bench.py refuses a figure whose kernel source has changed since.

    python profiles/evidence.py <traffic.json> <sq pass dir> <grbm pass dir> <out.json> [tag]

  hbm_bytes_per_launch    2 x FETCH_SIZE + WRITE_SIZE (KiB counters; the gfx950 x2 of MI355X_MICROARCH.md), mean over the
                          full-size launches (profiles/pmc_traffic.py)
  mfma_busy               SQ_VALU_MFMA_BUSY_CYCLES / (1,024 SIMDs x the kernel's own busy cycles), the latter from the
                          GRBM_GUI_ACTIVE pass (summed over the 8 XCDs by the profiler: / 8 per XCD) - i.e. the fraction of the
                          cycles the kernel actually ran for (at whatever clock the power cap allowed) in which a SIMD's matrix
                          pipe was busy.  mfma_busy_at_2400mhz = the same count over duration x 2.4 GHz.
  whole_step              sum over ALL kernels of the run's HBM bytes / the number of steps the run executed (k_sa3 launches / 3),
                          against the compulsory 7,192 B per object (6,168 read + 1,024 written)
"""
import collections
import csv
import glob
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(k):
    m = re.search(r"(k_\w+(<[^>]*>)?)", k)
    return m.group(1) if m else k[:40]


def source_hashes():
    out = {}
    for f in sorted(glob.glob(os.path.join(ROOT, "text2pos-cvpr2022_amd", "csrc", "*.h*"))):
        out["csrc/" + os.path.basename(f)] = hashlib.sha256(open(f, "rb").read()).hexdigest()[:16]
    return out


def pass_dir(d):
    f = glob.glob(f"{d}/**/p_counter_collection.csv", recursive=True)
    return os.path.dirname(f[0]) if f else None


def counters(d):
    """kernel -> {counter: sum}, kernel -> total ns, from one rocprofv3 --kernel-trace --pmc pass"""
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    ns = collections.defaultdict(float)
    if d is None:
        return agg, ns
    for r in csv.DictReader(open(f"{d}/p_counter_collection.csv")):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for r in csv.DictReader(open(f"{d}/p_kernel_trace.csv")):
        ns[short(r["Kernel_Name"])] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return agg, ns


def main():
    traffic_json, sq_dir, grbm_dir, out = sys.argv[1:5]
    tag = sys.argv[5] if len(sys.argv) > 5 else os.path.basename(out).split("_evidence")[0]
    traffic = json.load(open(traffic_json))["kernels"]
    sq, sq_ns = counters(pass_dir(sq_dir))
    gr, gr_ns = counters(pass_dir(grbm_dir))
    kernels = {}
    for k in sorted(set(traffic) | set(sq)):
        row = {}
        if k in traffic:
            row.update(hbm_bytes_per_launch=traffic[k]["hbm_bytes_per_launch"], launches=traffic[k]["launches"],
                       full_size_launches=traffic[k]["full_size_launches"],
                       hbm_bytes_all_launches=traffic[k]["hbm_bytes_per_launch_all_launches"] * traffic[k]["launches"])
        if k in sq and sq_ns.get(k):
            busy = sq[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
            row.update(sq_pass_ms=sq_ns[k] / 1e6, mfma_busy_cycles=busy, mfma_busy_at_2400mhz=busy / (sq_ns[k] * 2.4 * 1024))
            if k in gr and gr_ns.get(k):
                per_xcd = gr[k].get("GRBM_GUI_ACTIVE", 0.0) / gr_ns[k] / 8.0      # busy cycles per ns on one XCD = GHz
                row.update(shader_clock_ghz=per_xcd, mfma_busy=busy / (sq_ns[k] * per_xcd * 1024) if per_xcd else None)
        kernels[k] = row
    steps = (traffic.get("k_sa3", {}).get("launches", 0)) / 3.0
    total = sum(r.get("hbm_bytes_all_launches", 0.0) for r in kernels.values())
    json.dump({"tag": tag, "source_sha256_16": source_hashes(),
               "command": "profiles/collect.sh: rocprofv3 --kernel-trace --pmc {FETCH_SIZE | WRITE_SIZE | SQ_* | GRBM_GUI_ACTIVE} (separate "
                          "passes) -- python bench.py --steps 1 --warmup 1 --cell-streams 1 ...",
               "kernels": kernels,
               "whole_run": {"hbm_bytes": total, "steps_executed": steps, "hbm_bytes_per_step": total / steps if steps else None,
                             "note": "every kernel of the run (the step's kernels, torch's glue, the calibration and phase-rate passes) / "
                                     "the full-size passes the run made (k_sa3 launches / 3)"}},
              open(out, "w"), indent=1)
    for k in ("k_sa3", "k_sa_rows<128, 128, 64, 4>", "k_ga2", "k_sa_points<12>"):
        if k in kernels:
            print(k, {a: (round(b, 4) if isinstance(b, float) and b < 100 else b) for a, b in kernels[k].items()})
    print("whole run:", total / 1e9, "GB over", steps, "steps")


if __name__ == "__main__":
    main()

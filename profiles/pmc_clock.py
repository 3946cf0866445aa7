"""Effective shader clock per kernel from a rocprofv3 `--kernel-trace --pmc GRBM_GUI_ACTIVE` pass (CSV output):
    python profiles/pmc_clock.py <dir> <prefix> [top_n]
GRBM_GUI_ACTIVE counts the cycles the graphics / compute engine is busy; over a dispatch that fills the chip it is (summed
over the XCDs the profiler reports) proportional to the dispatch's duration x its shader clock.  The ratio is normalised by
the value of a light, latency-bound kernel's best dispatch, so a kernel that runs at the full 2.4 GHz reads ~1.00 and one the
power cap holds at 1.9 GHz reads ~0.79 (MI355X_MICROARCH.md, "DVFS give-back")."""
import collections
import csv
import re
import sys


def short(k):
    m = re.search(r"(k_\w+(<[^>]*>)?)", k)
    return m.group(1) if m else k[:40]


def main():
    d, pre = sys.argv[1], sys.argv[2]
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    dur = {}
    for r in csv.DictReader(open(f"{d}/{pre}_kernel_trace.csv")):
        dur[r["Dispatch_Id"]] = (short(r["Kernel_Name"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    cyc, ns = collections.defaultdict(float), collections.defaultdict(float)
    best = 0.0
    for r in csv.DictReader(open(f"{d}/{pre}_counter_collection.csv")):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or r["Dispatch_Id"] not in dur:
            continue
        k, t = dur[r["Dispatch_Id"]]
        v = float(r["Counter_Value"])
        cyc[k] += v
        ns[k] += t
        if t > 200000:           # dispatches of at least 0.2 ms: the ratio of a short one is dominated by its ramp
            best = max(best, v / t)
    light = [k for k in ns if k.startswith("k_dedup_rows")]
    if light:      # a light streaming kernel of this library as the yardstick (tiny torch kernels read high: ramp effects)
        best = cyc[light[0]] / ns[light[0]]
    print(f"# GRBM_GUI_ACTIVE per ns of dispatch time; reference = {'k_dedup_rows (a light streaming kernel)' if light else 'largest ratio of a dispatch >= 0.2 ms'}: {best:.3f} (= 1.00 below)")
    for k in sorted(ns, key=lambda k: -ns[k])[:top]:
        r = cyc[k] / ns[k]
        print(f"{k:34s} ms={ns[k] / 1e6:8.2f}  cycles/ns={r:7.3f}  relative clock={r / best if best else 0:5.2f}")


if __name__ == "__main__":
    main()

"""Per-kernel summary of a rocprofv3 --pmc CSV run:  python profiles/pmc_summary.py <dir> <prefix> [top_n]"""
import collections
import csv
import re
import sys


def short(k):
    m = re.search(r"(k_\w+(<[^>]*>)?)", k)
    return m.group(1) if m else k[:40]


def main():
    d, pre = sys.argv[1], sys.argv[2]
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f"{d}/{pre}_counter_collection.csv")):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
    tr = collections.defaultdict(float)
    for r in csv.DictReader(open(f"{d}/{pre}_kernel_trace.csv")):
        tr[short(r["Kernel_Name"])] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    for k, v in sorted(agg.items(), key=lambda kv: -tr[kv[0]])[:top]:
        print(f"{k:34s} ms={tr[k]:8.2f} " + " ".join(f"{c}={x:.4g}" for c, x in sorted(v.items())))


if __name__ == "__main__":
    main()

"""Turns a rocprofv3 `--kernel-trace --stats` result into the short per-kernel table committed under profiles/.
usage: python profiles/summarize.py <results.db | *_kernel_trace.csv> <out.md> [title]

  * a rocpd sqlite database (ROCm 7.2 default output): the tool's own `top_kernels` view;
  * a `--output-format csv` kernel trace: one row per dispatch, so the table also carries the median and the largest
    launch.  bench.py launches every cell kernel a few times on small batches too (BatchNorm calibration, phase timings):
    those dilute `avg`; `full avg` is the mean over the launches of at least 3/4 of the largest duration, i.e. the
    full-size chunks that `roofline.avg_launch_ms` of the bench line averages (the PCIe-inclusive phase runs blocks of
    2,048 cells = half-size launches)."""
import collections
import csv
import re
import sqlite3
import statistics
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:<>, ]+?)\(", name)
    name = m.group(1) if m else name
    return name if len(name) < 90 else name[:87] + "..."


def from_db(db, f):
    rows = list(sqlite3.connect(db).execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    f.write("| kernel | calls | total us | avg us | % |\n|---|---:|---:|---:|---:|\n")
    for n, c, t, a, p in rows:
        if p < 0.01:
            continue
        f.write(f"| `{short(n)}` | {c} | {t:.0f} | {a:.1f} | {p:.2f} |\n")


def from_csv(path, f):
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    total = sum(sum(v) for v in dur.values())
    f.write("| kernel | calls | total us | avg us | median us | max us | full-size calls | full avg us | % |\n"
            "|---|---:|---:|---:|---:|---:|---:|---:|---:|\n")
    for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        p = 100.0 * sum(v) / total
        if p < 0.01:
            continue
        full = [x for x in v if x >= 0.75 * max(v)]
        f.write(f"| `{short(n)}` | {len(v)} | {sum(v):.0f} | {sum(v) / len(v):.1f} | {statistics.median(v):.1f} | {max(v):.1f} | "
                f"{len(full)} | {sum(full) / len(full):.1f} | {p:.2f} |\n")


def main():
    src, out = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else src
    with open(out, "w") as f:
        f.write(f"# {title}\n\n`rocprofv3 --kernel-trace --stats` (durations in microseconds)\n\n")
        (from_csv if src.endswith(".csv") else from_db)(src, f)
    print(open(out).read())


if __name__ == "__main__":
    main()

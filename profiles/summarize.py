"""Turns a rocprofv3 `--kernel-trace --stats` result database (rocpd sqlite, ROCm 7.2 default output) into the short
per-kernel table committed under profiles/.   usage: python profiles/summarize.py <results.db> <out.md> [title]"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:<>, ]+?)\(", name)
    name = m.group(1) if m else name
    return name if len(name) < 90 else name[:87] + "..."


def main():
    db, out = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else db
    rows = list(sqlite3.connect(db).execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out, "w") as f:
        f.write(f"# {title}\n\n`rocprofv3 --kernel-trace --stats` (durations in microseconds)\n\n")
        f.write("| kernel | calls | total us | avg us | % |\n|---|---:|---:|---:|---:|\n")
        for n, c, t, a, p in rows:
            if p < 0.01:
                continue
            f.write(f"| `{short(n)}` | {c} | {t:.0f} | {a:.1f} | {p:.2f} |\n")
    print(open(out).read())


if __name__ == "__main__":
    main()

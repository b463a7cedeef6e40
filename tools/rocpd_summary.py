#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (the default --kernel-trace --stats output of ROCm 7.2) into a
per-kernel table: calls, total / average / min / max duration.  Usage: rocpd_summary.py <results.db> [out.md]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for n, c, s, a, mn, mx in rows[:40]:
        lines.append(f"| `{n[:110]}` | {c} | {s / 1e6:.3f} | {a / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | "
                     f"{100.0 * s / total:.2f} |")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            f.write(out + "\n")


if __name__ == "__main__":
    main()

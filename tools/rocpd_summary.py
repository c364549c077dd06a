#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2 rocpd sqlite) kernel trace into a small CSV + markdown table.
usage: rocpd_summary.py <results.db> <out_prefix> [title]"""
import csv
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "")
    if name.startswith("at::native") or "at::native" in name:
        m = re.search(r"(\w+)_kernel", name)
        return "torch:" + (m.group(0) if m else name[:40])
    return name[:80]


def main():
    db, prefix = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else db
    con = sqlite3.connect(db)
    rows = con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                       "group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    agg = {}
    for name, n, tot, avg, mn, mx in rows:
        k = short(name)
        a = agg.setdefault(k, [0, 0.0, 1e30, 0.0])
        a[0] += n; a[1] += tot; a[2] = min(a[2], mn); a[3] = max(a[3], mx)
    items = sorted(agg.items(), key=lambda kv: -kv[1][1])
    with open(prefix + ".csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"])
        for k, (n, tot, mn, mx) in items:
            w.writerow([k, n, f"{tot/1e3:.1f}", f"{tot/n/1e3:.2f}", f"{mn/1e3:.2f}", f"{mx/1e3:.2f}", f"{100*tot/total:.2f}"])
    with open(prefix + ".md", "w") as f:
        f.write(f"# {title}\n\nrocprofv3 --kernel-trace --stats; durations in microseconds; {sum(a[0] for a in agg.values())} dispatches, "
                f"{total/1e6:.1f} ms of kernel time.\n\n| kernel | calls | total us | avg us | min us | max us | % |\n|---|---:|---:|---:|---:|---:|---:|\n")
        for k, (n, tot, mn, mx) in items[:30]:
            f.write(f"| `{k}` | {n} | {tot/1e3:.0f} | {tot/n/1e3:.1f} | {mn/1e3:.1f} | {mx/1e3:.1f} | {100*tot/total:.2f} |\n")
    print(open(prefix + ".md").read())


if __name__ == "__main__":
    main()

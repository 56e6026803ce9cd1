#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME` writes
DIR/NAME_results.db on ROCm 7.2) into the per-kernel table `--stats` prints as CSV on older versions:
name, calls, total, average, min, max duration and share.  Usage: rocpd_stats.py DB [TOP_N] > summary.txt"""
import sqlite3
import sys


def pmc(db):
    """Per kernel and counter: dispatches, mean and total counter value (counters_collection view)."""
    rows = db.execute("select kernel_name, counter_name, count(*), avg(value), sum(value), avg(duration) "
                      "from counters_collection group by kernel_name, counter_name order by 5 desc").fetchall()
    print(f"{'calls':>7} {'avg_value':>14} {'sum_value':>16} {'avg_us':>9}  counter      kernel")
    for name, ctr, n, avg, tot, dur in rows:
        short = name if len(name) <= 110 else name[:107] + "..."
        print(f"{n:7d} {avg:14.2f} {tot:16.1f} {dur / 1e3:9.2f}  {ctr:<12} {short}")


def main():
    db = sqlite3.connect(sys.argv[1])
    if len(sys.argv) > 2 and sys.argv[2] == "--pmc":
        return pmc(db)
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = db.execute(f"select {name_col}, count(*), sum(end - start), min(end - start), max(end - start) "
                      f"from kernels group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# kernels: {sum(r[1] for r in rows)} dispatches, {len(rows)} distinct, total GPU time {total / 1e6:.3f} ms")
    print(f"{'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'pct':>6}  name")
    for name, n, tot, mn, mx in rows[:top]:
        short = name if len(name) <= 150 else name[:147] + "..."
        print(f"{n:7d} {tot / 1e6:10.3f} {tot / n / 1e3:10.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {100.0 * tot / total:6.2f}  {short}")


if __name__ == "__main__":
    main()

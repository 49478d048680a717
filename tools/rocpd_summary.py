#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (kernel trace) into a per-kernel table: calls, total/avg/min/max ms, % of GPU time.
Usage: python tools/rocpd_summary.py <results.db> [skip_first_n_dispatches_fraction]"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    agg = {}
    for n, s, e in rows:
        n = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        a = agg.setdefault(n, [0, 0.0, 1e30, 0.0])
        d = (e - s) / 1e6
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_ms':>9s} {'min_ms':>9s} {'max_ms':>9s} {'pct':>6s}")
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{n[:70]:70s} {a[0]:7d} {a[1]:10.3f} {a[1]/a[0]:9.4f} {a[2]:9.4f} {a[3]:9.4f} {100*a[1]/tot:6.2f}")
    print(f"{'TOTAL':70s} {sum(a[0] for a in agg.values()):7d} {tot:10.3f}")


if __name__ == "__main__":
    main(sys.argv[1])

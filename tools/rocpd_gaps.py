#!/usr/bin/env python3
"""GPU idle time of a rocprofv3 kernel trace (rocpd SQLite): union of all kernel [start, end] intervals over the last `steps` steps of a bench run, the idle
gaps between them (no kernel of ANY stream resident), and which kernels precede the largest gaps.  A step boundary = the first launch of `marker` (default: the
first kernel of the speech tower).  Usage: python tools/rocpd_gaps.py <results.db> [marker_substring] [steps_to_use]"""
import sqlite3
import sys


def main(path, marker="conv0_stats", use=3):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    if len(marks) < use + 1:
        print("not enough steps", len(marks)); return
    lo, hi = marks[-use - 1], marks[-1]
    sel = rows[lo:hi]
    t0, t1 = sel[0][1], rows[hi][1]
    busy_end = t0
    idle = 0
    gaps = []
    for n, s, e in sel:
        if s > busy_end:
            idle += s - busy_end
            gaps.append((s - busy_end, prev, n))
        if e > busy_end:
            busy_end = e; prev = n
    wall = (t1 - t0) / 1e6 / use
    print(f"steps {use}: wall {wall:.3f} ms/step, kernels {len(sel) // use} per step, idle (no kernel resident) {idle / 1e6 / use:.3f} ms/step")
    agg = {}
    for g, a, b in gaps:
        k = (a.split("(")[0][-60:], b.split("(")[0][-60:])
        x = agg.setdefault(k, [0, 0]); x[0] += 1; x[1] += g
    for k, (cnt, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"  {tot / 1e6 / use:8.4f} ms/step in {cnt / use:6.1f} gaps/step   after {k[0]}  ->  before {k[1]}")


if __name__ == "__main__":
    main(sys.argv[1], *(sys.argv[2:3] or ["conv0_stats"]), *(int(x) for x in sys.argv[3:4]))

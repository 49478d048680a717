#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files (one counter per pass) per kernel: calls, total and per-call value.
Usage: python tools/pmc_summary.py <dir_or_csv> [...]   (FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB)"""
import collections
import csv
import glob
import os
import re
import sys


def clean(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.split(r"\(", n, 1)[0][:60]


def main(paths):
    for p in paths:
        f = p if p.endswith(".csv") else glob.glob(os.path.join(p, "*counter_collection.csv"))[0]
        agg, cnt = collections.defaultdict(float), collections.Counter()
        cname = None
        for r in csv.DictReader(open(f)):
            k = clean(r["Kernel_Name"])
            cname = r["Counter_Name"]
            agg[(k, cname)] += float(r["Counter_Value"])
            cnt[(k, cname)] += 1
        print(f"# {f}")
        print(f"{'kernel':62s} {'counter':12s} {'calls':>7s} {'total':>14s} {'per_call':>14s}")
        for key in sorted(agg, key=lambda k: -agg[k])[:25]:
            print(f"{key[0]:62s} {key[1]:12s} {cnt[key]:7d} {agg[key]:14.1f} {agg[key] / cnt[key]:14.1f}")


if __name__ == "__main__":
    main(sys.argv[1:])

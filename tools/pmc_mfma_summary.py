#!/usr/bin/env python3
"""Per-kernel MFMA-pipe utilisation from the two rocprofv3 --pmc passes of tools/pmc_mfma.sh (raw/: busy cycles, ops/: MFMA op counts).
util = SQ_VALU_MFMA_BUSY_CYCLES / (kernel duration x 2.4 GHz x 1024 SIMDs); the duration is the dispatch's own Start/End timestamp in the
counter CSV (GRBM_GUI_ACTIVE comes back summed over the 8 XCCs, so the stock MfmaUtil expression under-reports 8x on gfx950).
flops = SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512 (executed, padding included)."""
import collections, csv, glob, re, sys


def clean(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.split(r"\(", n, 1)[0][:66]


def load(d):
    per = collections.defaultdict(lambda: collections.defaultdict(float)); dur, name = {}, {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            i = r["Dispatch_Id"]; per[i][r["Counter_Name"]] += float(r["Counter_Value"])
            dur[i] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); name[i] = clean(r["Kernel_Name"])
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for i, c in per.items():
        a = agg[name[i]]; a["calls"] += 1; a["ns"] += dur[i]
        for k, v in c.items():
            a[k] += v
    return agg


def main(o):
    raw, ops = load(o + "/raw"), load(o + "/ops")
    print("MFMA-pipe utilisation per kernel: 3 steps (1 warm-up + 2 timed) of the default bench workload, rocprofv3 --pmc on gfx950, 256 CUs x 4 SIMDs")
    print("util = SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024);  TF/s = SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512 / duration (executed flops, of 2500 peak)")
    print(f"{'kernel':68s} {'calls':>5s} {'ms':>8s} {'mfma_busy_cyc':>14s} {'util':>6s} {'exec TF/s':>10s}")
    tb = tn = 0.0
    for k, a in sorted(raw.items(), key=lambda kv: -kv[1]["ns"])[:12]:
        mb = a["SQ_VALU_MFMA_BUSY_CYCLES"]; b = ops.get(k, {})
        tf = b.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0) * 512 / b["ns"] / 1e3 if b.get("ns") else 0.0
        print(f"{k:68s} {int(a['calls']):5d} {a['ns'] / 1e6:8.2f} {mb:14.4g} {100 * mb / (a['ns'] * 2.4 * 1024):5.1f}% {tf:10.0f}")
    for a in raw.values():
        tb += a["SQ_VALU_MFMA_BUSY_CYCLES"]; tn += a["ns"]
    print(f"all kernels of the step: {tn / 3e6:.2f} ms/step of kernel time, MFMA pipe busy {100 * tb / (tn * 2.4 * 1024):.1f}% of it")


if __name__ == "__main__":
    main(sys.argv[1])

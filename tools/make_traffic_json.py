#!/usr/bin/env python3
"""profiles/rNN_gemm_hbm_traffic.json from the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over `bench.py --steps S --warmup 0`:
bytes per step and per launch of the GEMM kernels, next to their algorithmic bytes.  FETCH_SIZE is doubled (gfx950 correction prescribed by
MI355X_MICROARCH.md, calibrated in round 1 on layernorm768_kernel: corrected read = its 196.6 MB input exactly); both counters are in KiB.
usage: python tools/make_traffic_json.py <pmc_FETCH_SIZE dir> <pmc_WRITE_SIZE dir> <steps> <source label> > profiles/r02_gemm_hbm_traffic.json"""
import collections, csv, glob, json, os, sys


def load(d):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        agg[k] += float(r["Counter_Value"]) * 1024.0
        cnt[k] += 1
    return agg, cnt


def lib_sha16():
    """sha256[:16] of the product library the passes ran on: bench.py compares it with the library of the run that quotes these numbers."""
    import hashlib
    p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "speechclip_amd", "libspeechclip_hip.so")
    return hashlib.sha256(open(p, "rb").read()).hexdigest()[:16] if os.path.exists(p) else None


def main():
    fd, wd, steps, label = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    fetch, fc = load(fd)
    write, _ = load(wd)

    def group(pred):
        ks = [k for k in fetch if pred(k)]
        return {"kernels": sorted(ks), "launches_per_step": sum(fc[k] for k in ks) // steps,
                "fetch_bytes_per_step": int(2 * sum(fetch[k] for k in ks) / steps), "write_bytes_per_step": int(sum(write.get(k, 0.0) for k in ks) / steps)}
    is_gemm = lambda k: k.startswith(("gemm256_kernel", "gemm_bf16_kernel"))              # noqa: E731
    is_vit = lambda k: k.startswith("gemm256_kernel") and k.split("<")[1].split(",")[2].strip() == "2"   # noqa: E731  QuickGELU variant: image tower only
    out = {"method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes over `python bench.py --steps %d --warmup 0 --cpu-pairs 0 "
                     "--no-roofline-events --no-vendor-comparator` (tools/collect_round_profiles.sh); FETCH_SIZE x2 as MI355X_MICROARCH.md prescribes for gfx950; "
                     "both counters in KiB" % steps,
           "source": label, "lib_sha16": lib_sha16()}
    for name, pred in (("gemm_all", is_gemm), ("gemm_main_stream", lambda k: is_gemm(k) and not is_vit(k))):
        g = group(pred)
        g["bytes_per_launch"] = int((g["fetch_bytes_per_step"] + g["write_bytes_per_step"]) / max(1, g["launches_per_step"]))
        out[name] = g
    # algorithmic bytes of the main-stream GEMMs at B = 256, 10 s audio (unique operand + output bytes, bf16): conv stack + transformer + heads
    B, T = 256, 500
    rows = [32000, 16000, 8000, 4000, 2000, 1000, 500]
    conv_r = sum(B * r * 512 * 2 for r in rows[:6])            # each conv layer reads the previous layer's output once
    conv_w = sum(B * r * 512 * 2 for r in rows[1:])
    M, d, f = B * T, 768, 3072
    layer_r = M * 2 * (d + d + d + d + f + d)                  # qkv in, out-proj in + residual, fc1 in, fc2 in + residual
    layer_w = M * 2 * (3 * d + d + f + d)
    out["gemm_main_stream"]["algorithmic_bytes_per_step"] = {"read": conv_r + 12 * layer_r + M * 512 * 2, "write": conv_w + 12 * layer_w + M * d * 2}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

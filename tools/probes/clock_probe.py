#!/usr/bin/env python3
"""What clock and power does the GPU run at under the GEMMs of the step?  Polls rocm-smi (sclk, socket power) while one GEMM shape runs in a
loop for ~3 s; prints idle / loaded readings per shape.  usage: python tools/probes/clock_probe.py [shape ...]"""
import os, subprocess, sys, threading, time, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from speechclip_amd import ops

SH = {"qkv": (128000, 2304, 768, 0), "out": (128000, 768, 768, 0), "fc1": (128000, 3072, 768, 1), "fc2": (128000, 768, 3072, 0),
      "conv1": (4096000, 512, 1536, 1), "sq8k": (8192, 8192, 8192, 0)}


def smi():
    try:
        o = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=20).stdout
    except Exception as e:
        return {"err": str(e)}
    d = {}
    m = re.search(r"sclk clock level:\s*\d+:?\s*\(?(\d+)Mhz", o)
    if m: d["sclk_MHz"] = int(m.group(1))
    m = re.search(r"mclk clock level:\s*\d+:?\s*\(?(\d+)Mhz", o)
    if m: d["mclk_MHz"] = int(m.group(1))
    m = re.search(r"Power \(W\):\s*([\d.]+)", o) or re.search(r"Socket Power.*?:\s*([\d.]+)", o)
    if m: d["power_W"] = float(m.group(1))
    m = re.search(r"junction\) \(C\):\s*([\d.]+)", o)
    if m: d["Tj_C"] = float(m.group(1))
    if not d: d["raw"] = o[-600:]
    return d


def main():
    print("idle:", smi(), flush=True)
    for name in sys.argv[1:] or list(SH):
        M, N, K, act = SH[name]
        lda = 1024 if name.startswith("conv") else K
        a = (torch.randn(M * lda + K + 64, device="cuda") * 0.5).to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
        bias = torch.randn(N, device="cuda")
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        stop, samples = [False], []

        def poll():
            while not stop[0]:
                samples.append(smi())
        th = threading.Thread(target=poll)
        ops.gemm(a, w, bias, act, out=out, M=M, K=K, lda=lda)
        torch.cuda.synchronize()
        th.start()
        t0 = time.time()
        n = 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        while time.time() - t0 < 4.0:
            for _ in range(20):
                ops.gemm(a, w, bias, act, out=out, M=M, K=K, lda=lda)
            n += 20
            torch.cuda.synchronize()
        e1.record(); torch.cuda.synchronize()
        stop[0] = True
        th.join()
        ms = e0.elapsed_time(e1) / n
        print(f"{name}: {2.0 * M * N * K / ms / 1e9:.0f} TF/s sustained over {n} launches; samples under load:", samples[1:-1] or samples, flush=True)
        del a, w, out


if __name__ == "__main__":
    main()

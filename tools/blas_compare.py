#!/usr/bin/env python3
"""Vendor-library reference point: torch.nn.functional.linear (hipBLASLt / rocBLAS underneath) on the step's GEMM shapes, bf16, vs sc_gemm_bf16."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from speechclip_amd import ops

SHAPES = [("qkv", 128000, 2304, 768, 0), ("out", 128000, 768, 768, 0), ("fc1+gelu", 128000, 3072, 768, 1), ("fc2", 128000, 768, 3072, 0),
          ("conv2-like", 2048000, 512, 1536, 1), ("vit_fc1", 12800, 3072, 768, 0), ("sq8k", 8192, 8192, 8192, 0)]


def timeit(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, M, N, K, act in SHAPES:
    a = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    b32 = torch.randn(N, device="cuda")
    b16 = b32.to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t_mine = timeit(lambda: ops.gemm(a, w, b32, act, out=out))
    if act:
        t_lib = timeit(lambda: F.gelu(F.linear(a, w, b16)))
    else:
        t_lib = timeit(lambda: F.linear(a, w, b16))
    fl = 2.0 * M * N * K
    print(f"{name:11s} M={M:8d} N={N:5d} K={K:5d}  sc_gemm {fl/t_mine/1e9:7.1f} TF/s   torch/hipBLASLt {fl/t_lib/1e9:7.1f} TF/s{' (+ separate GELU pass)' if act else ''}", flush=True)

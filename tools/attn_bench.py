#!/usr/bin/env python3
"""Attention microbenchmark at the P-base shape (B=256, T=500, H=12, d=64)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechclip_amd import ops
B, T, H = 256, 500, 12
qkv = torch.randn(B * T, 3 * H * 64, device="cuda").to(torch.bfloat16)
lens = torch.full((B,), 499, dtype=torch.int32, device="cuda")
out = torch.empty(B * T, H * 64, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    ops.attention(qkv, B, T, H, lens, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.attention(qkv, B, T, H, lens, out=out)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"attention B={B} T={T} H={H}: {ms:.3f} ms  {4.0*T*T*64*H*B/ms/1e9:.1f} TF/s")

#!/usr/bin/env python3
"""conv0 (GroupNorm + GELU form) at the P-base shape: B = 256 utterances of 160000 samples -> bf16 [256, 32000, 512] (8.4 GB written)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechclip_amd import ops
B, L, C = 256, 160000, 512
g = torch.Generator().manual_seed(0)
wav = (0.1 * torch.randn(B, L, generator=g)).cuda()
w = (torch.randn(C, 10, generator=g) * 0.3).cuda()
gamma, beta = torch.ones(C).cuda(), torch.zeros(C).cuda()
T0 = (L - 10) // 5 + 1
P = (T0 + 63) // 64 * 64
out = torch.zeros(B * P + 8, C, device="cuda", dtype=torch.bfloat16)
for _ in range(2):
    ops.conv0(wav, w, T0, P, gamma, beta, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    ops.conv0(wav, w, T0, P, gamma, beta, out=out)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"conv0 (stats + coef + fwd): {ms:.3f} ms  {B * P * C * 2 / ms / 1e9:.2f} TB/s written")
# (round 6) the same launch without GroupNorm + GELU (conv + bias only): how far the GELU / normalise vector work keeps the kernel from the write ceiling
bias = torch.zeros(C).cuda()
for _ in range(2):
    ops.conv0(wav, w, T0, P, bias=bias, out=out)
torch.cuda.synchronize()
e0.record()
for _ in range(5):
    ops.conv0(wav, w, T0, P, bias=bias, out=out)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"conv0 raw (W fragments + forward, bias, no GELU): {ms:.3f} ms  {B * P * C * 2 / ms / 1e9:.2f} TB/s written")

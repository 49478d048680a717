#!/usr/bin/env python3
"""In-step GEMM time by shape (HIP events around every sc_gemm_bf16* launch of the B=256 P-base step): where the dominant kernel's
milliseconds go when producers/consumers run around it (tools/gemm_bench.py times the shapes in isolation)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from speechclip_amd import ops, parallel

model = bench.build_model().cuda()
B, L = 256, 160000
g = torch.Generator().manual_seed(7122)
batch = {"wav": (0.1 * torch.randn(B, L, generator=g)).cuda(), "wav_len": torch.full((B,), L), "image": torch.randn(B, 3, 224, 224, generator=g).cuda(),
         "id": torch.arange(B).cuda()}
def step():
    with torch.no_grad():
        lf, _, _ = model(batch)
        return model.compute_loss(parallel.gather_loss_feats(lf))["loss"]
for _ in range(2):
    step()
torch.cuda.synchronize()
steps = 4
ops.PROFILE = []
for _ in range(steps):
    step()
torch.cuda.synchronize()
prof, ops.PROFILE = ops.PROFILE, None
agg = collections.OrderedDict()
for e0, e1, fl, tag, *_ in prof:
    a = agg.setdefault(tag, [0, 0.0, 0.0])
    a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += fl
tot = sum(a[1] for a in agg.values())
print(f"{'M':>8} {'N':>5} {'K':>5} act res f32 | calls/step  ms/step   TF/s   share")
for tag, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{tag[0]:8d} {tag[1]:5d} {tag[2]:5d} {tag[3]:3d} {int(tag[4]):3d} {int(tag[5]):3d} | {n/steps:6.1f} {ms/steps:9.3f} {fl/ms/1e9:7.1f} {100*ms/tot:6.1f}%")
print(f"total GEMM ms/step {tot/steps:.3f}")

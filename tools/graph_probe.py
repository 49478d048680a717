#!/usr/bin/env python3
"""Probe: capture the whole forward + loss step in a HIP graph (torch.cuda.CUDAGraph on ROCm = hipGraph) and compare eager vs replay."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from speechclip_amd import parallel
B = int(os.environ.get("B", "16"))
model = bench.build_model().cuda()
g = torch.Generator().manual_seed(1)
L = 160000
batch = {"wav": (0.1 * torch.randn(B, L, generator=g)).cuda(), "wav_len": torch.full((B,), L), "image": torch.randn(B, 3, 224, 224, generator=g).cuda(),
         "id": torch.arange(B).cuda()}
def step():
    with torch.no_grad():
        lf, _, _ = model(batch)
        return model.compute_loss(parallel.gather_loss_feats(lf))["loss"], lf["parallel_audio_feat"]
for _ in range(3): loss, pa = step()
torch.cuda.synchronize()
def timeit(fn, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
t_eager = timeit(step)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): step()
torch.cuda.current_stream().wait_stream(s)
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    gl, gpa = step()
gr.replay(); torch.cuda.synchronize()
print("loss eager", float(loss), "graph", float(gl), "max|dpa|", float((gpa - pa).abs().max()))
t_graph = timeit(gr.replay)
# new inputs through the static buffers
batch["wav"].copy_(0.1 * torch.randn(B, L, generator=g)); gr.replay(); l2 = float(gl); e2 = float(step()[0])
print(f"B={B}: eager {t_eager:.3f} ms/step, graph replay {t_graph:.3f} ms/step; new-input loss graph {l2:.5f} eager {e2:.5f}")

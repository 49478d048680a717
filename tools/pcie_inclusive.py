#!/usr/bin/env python3
"""PCIe-inclusive rate of the default workload: the batch starts in pinned host memory every step (fp32 waves 164 MB + fp32 images 154 MB),
(a) copied on the compute stream, (b) double-buffered on a copy stream so the transfer of step i+1 hides under step i."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, bench
from speechclip_amd import parallel
model = bench.build_model().cuda()
B, L = 256, 160000
g = torch.Generator().manual_seed(1)
host = {"wav": (0.1 * torch.randn(B, L, generator=g)).pin_memory(), "image": torch.randn(B, 3, 224, 224, generator=g).pin_memory()}
wav_len, ids = torch.full((B,), L), torch.arange(B).cuda()
def step(dev):
    with torch.no_grad():
        lf, _, _ = model({"wav": dev["wav"], "wav_len": wav_len, "image": dev["image"], "id": ids})
        return model.compute_loss(parallel.gather_loss_feats(lf))["loss"]
def upload(stream=None):
    with torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream()):
        return {k: v.to("cuda", non_blocking=True) for k, v in host.items()}
dev = upload(); torch.cuda.synchronize()
for _ in range(3): step(dev)
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for _ in range(N): step(dev)
torch.cuda.synchronize(); t_res = (time.perf_counter() - t0) / N
t0 = time.perf_counter()
for _ in range(N): step(upload())
torch.cuda.synchronize(); t_ser = (time.perf_counter() - t0) / N
copy = torch.cuda.Stream()
nxt = upload(copy); ev = torch.cuda.Event(); ev.record(copy)
t0 = time.perf_counter()
for _ in range(N):
    torch.cuda.current_stream().wait_event(ev)
    cur = nxt
    for v in cur.values(): v.record_stream(torch.cuda.current_stream())
    nxt = upload(copy); ev = torch.cuda.Event(); ev.record(copy)
    step(cur)
torch.cuda.synchronize(); t_ovl = (time.perf_counter() - t0) / N
print(f"resident inputs {B / t_res:.0f} pairs/s ({t_res * 1e3:.2f} ms) | H2D on the compute stream {B / t_ser:.0f} pairs/s ({t_ser * 1e3:.2f} ms) | "
      f"H2D double-buffered on a copy stream {B / t_ovl:.0f} pairs/s ({t_ovl * 1e3:.2f} ms); 318 MB per step")

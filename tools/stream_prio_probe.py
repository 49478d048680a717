#!/usr/bin/env python3
"""Does stream PRIORITY help the two-tower overlap?  The image tower runs on a side stream of default priority; here the step (speech tower, head, loss) is issued
on a HIGH-priority stream, so the side stream's kernels should only take what the speech tower leaves.  (Round 2 tried the opposite: a high-priority SIDE stream: nothing.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, bench
from speechclip_amd import parallel
model = bench.build_model().cuda()
batch, _ = bench.make_batch(256, 160000, 0, "cuda")
def step():
    with torch.no_grad():
        lf, _, _ = model(batch)
        return model.compute_loss(parallel.gather_loss_feats(lf))["loss"]
lo, hi_range = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print("stream priority range (least, greatest):", lo, hi_range)
hi = torch.cuda.Stream(priority=-1)
def timeit(ctx, n=20):
    with ctx():
        for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with ctx():
        for _ in range(n): l = step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, float(l)
import contextlib
for rep in range(3):
    a = timeit(contextlib.nullcontext)
    b = timeit(lambda: torch.cuda.stream(hi))
    print(f"pass {rep}: default stream {a[0]:.3f} ms (loss {a[1]:.5f})   step on a high-priority stream {b[0]:.3f} ms (loss {b[1]:.5f})")

#!/usr/bin/env python3
"""Which torch (aten) device ops still run inside one forward + loss step, and from which source line?  (FillFunctor /
copyBuffer / bfloat16_copy launches on a path documented as 'every tensor op in the .so'.)  torch.profiler with Python stacks over ONE warm step."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from speechclip_amd import parallel
from torch.profiler import profile, ProfilerActivity
kind = sys.argv[1] if len(sys.argv) > 1 else "base"
model = bench.build_model(large=kind == "large", cascaded=kind == "cascaded").cuda()
B = int(os.environ.get("B", "256" if kind != "large" else "64"))
batch, lens = bench.make_batch(B, 160000, 0, "cuda", varlen=os.environ.get("VARLEN") == "1")
def step():
    with torch.no_grad():
        lf, _, _ = model(batch)
        return model.compute_loss(parallel.gather_loss_feats(lf))["loss"]
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_time_total <= 0 or not ev.name.startswith("aten::"): continue
    if any(c.name.startswith("aten::") for c in (ev.cpu_children or [])): continue        # leaf aten ops only
    where = "?"
    for fr in (ev.stack or []):
        if "/speechclip_amd/" in fr or "/bench.py" in fr:
            where = fr.split("/root/repo/")[-1].split("/repo/")[-1]; break
    k = (ev.name, where)
    agg[k][0] += 1; agg[k][1] += ev.device_time_total
tot = sum(v[1] for v in agg.values()); n = sum(v[0] for v in agg.values())
print(f"{kind}: {n} aten device ops in one step, {tot:.1f} us of device time")
for (name, where), (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{c:4d} x {name:28s} {us:8.1f} us   {where}")

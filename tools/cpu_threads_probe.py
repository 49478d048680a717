import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from oracle.hubert_ref import HubertModelRef, HubertRefConfig, hubert_forward
print("cpu_count", os.cpu_count(), "torch threads default", torch.get_num_threads())
os.system("lscpu | grep -E 'Model name|Socket|Core|Thread|NUMA node\\(s\\)' | head -6")
m = HubertModelRef(HubertRefConfig.base()).eval()
x = torch.randn(2, 160000) * 0.1
mask = torch.zeros(2, 160000, dtype=torch.bool)
for n in (16, 32, 64, 128):
    torch.set_num_threads(n)
    hubert_forward(m, x[:1], mask[:1])
    t0 = time.perf_counter(); hubert_forward(m, x, mask); dt = time.perf_counter() - t0
    print(n, "threads:", round(2 / dt, 3), "utt/s (HuBERT only)", flush=True)

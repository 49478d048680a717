import torch, time
n = 8552 * 1000 * 1000 // 2
x = torch.empty(n, dtype=torch.bfloat16, device="cuda")
y = torch.empty(n // 2, dtype=torch.bfloat16, device="cuda")
def t(f, k=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
ms = t(lambda: x.zero_()); print(f"zero_ 8.55 GB (write only): {ms:.3f} ms = {x.numel()*2/ms/1e6:.0f} GB/s")
ms = t(lambda: x.fill_(1.5)); print(f"fill_ 8.55 GB (write only): {ms:.3f} ms = {x.numel()*2/ms/1e6:.0f} GB/s")
ms = t(lambda: y.copy_(x[:n//2])); print(f"copy 4.28 -> 4.28 GB: {ms:.3f} ms = {x.numel()*2/ms/1e6:.0f} GB/s (read + write)")

#!/usr/bin/env python3
"""Microbenchmark of the CLS-row head's Linear layers (M = B rows): hi/lo-split MFMA GEMM (hp_linear) vs the fp32 SIMT sgemm vs the bf16 GEMM."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from speechclip_amd import ops
from speechclip_amd.module.kw_modules.TransformerModels import _w3, hp_linear
B = int(os.environ.get("B", "256"))
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, N, K in (("out_proj", 768, 768), ("linear1", 3072, 768), ("linear2", 768, 3072), ("proj", 512, 768)):
    a = torch.randn(B, K, device="cuda")
    w = torch.randn(N, K, device="cuda") * K ** -0.5
    b = torch.randn(N, device="cuda")
    w3, w16 = _w3(w), w.to(torch.bfloat16)
    a16 = a.to(torch.bfloat16)
    lin = torch.nn.Linear(K, N).cuda()
    with torch.no_grad():
        lin.weight.copy_(w); lin.bias.copy_(b)
    t_hp = t(lambda: hp_linear(a, lin.weight, lin.bias))          # what the head runs (deep K: deterministic split-K)
    t_split = t(lambda: ops.split_hilo(a, 3))
    t_sg = t(lambda: ops.sgemm(a, w, transb=True, bias=b))
    t_bf = t(lambda: ops.gemm(a16, w16, b, out_f32=True))
    ref = a.double() @ w.double().t() + b.double()
    e_hp = ((hp_linear(a, lin.weight, lin.bias).double() - ref).norm() / ref.norm()).item()
    e_sg = ((ops.sgemm(a, w, transb=True, bias=b).double() - ref).norm() / ref.norm()).item()
    e_bf = ((ops.gemm(a16, w16, b, out_f32=True).double() - ref).norm() / ref.norm()).item()
    print(f"{name:9s} M={B} N={N} K={K}: hp_linear {t_hp:7.1f} us (split alone {t_split:5.1f}) rel err {e_hp:.1e} | sgemm {t_sg:7.1f} us {e_sg:.1e} | bf16 gemm {t_bf:6.1f} us {e_bf:.1e}")

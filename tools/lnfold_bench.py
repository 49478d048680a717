#!/usr/bin/env python3
"""Cost of the folded-LayerNorm epilogues per GEMM shape of the P-base step (B = 256): plain sc_gemm_bf16 vs sc_gemm_bf16_ln (mode 1 / 2),
the statistics finalisation and the stand-alone LayerNorm they replace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechclip_amd import ops

BF = torch.bfloat16
M = 128000


def t(fn, reps=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = "cuda"
    stats = torch.stack([torch.zeros(M), torch.ones(M)], 1).contiguous().to(dev)
    for name, N, K, act, res in (("qkv", 2304, 768, 0, False), ("fc1", 3072, 768, 1, False), ("out", 768, 768, 0, True), ("fc2", 768, 3072, 0, True)):
        a = (0.5 * torch.randn(M, K, device=dev)).to(BF)
        w = (torch.randn(N, K, device=dev) * K ** -0.5).to(BF)
        b = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev, dtype=BF)
        r = torch.randn(M, N, device=dev).to(BF) if res else None
        plain = t(lambda: ops.gemm(a, w, b, act, r, out=out))
        if not res:
            c = torch.randn(N, device=dev)
            fused = t(lambda: ops.gemm_ln(a, w, b, 1, act, out=out, ln_stats=stats, ln_c=c))
        else:
            g, be = torch.ones(N, device=dev), torch.zeros(N, device=dev)
            part = torch.empty(M, N // 64, 2, device=dev)
            fused = t(lambda: ops.gemm_ln(a, w, b, 2, residual=r, out=out, res_stats=stats, res_gamma=g, res_beta=be, ln_partial=part))
        print(f"{name:5s} plain {plain:7.3f} ms   folded {fused:7.3f} ms   (+{(fused - plain) * 1e3:6.1f} us, {100 * (fused / plain - 1):5.1f} %)", flush=True)
    y = torch.randn(M, 768, device=dev).to(BF)
    g, be = torch.ones(768, device=dev), torch.zeros(768, device=dev)
    o = torch.empty_like(y)
    print(f"layernorm768 {t(lambda: ops.layernorm(y, g, be, out=o)) * 1e3:7.1f} us", flush=True)
    part = torch.randn(M, 12, 2, device=dev)
    st = torch.empty(M, 2, device=dev)
    print(f"ln_stats_finalize {t(lambda: ops.ln_stats_finalize(part, 768, out=st)) * 1e3:7.1f} us", flush=True)
    hid = torch.randn(13, M, 768, device=dev).to(BF)
    wts = torch.randn(13, device=dev)
    print(f"weighted_sum {t(lambda: ops.weighted_sum(hid, wts)) * 1e3:7.1f} us", flush=True)
    gam, bet = torch.ones(12, 768, device=dev), torch.zeros(12, 768, device=dev)
    print(f"weighted_sum_ln {t(lambda: ops.weighted_sum_ln(hid[0], hid[1:], gam, bet, wts)) * 1e3:7.1f} us", flush=True)


if __name__ == "__main__":
    main()

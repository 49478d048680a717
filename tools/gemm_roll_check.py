#!/usr/bin/env python3
"""Rolling-epilogue GEMM (gemm256r_kernel, -DSC_GEMM_ROLL=1 builds) against an fp32 reference: eligible shapes of the step, odd panel counts per block,
the conv-as-GEMM overlapping-row form with tap pairing, GELU / QuickGELU epilogues, run-to-run bitwise determinism."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechclip_amd import ops, _lib
torch.manual_seed(0)
CASES = [(2560, 512, 512, 0, None), (25600, 768, 768, 0, None), (12800, 2304, 768, 0, None), (128000, 2304, 768, 0, None), (128000, 3072, 768, 1, None),
         (33024, 768, 3072, 0, None), (12800, 3072, 768, 2, None), (65536, 512, 1536, 1, 1024), (4096000 // 8, 512, 1536, 1, 1024), (7680, 256, 1024, 0, None)]
bad = 0
for M, N, K, act, lda in CASES:
    ld = lda or K
    a = (torch.randn(M * ld + K + 64, device="cuda") * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ops.gemm(a, w, bias, act, out=out, M=M, K=K, lda=ld)
    out2 = torch.empty_like(out)
    ops.gemm(a, w, bias, act, out=out2, M=M, K=K, lda=ld)
    A = torch.as_strided(a, (M, K), (ld, 1)).float()
    rows = torch.randint(0, M, (4096,), device="cuda")
    rows[:512] = torch.arange(512, device="cuda"); rows[512:1024] = torch.arange(M - 512, M, device="cuda")
    ref = A[rows] @ w.float().t() + bias
    if act == 1: ref = torch.nn.functional.gelu(ref)
    if act == 2: ref = ref * torch.sigmoid(1.702 * ref)
    got = out[rows].float()
    err = (got - ref).abs().max().item(); scale = ref.abs().max().item()
    ok = err <= 2e-2 * scale + 2e-2 and torch.equal(out, out2)
    bad += not ok
    print(f"M={M} N={N} K={K} act={act} lda={ld}: max err {err:.4f} (scale {scale:.2f}) deterministic={torch.equal(out, out2)} {'ok' if ok else 'FAIL'}")
sys.exit(1 if bad else 0)

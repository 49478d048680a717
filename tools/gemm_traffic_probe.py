#!/usr/bin/env python3
"""HBM-side fetch of single GEMM launches vs their algorithmic operand bytes (run under `rocprofv3 --pmc FETCH_SIZE`; tools/pmc_gemm_traffic.sh parses it).
Shapes: name:M,N,K,lda,act ...   Each shape: 2 warm-up + 3 measured launches (the parser takes the last 3 dispatches of each shape)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechclip_amd import ops
from speechclip_amd._lib import lib
if os.environ.get('SC_GEMM_KERNEL_MODE'):
    lib().sc_debug_set_gemm_mode(int(os.environ['SC_GEMM_KERNEL_MODE']))
for spec in sys.argv[1:]:
    name, rest = spec.split(":")
    M, N, K, lda, act = (int(v) for v in rest.split(","))
    a = (torch.randn(M * lda + K + 64, device="cuda") * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(5):
        ops.gemm(a, w, bias, act, out=out, M=M, K=K, lda=lda)
    torch.cuda.synchronize()
    print(name, M, N, K, lda, "algorithmic_read_bytes", (M * lda + (K - lda)) * 2 + N * K * 2, "write_bytes", M * N * 2, flush=True)
    del a, w, out

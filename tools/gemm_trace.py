#!/usr/bin/env python3
"""Per-phase cycle breakdown of the 256x256 GEMM kernel (debug hook sc_debug_set_gemm_trace): wait / main loop / next-tile prefetch issue / epilogue."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechclip_amd import ops, _lib
SH = {"qkv": (128000, 2304, 768, 0), "fc1": (128000, 3072, 768, 1), "fc2": (128000, 768, 3072, 0), "conv1": (4096000, 512, 1536, 1), "sq8k": (8192, 8192, 8192, 0),
      "out": (128000, 768, 768, 0), "n1536": (128000, 1536, 768, 0), "out6": (131072, 768, 768, 0), "out5": (109056, 768, 768, 0)}
L = _lib.lib()
L.sc_debug_set_gemm_trace.argtypes = [ctypes.c_void_p]
for name in sys.argv[1:] or list(SH):
    M, N, K, act = SH[name]
    lda = 1024 if name.startswith("conv") else K
    a = (torch.randn(M * lda + K + 64, device="cuda") * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ops.gemm(a, w, bias, act, out=out, M=M, K=K, lda=lda)
    tr = torch.zeros(256, 40, dtype=torch.int64, device="cuda")
    L.sc_debug_set_gemm_trace(tr.data_ptr())
    ops.gemm(a, w, bias, act, out=out, M=M, K=K, lda=lda)
    torch.cuda.synchronize()
    L.sc_debug_set_gemm_trace(None)
    t = tr.double().cpu()
    tiles = t[:, 4].clamp(min=1)
    per = t[:, :4] / tiles[:, None]
    m = per.mean(0)
    nk = K // 64
    pw = t[:, 8:40].reshape(256, 8, 4) / tiles[:, None, None]      # per wave: wait / loop / set-up / epilogue
    pwm = pw.mean(0)
    print("   per wave  wait:", [int(v) for v in pwm[:, 0]], " epilogue:", [int(v) for v in pwm[:, 3]], " loop:", [int(v) for v in pwm[:, 1]])
    xcc = tr[:, 5].cpu().tolist()
    print("   XCC id of blocks 0..15:", [int(v) & 15 for v in xcc[:16]], " blocks with xcc == bid%8:", sum(int(v) & 15 == (i % 8) for i, v in enumerate(xcc)), "/ 256")
    print(f"{name:6s} tiles/block={tiles.mean():.2f} per-tile cycles(100MHz ticks?): wait={m[0]:.0f} loop={m[1]:.0f} ({m[1]/nk:.1f}/kstep) prefetch-issue={m[2]:.0f} epilogue={m[3]:.0f}  total={m.sum():.0f}")

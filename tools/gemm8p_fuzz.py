#!/usr/bin/env python3
"""Random-shape sweep of gemm8p_pers_kernel (forced: sc_debug_set_gemm_mode(16)) against fp32 torch: small / odd tile counts (1 tile, fewer tiles than
CUs, one XCD short), ragged M, K = 128 .. 4096, N = 256 .. 8192, overlapping rows, every epilogue variant, column bands.  usage: gemm8p_fuzz.py [n] [seed]"""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechclip_amd import ops
from speechclip_amd._lib import lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
bad = 0
for it in range(n):
    M = rng.choice([256, 257, 300, 511, 512, 1000, 2048, 4097, 9000, 20011, 65536 + 3, 131072])
    N = 256 * rng.choice([1, 1, 2, 3, 4, 5, 8, 9, 12, 16, 17, 32])
    K = 64 * rng.choice([2, 2, 3, 4, 5, 8, 12, 16, 24, 48, 64])
    if M * N > 3e8 or M * K > 3e8:
        M = 2048
    act = rng.choice([0, 0, 1, 2]); res = rng.random() < 0.4; f32 = rng.random() < 0.3
    overlap = rng.random() < 0.25 and K >= 256
    ld = (K // 2 + 64) // 8 * 8 if overlap else K + rng.choice([0, 0, 8, 64])
    mode = rng.choice([16, 16, 16, 21, 22, 23, 19, 24])
    g = torch.Generator(device="cpu").manual_seed(it)
    flat = (torch.randn(M * ld + K + 8, generator=g) * 0.5).to("cuda", torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to("cuda", torch.bfloat16)
    bias = torch.randn(N, generator=g).cuda() if rng.random() < 0.8 else None
    r = torch.randn(M, N, generator=g).to("cuda", torch.float32 if f32 else torch.bfloat16) if res else None
    lib().sc_debug_set_gemm_mode(mode)
    y = ops.gemm(flat, w, bias, act, r, out_f32=f32, M=M, K=K, lda=ld)
    path = lib().sc_gemm_last_path()
    torch.cuda.synchronize()
    a = torch.as_strided(flat, (M, K), (ld, 1))
    ref = a.float() @ w.float().t()
    if bias is not None: ref = ref + bias
    if act == 1: ref = torch.nn.functional.gelu(ref)
    elif act == 2: ref = ref * torch.sigmoid(1.702 * ref)
    if r is not None: ref = ref + r.float()
    d = (y.float() - ref).abs()
    tol = (3e-3 if f32 else 2e-2) * (1 + ref.abs())
    nbad = int((d > tol).sum())
    ok = nbad == 0 and path == 3
    bad += not ok
    print(f"{it:3d} M={M:6d} N={N:5d} K={K:5d} ld={ld:5d} act={act} res={int(res)} f32={int(f32)} mode={mode} path={path} maxerr={float(d.max()):.4f} {'ok' if ok else 'FAIL (%d)' % nbad}", flush=True)
lib().sc_debug_set_gemm_mode(-1)
print("FUZZ", "OK" if bad == 0 else f"FAILED ({bad})")
sys.exit(1 if bad else 0)

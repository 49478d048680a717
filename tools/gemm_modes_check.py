#!/usr/bin/env python3
"""The GEMM kernels behind sc_gemm_bf16 side by side (sc_debug_set_gemm_mode): correctness against fp32 torch on a spread of shapes, then sustained-clock timing on the step's shapes.
usage: duet_check.py [check] [time] [--sustain S] [--modes 0,4,8]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechclip_amd import ops
from speechclip_amd._lib import lib


def ref(a, w, bias, act, res):
    y = a.float() @ w.float().t()
    if bias is not None:
        y = y + bias
    if act == 1:
        y = torch.nn.functional.gelu(y)
    elif act == 2:
        y = y * torch.sigmoid(1.702 * y)
    if res is not None:
        y = y + res.float()
    return y


CHECK_MODES = [0, 4, 8]


def check():
    CHECK_MODES[:] = sorted(set([0] + MODES))
    ok = True
    cases = [  # M, N, K, lda, act, res
        (128 * 200, 768, 768, None, 0, False), (128 * 200 + 37, 768, 768, None, 0, True), (50000, 2304, 768, None, 0, False),
        (40000, 3072, 768, None, 1, False), (30011, 768, 3072, None, 0, True), (300000, 512, 1536, 1024, 1, False),
        (70000, 512, 1024, 1024, 1, False), (60000, 256, 512, None, 2, False), (45000, 768, 512, None, 0, False),
        (128 * 66 * 6, 768, 768, None, 2, True), (128000, 2304, 768, None, 0, False), (256 * 300, 768, 3072, None, 0, True), (256 * 999, 512, 1536, 1024, 1, False),
        (256 * 100, 3072, 768, None, 1, False), (8192, 8192, 8192, None, 0, False),
    ]
    for M, N, K, lda, act, use_res in cases:
        g = torch.Generator(device="cpu").manual_seed(M + N + K)
        ld = lda or K
        flat = (torch.randn(M * ld + K + 8, generator=g) * 0.5).to("cuda", torch.bfloat16)
        w = (torch.randn(N, K, generator=g) * K ** -0.5).to("cuda", torch.bfloat16)
        bias = torch.randn(N, generator=g).cuda()
        res = torch.randn(M, N, generator=g).to("cuda", torch.bfloat16) if use_res else None
        a = torch.as_strided(flat, (M, K), (ld, 1))
        r = None
        # reference in row chunks (fp32 of the big shapes does not fit comfortably otherwise)
        outs = {}
        for mode in CHECK_MODES:
            lib().sc_debug_set_gemm_mode(mode)
            y = ops.gemm(flat, w, bias, act, res, M=M, K=K, lda=ld)
            path = lib().sc_gemm_last_path()
            torch.cuda.synchronize()
            outs[mode] = (y, path)
        worst = {}
        for m0 in range(0, M, 65536):
            m1 = min(M, m0 + 65536)
            rr = ref(a[m0:m1], w, bias, act, res[m0:m1] if res is not None else None)
            for mode, (y, path) in outs.items():
                d = (y[m0:m1].float() - rr).abs()
                tol = 2e-2 + 2e-2 * rr.abs()
                bad = int((d > tol).sum())
                e = worst.setdefault(mode, [0.0, 0])
                e[0] = max(e[0], float(d.max())); e[1] += bad
        line = f"M={M} N={N} K={K} lda={ld} act={act} res={use_res}: "
        for mode, (y, path) in outs.items():
            line += f" mode{mode}(path {path}) maxerr {worst[mode][0]:.4f} bad {worst[mode][1]};"
            if worst[mode][1]:
                ok = False
        same = all(torch.equal(outs[0][0], outs[m][0]) for m in CHECK_MODES[1:])
        print(line, "bitwise-equal-to-256-tile" if same else "", flush=True)
    lib().sc_debug_set_gemm_mode(-1)
    print("CHECK", "OK" if ok else "FAILED", flush=True)
    return ok


SHAPES = [  # name, M, N, K, lda, act, res
    ("qkv", 128000, 2304, 768, None, 0, False), ("out_res", 128000, 768, 768, None, 0, True), ("fc1", 128000, 3072, 768, None, 1, False),
    ("fc2_res", 128000, 768, 3072, None, 0, True), ("out", 128000, 768, 768, None, 0, False),
    ("conv1", 4096000, 512, 1536, 1024, 1, False), ("conv2", 2048000, 512, 1536, 1024, 1, False), ("conv5", 256000, 512, 1024, 1024, 1, False),
    ("proj", 128000, 768, 512, None, 0, False), ("sq8k", 8192, 8192, 8192, None, 0, False), ("fc2", 128000, 768, 3072, None, 0, False),
    # P-large at 64 pairs (pre-LN: fp32 residual stream) and the ViT-B/32 residual GEMMs: "f32" = fp32 output + fp32 residual
    ("out_l_f32", 31936, 1024, 1024, None, 0, "f32"), ("fc2_l_f32", 31936, 1024, 4096, None, 0, "f32"), ("fc1_l", 31936, 4096, 1024, None, 1, False),
    ("qkv_l", 31936, 3072, 1024, None, 0, False), ("vit_fc2_f32", 12800, 768, 3072, None, 0, "f32"), ("vit_out_f32", 12800, 768, 768, None, 0, "f32"),
]


def timeit(sustain, modes, only):
    res = {}
    for name, M, N, K, lda, act, use_res in SHAPES:
        if only and name not in only:
            continue
        ld = lda or K
        a = (torch.randn(M * ld + K + 64, device="cuda") * 0.5).to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
        bias = torch.randn(N, device="cuda")
        f32 = use_res == "f32"
        resid = torch.randn(M, N, device="cuda").to(torch.float32 if f32 else torch.bfloat16) if use_res else None
        out = torch.empty(M, N, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)

        def run(n):
            for _ in range(n):
                ops.gemm(a, w, bias, act, resid, out=out, out_f32=f32, M=M, K=K, lda=ld)
        run(3)
        torch.cuda.synchronize()
        t0 = time.time()
        while time.time() - t0 < sustain:
            run(20)
            torch.cuda.synchronize()
        row = {}
        for rnd in range(3):          # interleaved rounds
            for mode in modes:
                ops.set_vendor_gemm(mode == 99)            # mode 99: the hipBLASLt comparator (plain shapes only)
                lib().sc_debug_set_gemm_mode(mode if mode != 99 else -1)
                run(5)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(30); e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 30
                row.setdefault(mode, []).append(round(2.0 * M * N * K / (ms * 1e-3) / 1e12, 1))
        res[name] = row
        print(f"{name:8s} " + "  ".join(f"mode{m}: {v}" for m, v in row.items()), flush=True)
        del a, w, out, resid
    lib().sc_debug_set_gemm_mode(-1)
    ops.set_vendor_gemm(False)
    print(json.dumps(res))


def trace8p(only):
    """PROBES library: per-tile phases of gemm8p_kernel (mode 17): prologue / k-loop / epilogue cycles per wave, block lifetime and gaps."""
    import ctypes
    L = lib()
    L.sc_debug_set_gemm_trace.argtypes = [ctypes.c_void_p]
    for name, M, N, K, lda, act, use_res in SHAPES:
        if only and name not in only:
            continue
        ld = lda or K
        a = (torch.randn(M * ld + K + 64, device="cuda") * 0.5).to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
        bias = torch.randn(N, device="cuda")
        resid = torch.randn(M, N, device="cuda").to(torch.bfloat16) if use_res else None
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        L.sc_debug_set_gemm_mode(17)
        for _ in range(10):
            ops.gemm(a, w, bias, act, resid, out=out, M=M, K=K, lda=ld)
        tr = torch.zeros(4096 * 64, dtype=torch.int64, device="cuda")
        L.sc_debug_set_gemm_trace(tr.data_ptr())
        ops.gemm(a, w, bias, act, resid, out=out, M=M, K=K, lda=ld)
        torch.cuda.synchronize()
        L.sc_debug_set_gemm_trace(None)
        ntiles = min(4096, (M // 256) * (N // 256))
        t = tr.reshape(4096, 8, 8)[:ntiles].double()
        nk = K // 64
        pro, loop, epi = t[:, :, 0].mean(), t[:, :, 1].mean(), t[:, :, 2].mean()
        life = (t[:, :, 4].max(dim=1).values - t[:, :, 3].min(dim=1).values).mean()
        span = t[:, :, 4].max() - t[:, :, 3][t[:, :, 3] > 0].min()
        print(f"{name}: per tile: prologue {pro:7.0f}  loop {loop:7.0f} ({loop/nk:6.0f} per k-step)  epilogue+drain {epi:7.0f}  block lifetime {life:7.0f}; "
              f"{ntiles} traced tiles span {span:.0f} cycles = {span/ (ntiles/256):.0f} per round of 256", flush=True)
        L.sc_debug_set_gemm_mode(-1)
        del a, w, out, resid


def trace8pp(only):
    """PROBES library: gemm8p_pers_kernel (mode 16): per tile, cycles in the first k-step (incl. waiting for the partner's epilogue), the others, the epilogue."""
    import ctypes
    L = lib()
    L.sc_debug_set_gemm_trace.argtypes = [ctypes.c_void_p]
    for name, M, N, K, lda, act, use_res in SHAPES:
        if only and name not in only:
            continue
        ld = lda or K
        a = (torch.randn(M * ld + K + 64, device="cuda") * 0.5).to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
        bias = torch.randn(N, device="cuda")
        resid = torch.randn(M, N, device="cuda").to(torch.bfloat16) if use_res else None
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        L.sc_debug_set_gemm_mode(16)
        for _ in range(10):
            ops.gemm(a, w, bias, act, resid, out=out, M=M, K=K, lda=ld)
        tr = torch.zeros(256 * 64, dtype=torch.int64, device="cuda")
        L.sc_debug_set_gemm_trace(tr.data_ptr())
        ops.gemm(a, w, bias, act, resid, out=out, M=M, K=K, lda=ld)
        torch.cuda.synchronize()
        L.sc_debug_set_gemm_trace(None)
        t = tr.reshape(256, 8, 8).double()
        nk = K // 64
        for grp in (0, 1):
            x = t[:, 4 * grp:4 * grp + 4, :]
            tiles = x[:, :, 4].mean()
            print(f"{name} g{grp}: per tile: first k-step {x[:, :, 0].mean()/tiles:7.0f}  other k-steps {x[:, :, 1].mean()/tiles/(nk-1):6.0f} each  epilogue {x[:, :, 2].mean()/tiles:7.0f}"
                  f"  | tile {x[:, :, 3].mean()/tiles:7.0f}  tiles/block {tiles:.2f}  lifetime min/max {x[:, :, 3].min():.0f}/{x[:, :, 3].max():.0f}", flush=True)
        # dynamic tile order: tiles taken and lifetime per XCD (block b lives on XCD b % 8) -- how uneven the XCDs run
        tiles_b, life_b = t[:, 0, 4], t[:, 0, 3]
        print(f"{name} per XCD: tiles " + " ".join(f"{int(tiles_b[x::8].sum())}" for x in range(8)) + "  | lifetime max (k cycles) "
              + " ".join(f"{life_b[x::8].max()/1e3:.0f}" for x in range(8)) + f"  | tiles per block min/max {int(tiles_b.min())}/{int(tiles_b.max())}", flush=True)
        L.sc_debug_set_gemm_mode(-1)
        del a, w, out, resid


def trace(only, modes):
    """PROBES library only (SPEECHCLIP_HIP_LIB=speechclip_amd/libspeechclip_hip_probes.so): per-wave cycles spent in solo / joint / epilogue / null steps."""
    import ctypes
    L = lib()
    L.sc_debug_set_gemm_trace.argtypes = [ctypes.c_void_p]
    for name, M, N, K, lda, act, use_res in SHAPES:
        if only and name not in only:
            continue
        ld = lda or K
        a = (torch.randn(M * ld + K + 64, device="cuda") * 0.5).to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
        bias = torch.randn(N, device="cuda")
        resid = torch.randn(M, N, device="cuda").to(torch.bfloat16) if use_res else None
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        for mode in modes:
            if mode == 0:
                continue
            L.sc_debug_set_gemm_mode(mode)
            for _ in range(10):
                ops.gemm(a, w, bias, act, resid, out=out, M=M, K=K, lda=ld)
            tr = torch.zeros(256 * 64, dtype=torch.int64, device="cuda")
            L.sc_debug_set_gemm_trace(tr.data_ptr())
            ops.gemm(a, w, bias, act, resid, out=out, M=M, K=K, lda=ld)
            torch.cuda.synchronize()
            L.sc_debug_set_gemm_trace(None)
            t = tr.reshape(256, 8, 8).double()
            nk, E = K // 64, mode
            tn = N // 256; rows = 256 // tn; units = (M + 127) // 128
            tiles_g = units / rows / 2
            live = t.sum(dim=(1, 2)) > 0
            tt = t[live]
            for grp in (0, 1):
                x = tt[:, 4 * grp:4 * grp + 4, :].mean(dim=(0, 1))
                ns, nj, ne = tiles_g * E, tiles_g * (nk - E), tiles_g * E
                print(f"{name} mode{mode} g{grp}: per step  solo work {x[0]/ns:6.0f} +bar {x[1]/ns:6.0f} | joint work {x[2]/max(nj,1):6.0f} +bar {x[3]/max(nj,1):6.0f} | "
                      f"epi work {x[4]/ne:6.0f} +bar {x[5]/ne:6.0f} (c0 dma wait {x[7]/tiles_g:6.0f} per tile) | null {x[6]:8.0f}  total {x.sum():.0f}", flush=True)
        L.sc_debug_set_gemm_mode(-1)
        del a, w, out, resid


if __name__ == "__main__":
    args = sys.argv[1:]
    sustain = 1.0
    modes = [0, 4, 8]
    only = []
    global MODES
    i = 0
    todo = []
    while i < len(args):
        if args[i] == "--sustain": sustain = float(args[i + 1]); i += 2
        elif args[i] == "--modes": modes = [int(v) for v in args[i + 1].split(",")]; i += 2
        elif args[i] in ("check", "time", "trace", "trace8p", "trace8pp"): todo.append(args[i]); i += 1
        else: only.append(args[i]); i += 1
    if not todo: todo = ["check", "time"]
    MODES = modes
    ok = True
    if "check" in todo: ok = check()
    if "time" in todo: timeit(sustain, modes, only)
    if "trace" in todo: trace(only, modes)
    if "trace8p" in todo: trace8p(only)
    if "trace8pp" in todo: trace8pp(only)
    sys.exit(0 if ok else 1)

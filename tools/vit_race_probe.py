#!/usr/bin/env python3
"""Locate the op of the image tower whose output changes when the tower runs on a side stream beside a loaded main stream (tools/determinism_probe.py found
image_feat rows differing run to run only with the towers overlapped).  The tower is replayed op by op with a clone after every op; the reference is the same
sequence on an idle GPU.  usage: python tools/vit_race_probe.py [B] [runs] [load: gemm | speech]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from speechclip_amd import ops  # noqa: E402
from speechclip_amd.ops import ACT_QUICKGELU  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 12
load = sys.argv[3] if len(sys.argv) > 3 else "speech"
variant = sys.argv[4] if len(sys.argv) > 4 else "default"      # default | ln_generic | out_of_place | old_gemm | static_order
from speechclip_amd._lib import lib  # noqa: E402
model = bench.build_model().cuda().eval()
batch, lens = bench.make_batch(B, 160000, 0, "cuda")
clip = model.clip.model
v = clip.visual
P = clip.packed(torch.device("cuda"))
ntok = (v.input_resolution // v.patch) ** 2 + 1
W = clip.cfg.vision_width
image = batch["image"].float().contiguous()


def tower(record):
    cols = ops.vit_patchify(image, v.patch, P["Kpad"]); record("cols", cols)
    patch = ops.gemm(cols, P["conv_w"]); record("patch", patch)
    x = ops.vit_embed(patch, P["cls"], P["pos"], *P["ln_pre"], B, ntok, W); record("embed", x)
    M = x.shape[0]
    bf = torch.bfloat16
    n = torch.empty(M, W, device="cuda", dtype=bf); qkv = torch.empty(M, 3 * W, device="cuda", dtype=bf)
    att = torch.empty(M, W, device="cuda", dtype=bf); ffn = torch.empty(M, 4 * W, device="cuda", dtype=bf)
    x2 = torch.empty_like(x) if variant == "out_of_place" else x
    nf = torch.empty(M, W, device="cuda", dtype=torch.float32) if variant == "ln_generic" else None

    def ln(src, g, b):
        if variant.startswith("ln_v"):            # tools/probes/ln768f_variants.hip: the pre-fix kernel (SLP-packed) and three variants of it
            assert LV.ln768f_variant(int(variant[4:]), src.data_ptr(), g.data_ptr(), b.data_ptr(), n.data_ptr(), src.shape[0], 1e-5, torch.cuda.current_stream().cuda_stream) == 0
            return
        if nf is not None:
            ops.layernorm(src, g, b, out=nf)          # generic kernel (fp32 out), then a cast
            n.copy_(nf)
        else:
            ops.layernorm(src, g, b, out=n)
    for i, L_ in enumerate(P["vis"]):
        if variant == "diagnose":
            xc = x.clone()                                   # what a plain copy kernel sees in x right now
            ln(x, *L_["ln1"]); n1 = n.clone()                # n as a copy kernel sees it right behind the LayerNorm
            nchk = torch.empty_like(n); ops.layernorm(xc, *L_["ln1"], out=nchk)      # the same kernel on the COPY of x
            ops.gemm(n, L_["wqkv"], L_["bqkv"], out=qkv)
            n2 = n.clone()                                   # n again, one kernel later
            DIAG.append((i, n1, nchk, n2))
            record("L%02d ln1" % i, n1); record("L%02d qkv" % i, qkv)
        else:
            ln(x, *L_["ln1"]); record("L%02d ln1" % i, n)
            ops.gemm(n, L_["wqkv"], L_["bqkv"], out=qkv); record("L%02d qkv" % i, qkv)
        ops.attention(qkv, B, ntok, v.transformer.heads, None, out=att); record("L%02d att" % i, att)
        ops.gemm(att, L_["wo"], L_["bo"], residual=x, out=x2, out_f32=True); x, x2 = x2, x; record("L%02d out_proj+res" % i, x)
        ln(x, *L_["ln2"]); record("L%02d ln2" % i, n)
        ops.gemm(n, L_["w1"], L_["b1"], ACT_QUICKGELU, out=ffn); record("L%02d fc1" % i, ffn)
        ops.gemm(ffn, L_["w2"], L_["b2"], residual=x, out=x2, out_f32=True); x, x2 = x2, x; record("L%02d fc2+res" % i, x)
    cls = ops.layernorm(x, *P["ln_post"], rows=B, D=W, ld_in=ntok * W); record("ln_post", cls)
    out = ops.gemm(cls, P["proj_t"], out_f32=True); record("proj", out)


DIAG = []
DUMPED = []


def run(record_all):
    names, vals = [], []
    DIAG.clear()

    def rec(name, t):
        if record_all:
            names.append(name); vals.append(t.clone())
    tower(rec)
    return names, vals


if variant == "old_gemm":
    lib().sc_debug_set_gemm_mode(0)
if variant == "static_order":
    lib().sc_debug_set_gemm_mode(26)
LV = None
if variant.startswith("ln_v"):
    import ctypes
    so_ = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes", "ln768f_variants_bin.so")
    if not os.path.exists(so_):
        os.system("/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared %s -o %s" % (so_.replace("_bin.so", ".hip"), so_))
    LV = ctypes.CDLL(so_)
    LV.ln768f_variant.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p]
print("variant:", variant)
torch.cuda.synchronize()
names, ref = run(True)
torch.cuda.synchronize()
names2, again = run(True)
torch.cuda.synchronize()
print("idle GPU, two runs identical:", all(torch.equal(a, b) for a, b in zip(ref, again)))
side = torch.cuda.Stream()
a_big = torch.randn(128000, 768, device="cuda").to(torch.bfloat16); w_big = torch.randn(3072, 768, device="cuda").to(torch.bfloat16)
o_big = torch.empty(128000, 3072, device="cuda", dtype=torch.bfloat16)
hb = torch.randn(128000, 768, device="cuda").to(torch.bfloat16); hb2 = torch.empty_like(hb)
lng, lnb = torch.ones(768, device="cuda"), torch.zeros(768, device="cuda")
qkv_big = torch.randn(B * 500, 3 * 768, device="cuda").to(torch.bfloat16) if load == "attn" else None
att_big = torch.empty(B * 500, 768, device="cuda", dtype=torch.bfloat16) if load == "attn" else None
first_bad = {}
for r in range(runs):
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        _, vals = run(True)
    if load == "gemm":
        for _ in range(12):
            ops.gemm(a_big, w_big, None, 1, out=o_big)
    elif load.startswith("front"):       # the speech tower's front end only: conv0 .. positional conv (front), or up to transformer layer k (front<k>)
        hub = model.audio_encoder.encoder
        k = int(load[5:] or 0)
        for _ in range(3 if k == 0 else 1):
            hub.extract_all_layers(batch["wav"], [int(v_) for v_ in batch["wav_len"].tolist()], stop_layer=k)
    elif load == "ln":                    # a stream of HBM-bound row kernels
        for _ in range(200):
            ops.layernorm(hb, lng, lnb, out=hb2)
    elif load == "attn":
        for _ in range(60):
            ops.attention(qkv_big, B, 500, 12, None, out=att_big)
    else:
        with torch.no_grad():
            model.forward_audio(batch["wav"], batch["wav_len"])
    cur.wait_stream(side)
    torch.cuda.synchronize()
    if variant == "diagnose":
        for (i, n1, nchk, n2) in DIAG:
            d1, d2 = int((n1 != nchk).sum()), int((n1 != n2).sum())
            if d1 and not DUMPED:
                DUMPED.append(1)
                k = names.index("L%02d ln1" % i)
                xk = names.index("L%02d fc2+res" % (i - 1)) if i > 0 else names.index("embed")
                xin = vals[xk]                                           # the layer's input x as recorded (a copy kernel's view)
                rows_bad = (n1 != nchk).any(1).nonzero().flatten().tolist()
                rr = rows_bad[0]
                g_, b_ = P["vis"][i]["ln1"]
                xr = xin[rr].double()
                want = ((xr - xr.mean()) / torch.sqrt(xr.var(unbiased=False) + 1e-5) * g_.double() + b_.double())
                cols = (n1[rr] != nchk[rr]).nonzero().flatten().tolist()
                print("   DUMP layer %d row %d (bad rows %s): recorded input row == reference input row: %s" % (i, rr, rows_bad[:8], bool(torch.equal(xin[rr], ref[xk][rr]))))
                print("   differing columns (%d): %s" % (len(cols), cols[:40]))
                e1 = (n1[rr].double() - want).abs(); e2 = (nchk[rr].double() - want).abs()
                print("   |LN(x) - exact| max %.4e mean %.4e;  |LN(copy) - exact| max %.4e mean %.4e   (bf16 half-ulp at 1.0 = 3.9e-3)" % (e1.max(), e1.mean(), e2.max(), e2.mean()))
                print("   LN(x)==ref row: %s   LN(copy)==ref row: %s" % (bool(torch.equal(n1[rr], ref[k][rr])), bool(torch.equal(nchk[rr], ref[k][rr]))))
            if d1 or d2:
                print("   run %d layer %d: LN(x) vs LN(copy of x): %d elements differ; n right behind the LN vs n one kernel later: %d differ" % (r, i, d1, d2))
    bad = [(n, int((a != b).sum()), float((a.float() - b.float()).abs().max())) for n, a, b in zip(names, vals, ref) if not torch.equal(a, b)]
    if bad:
        n0 = bad[0]
        k = names.index(n0[0])
        rows = (vals[k] != ref[k]).reshape(vals[k].shape[0], -1).any(1).nonzero().flatten()
        print("run %d: first differing op: %s (%d elements, max |diff| %.3e), rows %s..%s (%d rows), ops differing %d" % (r, n0[0], n0[1], n0[2], rows[:4].tolist(), rows[-2:].tolist(), rows.numel(), len(bad)))
        first_bad[n0[0].split(" ", 1)[-1]] = first_bad.get(n0[0].split(" ", 1)[-1], 0) + 1
    else:
        print("run %d: identical" % r)
print("first differing op, by kind:", first_bad)

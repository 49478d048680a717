#!/usr/bin/env python3
"""Per-phase cycle breakdown of attn_fwd_kernel (debug hook sc_debug_set_attn_trace; wave 0 of every block): S = K.Q^T complete /
softmax arithmetic / P.V complete / end-of-tile wait + barrier, averaged per KV tile."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechclip_amd import ops, _lib
B, T, H = int(os.environ.get("ATTN_B", "256")), 500, 12
qkv = torch.randn(B * T, 3 * H * 64, device="cuda").to(torch.bfloat16)
lens = torch.full((B,), 499, dtype=torch.int32, device="cuda")
out = torch.empty(B * T, H * 64, device="cuda", dtype=torch.bfloat16)
L = _lib.lib()
L.sc_debug_set_attn_trace.argtypes = [ctypes.c_void_p]
ops.attention(qkv, B, T, H, lens, out=out)
nblk = ((B * H + 7) // 8) * 8 * 4
NW = 8
tr = torch.zeros(nblk * NW, 8, dtype=torch.int64, device="cuda")
L.sc_debug_set_attn_trace(tr.data_ptr())
ops.attention(qkv, B, T, H, lens, out=out)
torch.cuda.synchronize()
L.sc_debug_set_attn_trace(None)
tw = tr.view(nblk, NW, 8).double().cpu()
t = tw[:, 0]
keep = t[:, 4] > 0
tw = tw[keep]
t = t[keep]
per = t[:, :4] / t[:, 4:5]
m = per.mean(0)
print(f"blocks traced {len(t)}  tiles/block {t[:,4].mean():.1f}  per-tile cycles: QK-done={m[0]:.0f} softmax={m[1]:.0f} PV-done={m[2]:.0f} wait+barrier={m[3]:.0f} total={m.sum():.0f}")
import numpy as np
st = (t[:, 5] - t[:, 5].min()).numpy() * 10.0        # ns (s_memrealtime ticks at 100 MHz)
en = (t[:, 6] - t[:, 5].min()).numpy() * 10.0
dur = en - st
print(f"kernel span {en.max()/1e3:.1f} us; block lifetime mean {dur.mean()/1e3:.1f} us (min {dur.min()/1e3:.1f}, max {dur.max()/1e3:.1f})")
# resident blocks over time: sample at 25 %, 50 %, 75 % of the span
for f in (0.1, 0.25, 0.5, 0.75):
    ts = f * en.max()
    print(f"  resident blocks at {f:.2f} of the span: {int(((st <= ts) & (en > ts)).sum())}  (256 CUs)")

# per-wave picture: cycles per tile in each phase, by wave index (0..7), and by SIMD (HW_ID bits 4..5)
pw = tw[:, :, :4] / tw[:, :, 4:5]
print("wave   QK-done  softmax  PV-done  wait+barrier   (mean cycles per tile)")
for w in range(NW):
    m = pw[:, w].mean(0)
    print(f"  {w}   {m[0]:8.0f} {m[1]:8.0f} {m[2]:8.0f} {m[3]:8.0f}")
work = pw[:, :, :3].sum(-1)                      # busy cycles per tile per wave
print(f"busy cycles/tile: fastest wave of a block {work.min(1).values.mean():.0f}, slowest {work.max(1).values.mean():.0f}, mean {work.mean():.0f}; "
      f"barrier wait min-wave {pw[:, :, 3].min(1).values.mean():.0f} max-wave {pw[:, :, 3].max(1).values.mean():.0f}")
hw = tw[:, :, 7].long()
simd = (hw >> 4) & 3
for sd in range(4):
    sel = simd == sd
    print(f"  SIMD {sd}: waves {int(sel.sum())}, busy/tile {work[sel].mean():.0f}")

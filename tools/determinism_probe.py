#!/usr/bin/env python3
"""Run-to-run bitwise reproducibility of the forward path on one GPU: N forwards of the same batch, every hidden state of the speech tower, the image
features and the embeddings compared with the first run; reports the first tensor that differs (layer index, elements, max |diff|).
usage: python tools/determinism_probe.py [runs] [gemm_mode: -1 default | 26 static tile order | 0 old kernels] [B] [base | large | cascaded]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from speechclip_amd._lib import lib  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 20
mode = int(sys.argv[2]) if len(sys.argv) > 2 else -1
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
kind = sys.argv[4] if len(sys.argv) > 4 else "base"          # base | large | cascaded
lib().sc_debug_set_gemm_mode(mode)
model = bench.build_model(large=kind == "large", cascaded=kind == "cascaded").cuda().eval()
batch, lens = bench.make_batch(B, 160000, 0, "cuda")
lens = list(lens)
for i in range(0, B, 7):
    lens[i] = 96000
    batch["wav"][i, lens[i]:] = 0
batch["wav_len"] = torch.tensor(lens).cuda()


def once():
    with torch.no_grad():
        _, flen, hidden = model.forward_audio(batch["wav"], batch["wav_len"], return_hidden_states=True)
        lf, _, _ = model(batch)
    torch.cuda.synchronize()
    out = {"hidden%02d" % i: h.clone() for i, h in enumerate(hidden)}
    out["image_feat"] = lf["image_feat"].clone()
    for k in ("parallel_audio_feat", "cascaded_audio_feat"):
        if k in lf and lf[k] is not None:
            out[k] = lf[k].clone()
    return out


ref = once()
bad = 0
for r in range(1, runs):
    cur = once()
    diffs = [(k, int((cur[k] != ref[k]).sum()), float((cur[k].float() - ref[k].float()).abs().max())) for k in ref if not torch.equal(cur[k], ref[k])]
    if diffs:
        bad += 1
        print("run %d differs: first %s (%d elements, max |diff| %.3e); tensors differing: %d of %d" % (r, diffs[0][0], diffs[0][1], diffs[0][2], len(diffs), len(ref)))
        k = diffs[0][0]
        idx = (cur[k] != ref[k]).nonzero()
        print("   where:", idx[:6].tolist(), "...", idx[-2:].tolist(), "shape", tuple(ref[k].shape))
print("mode %d: %d of %d repeat runs differ from the first" % (mode, bad, runs - 1))

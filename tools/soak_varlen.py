#!/usr/bin/env python3
"""Soak test of the padding-free engine: 200 forward + 100 training steps at B = 256 with NEW random lengths every step (L_i ~ U{32000..160000}): the row
count of every workspace changes each step -- step time per 50-step window, memory growth (capacity buffers only grow), finite losses, and every
20th step checked against the padded engine on the same batch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, bench
from speechclip_amd import parallel
model = bench.build_model().cuda()
B, L = 256, 160000
g = torch.Generator().manual_seed(3)
base = (0.1 * torch.randn(B, L, generator=g)).cuda()
img = torch.randn(B, 3, 224, 224, generator=g).cuda()
ar = torch.arange(L, device="cuda")[None, :]
def batch():
    lens = torch.randint(32000, L + 1, (B,), generator=g)
    lm = int(lens.max())
    wav = (base[:, :lm] * (ar[:, :lm] < lens.cuda()[:, None])).contiguous()
    return {"wav": wav, "wav_len": lens, "image": img, "id": torch.arange(B).cuda()}
def fwd(b):
    with torch.no_grad():
        lf, _, _ = model(b); return lf, model.compute_loss(parallel.gather_loss_feats(lf))["loss"]
for _ in range(3): fwd(batch())
torch.cuda.synchronize(); m0 = torch.cuda.memory_allocated(); r0 = torch.cuda.memory_reserved()
t = time.perf_counter(); times = []; worst = 1.0
for i in range(200):
    b = batch()
    lf, l = fwd(b)
    assert torch.isfinite(l), i
    if i % 20 == 0:
        os.environ["SC_VARLEN_PACK"] = "0"
        lf0, l0 = fwd(b)
        os.environ.pop("SC_VARLEN_PACK")
        c = torch.nn.functional.cosine_similarity(lf["parallel_audio_feat"].float(), lf0["parallel_audio_feat"].float(), dim=-1).min().item()
        worst = min(worst, c)
        assert c > 0.9999 and abs(float(l) - float(l0)) < 1e-3, (i, c, float(l), float(l0))
    if i % 50 == 49:
        torch.cuda.synchronize(); t1 = time.perf_counter(); times.append((t1 - t) / 50 * 1e3); t = t1
print("varlen fwd ms/step per 50-step window (incl. batch synthesis and the padded cross-checks):", [round(x, 2) for x in times], "min cos packed/padded", round(worst, 6),
      "alloc delta MB", (torch.cuda.memory_allocated() - m0) / 1e6, "reserved delta MB", (torch.cuda.memory_reserved() - r0) / 1e6)
model.train(); (opt,), (sch,) = model.configure_optimizers()
def tr(b):
    opt.zero_grad(); loss = model.training_step_end(model.training_step(b, 0))["loss"]; loss.backward(); opt.step(); sch["scheduler"].step(); return loss.detach()
for _ in range(3): tr(batch())
torch.cuda.synchronize(); m0 = torch.cuda.memory_allocated(); r0 = torch.cuda.memory_reserved(); t = time.perf_counter(); times = []; ls = []
for i in range(100):
    l = tr(batch())
    if i % 50 == 49:
        torch.cuda.synchronize(); t1 = time.perf_counter(); times.append((t1 - t) / 50 * 1e3); t = t1; ls.append(float(l))
        assert torch.isfinite(l)
print("varlen train ms/step per 50-step window:", [round(x, 2) for x in times], "losses", [round(x, 4) for x in ls], "alloc delta MB",
      (torch.cuda.memory_allocated() - m0) / 1e6, "reserved delta MB", (torch.cuda.memory_reserved() - r0) / 1e6)

#!/usr/bin/env python3
"""Odd-shape sweep of the training step of the trainable tail (parallel and cascaded tiny models): ragged and tiny batches run and give finite gradients."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from forward_fuzz import build


def run(model, lens, tag):
    B, L = len(lens), max(lens)
    g = torch.Generator().manual_seed(sum(lens) + B)
    wav = torch.zeros(B, L)
    for i, l in enumerate(lens):
        wav[i, :l] = 0.3 * torch.randn(l, generator=g)
    batch = {"wav": wav.cuda(), "wav_len": torch.tensor(lens).cuda(), "image": torch.randn(B, 3, 64, 64, generator=g).cuda(), "id": torch.arange(B).cuda()}
    model.train()
    model.zero_grad()
    try:
        loss = model.training_step_end(model.training_step(batch, 0))["loss"]
        loss.backward()
    except (ValueError, RuntimeError) as e:
        print(f"{tag} lens={lens}: raised {type(e).__name__}: {str(e)[:120]}")
        return
    bad = [k for k, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    n = sum(p.grad is not None for p in model.parameters())
    print(f"{tag} lens={lens}: loss {loss.item():.4f}, {n} gradient tensors, non-finite: {bad}")
    assert not bad


for cascaded in (False, True):
    os.environ["SC_FROZEN_DROPOUT"] = "1"
    model, _ = build(cascaded, False)
    for lens in ([8000], [400, 400], [401, 8000, 123456 // 20], [2000] * 5, [719, 720, 721, 8000]):
        run(model, lens, "C-base" if cascaded else "P-base")
print("train fuzz OK")

import dataclasses, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["SC_FROZEN_DROPOUT"] = "0"
import torch, numpy as np
import torch.nn.functional as F
from test_finetune_gpu import _finetune_pair, _cos
from oracle import hubert_ref as HR, speechclip_ref as R

def run(lens, large, everything, layers):
    model, ref, batch = _finetune_pair(layers, everything=everything, large=large)
    B = len(lens)
    g = torch.Generator().manual_seed(sum(lens))
    L = max(lens)
    wav = torch.zeros(B, L)
    for i, l in enumerate(lens):
        wav[i, :l] = 0.3 * torch.randn(l, generator=g)
    batch = {"wav": wav, "wav_len": torch.tensor(lens), "image": batch["image"][:1].repeat(B, 1, 1, 1) + 0.1 * torch.randn(B, *batch["image"].shape[1:], generator=g), "id": torch.arange(B)}
    model = model.cuda().eval()
    feats, _, _ = model({k: v.cuda() for k, v in batch.items()})
    loss = model.compute_loss(feats)["loss"]
    loss.backward()
    for p in ref.parameters(): p.requires_grad_(False)
    names = []
    for k, p in ref.encoder.named_parameters():
        on = (not k.startswith(("mask_emb", "final_proj", "label_embs_concat") + (("encoder.layer_norm",) if large else ()))) if everything else any(f"encoder.layers.{i}." in k for i in layers)
        p.requires_grad_(on)
        if on: names.append(k)
    ref.encoder.feature_grad_mult = (1.0 if large else 0.1) if everything else 0
    for p in ref.parallel_branch.parameters(): p.requires_grad_(True)
    ref.ws_weights.requires_grad_(True)
    wavs = [wav[b, :lens[b]] for b in range(B)]
    padded, mask = HR.preprocess_input(wavs, ref.hubert_cfg.normalize)
    with torch.enable_grad():
        hidden = HR.hubert_forward.__wrapped__(ref.encoder, padded, mask)["layer_results"]
        flen = HR.feat_lengths(lens, 320, hidden[-1].shape[1])
        pa = R.l2_normalize(ref.parallel_branch(R.weighted_sum(hidden, ref.ws_weights, large), flen))
        with torch.no_grad(): img = R.l2_normalize(ref.clip.encode_image(batch["image"]))
        rl = R.masked_contrastive_loss(pa, img, batch["id"], ref.inv_temperature)
    rl.backward()
    mine = dict(model.named_parameters())
    worst = (1.0, "")
    rp = dict(ref.encoder.named_parameters())
    for k in names:
        gm, gr = mine["audio_encoder.encoder." + k].grad, rp[k].grad
        if gr is None or gr.norm() < 1e-7: continue
        assert gm is not None and torch.isfinite(gm).all(), k
        worst = min(worst, (_cos(gm, gr), k))
    print(f"lens={lens} large={large} all={everything} layers={layers}: loss {loss.item():.4f} vs {rl.item():.4f}; worst cos {worst}")

run([8000], False, True, [])
run([3000, 8000, 5123], False, True, [])
run([2500, 1900], False, True, [])
run([8000, 700, 4000], True, True, [])
run([4000], True, False, [1, 2])
run([900, 640, 1300, 801, 2000], False, False, [0, 2])

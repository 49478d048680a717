"""End-to-end recall@K parity on a DISCRIMINATIVE retrieval set (">= 1000-pair synthetic retrieval set within
+-0.5 pt and identical top-1 on well-separated pairs"; reference: avssl/module/retrieval.py:45-121, avssl/model/kwClip.py:468-502).

With random-init towers every candidate scores within a few 1e-3 of every other one, so a recall comparison there can only bound flips.
Here the retrieval set has class structure and the trainable tail is TRAINED on it first:

  * 200 image prototypes (fixed random 64 x 64 images; the tiny CLIP tower is re-scaled so its embeddings are spread: mean pairwise
    cosine 0.57 instead of 0.94) and, per prototype, a three-tone mixture; an utterance = that mixture with random phases, amplitude,
    length (0.25 - 0.5 s) and white noise.  Flickr8k's structure: 5 spoken captions per image.
  * the tail (parallel branch + layer-mix weights) is trained for 400 steps of 64 pairs with the HIP trainer (training_step ->
    training_step_end -> backward -> FusedAdam), frozen bf16 towers;
  * the trained state_dict is loaded into the fp32 CPU oracle; 1000 HELD-OUT utterances (fresh noise / phases / lengths) and the 200
    images are embedded on both sides: the HIP side through Lightning's validation hooks (validation_step -> validation_step_end ->
    validation_epoch_end -> mutualRetrieval on sc_sgemm + sc_retrieval_ranks), the oracle through its own forward + argsort ranks.

Asserted: the oracle's recall@1 is far above chance (0.5 %), so the test cannot pass on noise; |recall@K(HIP) - recall@K(oracle)| <= 0.5
points for K = 1, 5, 10 in both directions; the top-1 candidate is IDENTICAL for every query whose oracle top-1 / top-2 margin exceeds
0.05 -- no flip-count allowance.  The margin histogram is printed."""
import dataclasses
import math

import pytest
import torch

from helpers import make_config

pytestmark = pytest.mark.gpu

N_PROTO, CAPS, SR = 200, 5, 16000


class ToneSet:
    """Prototype p = (image_p, three tones with amplitudes); render() draws utterances of the given prototypes."""

    def __init__(self, seed=5, res=64):
        g = torch.Generator().manual_seed(seed)
        self.images = torch.randn(N_PROTO, 3, res, res, generator=g)
        self.freqs = 150.0 + 3500.0 * torch.rand(N_PROTO, 3, generator=g)
        self.amps = 0.5 + torch.rand(N_PROTO, 3, generator=g)

    def render(self, protos, gen, noise=0.005):
        ps = torch.as_tensor(protos)
        n = len(ps)
        lens = torch.randint(4000, 8000, (n,), generator=gen)
        lmax = int(lens.max())
        t = torch.arange(lmax, dtype=torch.float32) / SR
        ph = 2 * math.pi * torch.rand(n, 3, generator=gen)
        gain = 0.1 * (0.7 + 0.6 * torch.rand(n, 1, generator=gen))
        w = (self.amps[ps][:, :, None] * torch.sin(2 * math.pi * self.freqs[ps][:, :, None] * t[None, None, :] + ph[:, :, None])).sum(1)
        w = gain * w + noise * torch.randn(n, lmax, generator=gen)
        w = w * (torch.arange(lmax)[None, :] < lens[:, None])
        return {"wav": w, "wav_len": lens, "image": self.images[ps], "id": ps.clone()}


def spread_image_tower(visual):
    """Random-init ViT embeddings are dominated by the input-independent class / positional terms (pairwise cosine 0.94): weight the
    patch embedding and the residual branches up so the embedding follows the image (a statement about the TEST weights, applied
    identically to the oracle through the state_dict)."""
    with torch.no_grad():
        visual.conv1.weight.mul_(8.0)
        visual.class_embedding.mul_(0.05)
        visual.positional_embedding.mul_(0.05)
        for blk in visual.transformer.resblocks:
            blk.attn.out_proj.weight.mul_(4.0)
            blk.mlp.c_proj.weight.mul_(4.0)


def _build():
    from oracle.clip_ref import ClipRefConfig
    from oracle.hubert_ref import HubertRefConfig
    from speechclip_amd.model import KWClip_GeneralTransformer
    from speechclip_amd.module.clip_model import ClipConfig
    from speechclip_amd.module.hubert import HubertConfig
    href, cref = HubertRefConfig.tiny(), ClipRefConfig.tiny()
    cfg = make_config(d_model=128, branch_heads=4, hubert_config=HubertConfig(**dataclasses.asdict(href)),
                      clip_config=ClipConfig(**dataclasses.asdict(cref)))
    cfg.audio_encoder.optim.args.lr = 2e-3
    cfg.audio_encoder.scheduler.warmup = 10
    torch.manual_seed(0)
    model = KWClip_GeneralTransformer(cfg)
    spread_image_tower(model.clip.model.visual)
    return model, href, cref


def _oracle_from(model, href, cref):
    from oracle.speechclip_ref import SpeechClipRef
    ref = SpeechClipRef(href, cref, parallel=True, branch_heads=4).eval()
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    ref.encoder.load_state_dict({k[len("audio_encoder.encoder."):]: v for k, v in sd.items() if k.startswith("audio_encoder.encoder.")})
    ref.clip.load_state_dict({k[len("clip.model."):]: v for k, v in sd.items() if k.startswith("clip.model.")})
    ref.parallel_branch.load_state_dict({k[len("parallel_branch."):]: v for k, v in sd.items() if k.startswith("parallel_branch.")})
    with torch.no_grad():
        ref.ws_weights.copy_(sd["audio_encoder.weightedsum_layer.weights"])
    return ref


def _hist(m, edges=(0.0, 0.005, 0.01, 0.02, 0.05, 0.1, 0.2, 0.4, 2.0)):
    return {f"[{a:g},{b:g})": int(((m >= a) & (m < b)).sum()) for a, b in zip(edges[:-1], edges[1:])}


def test_recall_at_k_parity_on_a_trained_1000_pair_set():
    from oracle.speechclip_ref import mutual_retrieval
    data = ToneSet()
    model, href, cref = _build()
    model = model.cuda().train()
    (opt,), (sch,) = model.configure_optimizers()
    gen = torch.Generator().manual_seed(11)
    losses = []
    for step in range(400):
        b = data.render(torch.randperm(N_PROTO, generator=gen)[:64], gen)
        opt.zero_grad()
        out = model.training_step({k: v.cuda() for k, v in b.items()}, step)
        loss = model.training_step_end(out)["loss"]
        loss.backward()
        opt.step()
        sch["scheduler"].step()
        losses.append(float(loss.detach()))
    assert all(math.isfinite(x) for x in losses)
    print(f"tail training: loss {sum(losses[:5]) / 5:.3f} -> {sum(losses[-5:]) / 5:.3f}")
    assert sum(losses[-20:]) / 20 < 0.5 * sum(losses[:5]) / 5, (losses[:5], losses[-5:])
    model.eval()
    ref = _oracle_from(model, href, cref)

    # ---- the held-out retrieval set: 1000 utterances, caption c of image p is utterance 5 p + c; 10 dev batches of 100
    gen2 = torch.Generator().manual_seed(99)
    n_utt = N_PROTO * CAPS
    batches = [data.render([(i * 100 + k) // CAPS for k in range(100)], gen2) for i in range(n_utt // 100)]
    outs = []
    with torch.no_grad():
        for i, b in enumerate(batches):
            outs.append(model.validation_step_end(model.validation_step({k: v.cuda() for k, v in b.items()}, i)))
        r_ab, r_ba, r_mean = model.validation_epoch_end(outs)
    aud_o, img_o = [], {}
    for b in batches:
        o = ref(b)
        aud_o.append(o["parallel_audio_feat"])
        for j, _id in enumerate(b["id"].tolist()):
            img_o[_id] = o["image_feat"][j]            # validation_epoch_end keeps the LAST occurrence of an id, ids in first-seen order
    all_ids = torch.cat([b["id"] for b in batches])
    img_ids = torch.tensor(list(img_o.keys()))
    score_o = torch.cat(aud_o) @ torch.stack([img_o[i] for i in img_ids.tolist()]).t()            # [1000, 200]
    o_ab, o_ba, o_mean = mutual_retrieval(score_o, score_o.t().contiguous(), all_ids, img_ids, [1, 5, 10])
    print("recall A->I  HIP", {k: round(v, 2) for k, v in r_ab.items()}, " oracle", {k: round(v, 2) for k, v in o_ab.items()})
    print("recall I->A  HIP", {k: round(v, 2) for k, v in r_ba.items()}, " oracle", {k: round(v, 2) for k, v in o_ba.items()})

    # the set is discriminative: far above chance (1 / 200 = 0.5 %), far from saturated margins
    assert o_ab["recall@1"] > 50.0 and o_ba["recall@1"] > 50.0, (o_ab, o_ba)
    for k in ("recall@1", "recall@5", "recall@10"):
        assert abs(r_ab[k] - o_ab[k]) <= 0.5 + 1e-4, ("A->I", k, r_ab[k], o_ab[k])
        assert abs(r_ba[k] - o_ba[k]) <= 0.5 + 1e-4, ("I->A", k, r_ba[k], o_ba[k])

    # ---- identical top-1 wherever the oracle's decision is not a near-tie (margin > 0.05 in cosine units)
    aud_d = torch.cat([x["audio_feat"] for x in outs]).float()
    last = {}
    for i, _id in enumerate(torch.cat([x["id"] for x in outs]).tolist()):
        last[_id] = i
    img_d = torch.cat([x["image_feat"] for x in outs]).float()[torch.tensor([last[i] for i in img_ids.tolist()])]
    score_d = aud_d @ img_d.t()
    print(f"score matrices: max |HIP - oracle| = {(score_d - score_o).abs().max():.4f}, oracle score range [{score_o.min():.3f}, {score_o.max():.3f}]")
    for name, so, sd_ in (("A->I", score_o, score_d), ("I->A", score_o.t(), score_d.t())):
        top2 = torch.topk(so, 2, dim=1)
        margin = top2.values[:, 0] - top2.values[:, 1]
        decisive = margin > 0.05
        same = sd_.argmax(dim=1) == top2.indices[:, 0]
        print(f"{name}: oracle top-1/top-2 margin histogram {_hist(margin)}; decisive (> 0.05): {int(decisive.sum())} of {len(margin)}; "
              f"top-1 agrees on {int(same.sum())} of {len(margin)} overall")
        assert bool(same[decisive].all()), (name, "top-1 differs on a decisive query", torch.nonzero(decisive & ~same).flatten().tolist(),
                                            margin[decisive & ~same].tolist())
        if name == "A->I":
            assert int(decisive.sum()) >= 400, "the decisive set must not be vacuous"

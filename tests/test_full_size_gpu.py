"""Full-size runs of the configurations the fp32 oracle is too slow for (BASELINE.json configs[2] C-base, configs[4] P-large), checked through
size-independent properties (the P-base analogue is tests/test_e2e_gpu.py::test_full_size_properties):

  * bitwise run-to-run determinism, unit-norm embeddings, finite outputs;
  * batch-permutation equivariance of the embeddings (not bitwise: the GEMM rotates its K loop by the tile's M-panel index, and the packed
    layout moves an utterance's rows with its position);
  * utterance independence: an utterance's embedding does not change when the OTHER utterances' audio / images are replaced (lengths kept:
    the batch maximum is the one thing utterances share in the reference -- speech_encoder_plus.py:506-518);
  * the padding-free engine and the padded engine agree on the same ragged batch;
  * the HIP masked InfoNCE equals the fp32 oracle loss evaluated on the SAME embeddings (trainable temperature for the large configs);
  * C-base: VQ targets inside the reduced vocabulary, keyword count, the key-padding semantics of the keyword head (changing samples beyond
    wav_len changes nothing).
Reference configs: config/speechCLIP/model_large/coco/spchclp_p.yaml:10,104 (batch 64 over 4 GPUs = 16 per GPU; SpokenCOCO utterances are a
few seconds long), config/speechCLIP/model_base/spchclp_c.yaml:94 (reduced vocabulary of 8112 sub-words)."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _cos_rows(a, b):
    return F.cosine_similarity(a.float().cpu(), b.float().cpu(), dim=-1)


def _ragged_batch(B, lens, seed, res=224):
    g = torch.Generator().manual_seed(seed)
    wav = torch.zeros(B, max(lens))
    for i, n in enumerate(lens):
        wav[i, :n] = 0.1 * torch.randn(n, generator=g)
    return {"wav": wav.cuda(), "wav_len": torch.tensor(lens).cuda(), "image": torch.randn(B, 3, res, res, generator=g).cuda(),
            "id": (torch.arange(B) // 2).cuda()}


def _with_env(key, val, fn):
    old = os.environ.get(key)
    os.environ[key] = val
    try:
        return fn()
    finally:
        if old is None:
            os.environ.pop(key)
        else:
            os.environ[key] = old


def test_p_large_full_size_properties():
    """HuBERT-large + ViT-L/14 (wave layer-norm, LayerNorm extractor, pre-LN layers, normalised layer mix, trainable temperature), 16 pairs per
    GPU, SpokenCOCO-shaped lengths (2 - 15 s)."""
    import bench
    from oracle.speechclip_ref import masked_contrastive_loss
    model = bench.build_model(large=True).cuda()
    B = 16
    g = torch.Generator().manual_seed(5)
    lens = [int(x) for x in torch.randint(32000, 240001, (B,), generator=g)]
    lens[3] = 240000
    batch = _ragged_batch(B, lens, 6)
    with torch.no_grad():
        lf1, lm, _ = model(batch)
        a1, i1 = lf1["parallel_audio_feat"].clone(), lf1["image_feat"].clone()
        lf2, _, _ = model(batch)
        assert torch.equal(a1, lf2["parallel_audio_feat"]) and torch.equal(i1, lf2["image_feat"]), "not run-to-run deterministic"
        perm = torch.randperm(B, generator=g).cuda()
        lf3, _, _ = model({k: v[perm] for k, v in batch.items()})
        loss = model.compute_loss(lf1)["loss"].item()
        # other utterances replaced (same lengths): utterances 0, 3, 7 keep their audio and images
        keep = [0, 3, 7]
        other = _ragged_batch(B, lens, 77)
        for k in ("wav", "image"):
            other[k][keep] = batch[k][keep]
        lf4, _, _ = model(other)
        a_pad = _with_env("SC_VARLEN_PACK", "0", lambda: model(batch)[0]["parallel_audio_feat"].clone())
        a_pack = _with_env("SC_VARLEN_PACK", "1", lambda: model(batch)[0]["parallel_audio_feat"].clone())
    assert a1.shape == (B, 768) and i1.shape == (B, 768) and torch.isfinite(a1).all() and torch.isfinite(i1).all()
    torch.testing.assert_close(a1.norm(dim=-1), torch.ones(B, device="cuda"), atol=1e-5, rtol=0)
    torch.testing.assert_close(i1.norm(dim=-1), torch.ones(B, device="cuda"), atol=1e-5, rtol=0)
    assert _cos_rows(lf3["image_feat"], i1[perm]).min().item() > 0.99999
    assert _cos_rows(lf3["parallel_audio_feat"], a1[perm]).min().item() > 0.9999
    assert _cos_rows(lf4["parallel_audio_feat"][keep], a1[keep]).min().item() > 0.9999, "an utterance's embedding depends on its batch neighbours"
    assert _cos_rows(lf4["image_feat"][keep], i1[keep]).min().item() > 0.99999
    assert _cos_rows(a_pack, a_pad).min().item() > 0.9999, "padding-free engine != padded engine at full size"
    geo = model.audio_encoder.encoder.packed_geometry(lens, max(lens))
    assert geo["total"] < 0.75 * geo["padded_rows"]
    inv_t = float(lm["cl_temp"])
    assert abs(inv_t - 1 / 0.07) < 1e-3
    ref = masked_contrastive_loss(a1.cpu(), i1.cpu(), batch["id"].cpu(), inv_temperature=inv_t).item()
    assert abs(loss - ref) < 1e-4, (loss, ref)
    print(f"P-large B={B}: loss {loss:.5f} (oracle on the same embeddings {ref:.5f}); packed rows {geo['total']} of {geo['padded_rows']}; "
          f"min cos packed/padded {_cos_rows(a_pack, a_pad).min().item():.6f}")


def test_c_base_full_size_properties():
    """Cascaded SpeechCLIP base as shipped: reduced vocabulary of 8112 sub-words, 8 keywords, B = 64, 10 s audio with a few shorter utterances."""
    import bench
    from oracle.speechclip_ref import masked_contrastive_loss
    model = bench.build_model(cascaded=True, vocab=8112).cuda()
    B, L = 64, 160000
    lens = [L] * B
    for i in range(0, B, 5):
        lens[i] = 64000 + 3000 * i
    batch = _ragged_batch(B, lens, 9)
    with torch.no_grad():
        lf1, lm, ot1 = model(batch)
        c1, i1 = lf1["cascaded_audio_feat"].clone(), lf1["image_feat"].clone()
        tg = lambda o: o["vq_results"]["targets"].reshape(B, -1)      # noqa: E731  ([B, K, 1] as the reference returns them)
        t1 = tg(ot1).clone()
        lf2, _, ot2 = model(batch)
        assert torch.equal(c1, lf2["cascaded_audio_feat"]) and torch.equal(t1, tg(ot2)), "not run-to-run deterministic"
        loss = model.compute_loss(lf1)["loss"].item()
        # samples beyond wav_len are never read (speech_encoder_plus.py:520-534 slices wav[:wav_len])
        dirty = dict(batch, wav=batch["wav"].clone())
        for i, n in enumerate(lens):
            dirty["wav"][i, n:] = 3.0
        lf3, _, ot3 = model(dirty)
        perm = torch.randperm(B, generator=torch.Generator().manual_seed(1)).cuda()
        lf4, _, ot4 = model({k: v[perm] for k, v in batch.items()})
    V = model.clip.model.token_embedding.weight.shape[0]
    assert V == 8112 and t1.shape == (B, 8) and int(t1.min()) >= 0 and int(t1.max()) < V
    assert ot1["keywords"].shape[:2] == (B, 8) and torch.isfinite(ot1["keywords"]).all()
    assert c1.shape == (B, 512) and torch.isfinite(c1).all()
    torch.testing.assert_close(c1.norm(dim=-1), torch.ones(B, device="cuda"), atol=1e-5, rtol=0)
    assert torch.equal(lf3["cascaded_audio_feat"], c1) and torch.equal(tg(ot3), t1), "samples beyond wav_len reached the output"
    # permutation: the sub-word arg-max of a random-init model has near-ties (margins ~1e-3), and the fp32 summation order
    # of a row moves with its position in the batch: most targets must survive, and where ALL 8 keywords of an utterance do, so does its embedding
    same = (tg(ot4) == t1[perm]).all(dim=1)
    assert (tg(ot4) == t1[perm]).float().mean().item() > 0.95
    assert int(same.sum()) >= B // 2
    assert _cos_rows(lf4["cascaded_audio_feat"][same], c1[perm][same]).min().item() > 0.9999
    ref = masked_contrastive_loss(c1.cpu(), i1.cpu(), batch["id"].cpu()).item()
    assert abs(loss - ref) < 1e-4, (loss, ref)
    assert abs(float(lm["softmax_temp"]) - 0.1) < 1e-6
    print(f"C-base B={B}, V={V}: loss {loss:.5f} (oracle on the same embeddings {ref:.5f}); targets stable under permutation: "
          f"{(tg(ot4) == t1[perm]).float().mean().item():.4f}")


def test_global_batch_2048_on_one_gpu_equals_eight_shards():
    """Maximum size of the path: the GLOBAL batch of BASELINE.json configs[3] (2048 pairs, 10 s) on ONE GPU -- what `bench.py --global-batch 2048` runs at
    N = 1, ~175 GB of activations and workspaces, conv-layer operands of 3.4e10 elements (67 GB: far beyond every 32-bit offset) -- against the same
    utterances run as eight B = 256 shards (what eight ranks would compute): the embeddings agree (the GEMM's fp32 summation order moves with a row's
    tile, so not bitwise), and the loss on the global batch is the fp32 oracle's InfoNCE on those embeddings (duplicate ids: the false-negative mask at
    Bg = 2048, which the reference's MAX_EYE = 256 cannot represent, losses.py:126)."""
    import bench
    from oracle.speechclip_ref import masked_contrastive_loss
    free, total = torch.cuda.mem_get_info()
    if total < 250e9:
        pytest.skip("needs the 288 GB of an MI355X")
    model = bench.build_model().cuda()
    Bg, Bl, L = 2048, 256, 160000
    g = torch.Generator().manual_seed(2048)
    wav = torch.empty(Bg, L)
    for s0 in range(0, Bg, 256):
        wav[s0:s0 + 256] = 0.1 * torch.randn(256, L, generator=g)
    for i in range(5, Bg, 97):                                  # some shorter utterances
        wav[i, 64000 + 31 * i:] = 0
    lens = torch.full((Bg,), L)
    for i in range(5, Bg, 97):
        lens[i] = 64000 + 31 * i
    img = torch.randn(Bg, 3, 224, 224, generator=g)
    ids = torch.arange(Bg) // 5                                 # 5 captions per image
    batch = {"wav": wav.cuda(), "wav_len": lens, "image": img.cuda(), "id": ids.cuda()}
    with torch.no_grad():
        a_sh, i_sh = [], []
        for r in range(Bg // Bl):                               # the eight shards first (small workspaces), then the global batch
            sl = slice(r * Bl, (r + 1) * Bl)
            lf, _, _ = model({k: v[sl] for k, v in batch.items()})
            a_sh.append(lf["parallel_audio_feat"].clone())
            i_sh.append(lf["image_feat"].clone())
        a_sh, i_sh = torch.cat(a_sh), torch.cat(i_sh)
        lf, _, _ = _with_env("SC_VARLEN_PACK", "0", lambda: model(batch))     # the padded layout: M = 2048 x 500 rows in every GEMM
        a_g, i_g = lf["parallel_audio_feat"], lf["image_feat"]
        loss = model.compute_loss(lf)["loss"].item()
    assert a_g.shape == (Bg, 512) and torch.isfinite(a_g).all() and torch.isfinite(i_g).all()
    # shards differ from the global run in their padded length only where a shard has no full-length utterance (none here: every shard keeps 10 s ones)
    ca, ci = _cos_rows(a_g, a_sh), _cos_rows(i_g, i_sh)
    assert ca.min().item() > 0.99999 and ci.min().item() > 0.99999, (ca.min().item(), ci.min().item())
    assert (a_g - a_sh).abs().max().item() < 2e-3
    ref = masked_contrastive_loss(a_g.float().cpu(), i_g.float().cpu(), ids).item()
    assert abs(loss - ref) < 1e-4, (loss, ref)
    assert abs(loss - 7.6) < 0.3                                # ~ ln 2048 for random-init towers
    print(f"Bg = 2048 on one GPU: loss {loss:.5f} (oracle on the same embeddings {ref:.5f}); min cos vs eight B = 256 shards: audio {ca.min().item():.7f} image {ci.min().item():.7f}; "
          f"peak memory {torch.cuda.max_memory_allocated() / 1e9:.0f} GB")

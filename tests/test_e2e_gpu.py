"""GPU end-to-end parity: the `avssl`-surface model (HIP kernels, bf16) against
  * the golden vectors produced by the reference's own glue (tests/golden/e2e_*.npz), and
  * the fp32 CPU oracle at the real base dimensions with shared random weights.
Tolerances, bf16 GPU vs fp32 oracle: hidden states cosine >= 0.998-0.999 per utterance (they differ between
utterances: cos 0.72); final embeddings in CENTRED cosine (helpers.centred_cos: the batch-mean component, which makes the raw cosine of
different utterances 0.988-0.9986, is removed) with the rotated-rows negative control; |loss diff| <= 2e-2."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import assert_rows_match, make_config

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _cos(a, b):
    return F.cosine_similarity(a.float().cpu().reshape(a.shape[0], -1), b.float().cpu().reshape(b.shape[0], -1), dim=-1)


def _tiny_cfgs(large=False):
    from oracle.clip_ref import ClipRefConfig
    from oracle.hubert_ref import HubertRefConfig
    from speechclip_amd.module.clip_model import ClipConfig
    from speechclip_amd.module.hubert import HubertConfig
    import dataclasses
    h = HubertRefConfig.tiny(layer_norm_first=large, extractor_mode="layer_norm" if large else "default", conv_bias=large)
    c = ClipRefConfig.tiny()
    return HubertConfig(**dataclasses.asdict(h)), ClipConfig(**dataclasses.asdict(c))


def _load_model(tag, large=False, cascaded=False, vocab_path=None):
    from speechclip_amd.model import KWClip_GeneralTransformer
    g = np.load(os.path.join(GOLD, f"e2e_{tag}.npz"))
    hc, cc = _tiny_cfgs(large)
    cfg = make_config(d_model=128, branch_heads=4, parallel=not cascaded, cascaded=cascaded, hubert_config=hc, clip_config=cc,
                      hubert_name="hubert_large_ll60k" if large else "hubert", normalize_hiddenstates=large, reduce_vocab=vocab_path)
    model = KWClip_GeneralTransformer(cfg)
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith(("criterion.", "cascaded_branch.clip.")) or "vector_quantizer" in k or k.endswith("mask_emb") for k in missing), missing
    model = model.cuda().eval()
    batch = {k: torch.from_numpy(g[k]).cuda() for k in ("wav", "wav_len", "image", "id")}
    return g, model, batch


@pytest.mark.parametrize("tag,large", [("tiny_base_p", False), ("tiny_large_p", True)])
def test_tiny_parallel_vs_reference_glue(tag, large):
    g, model, batch = _load_model(tag, large)
    with torch.no_grad():
        audio_feat, audio_len, hidden = model.forward_audio(batch["wav"], batch["wav_len"], return_hidden_states=True)
        loss_feats, log_metrics, others = model(batch)
        loss = model.compute_loss(loss_feats)["loss"].item()
    assert np.array_equal(audio_len.cpu().numpy(), g["feat_len"])
    T = g["audio_feat"].shape[1]
    assert audio_feat.shape[1] == T
    for b, L in enumerate(g["feat_len"]):      # frames the head can see
        assert _cos(hidden[0][b:b + 1, :L], torch.from_numpy(g["hidden_0"])[b:b + 1, :L]).item() > 0.999
        assert _cos(hidden[-1][b:b + 1, :L], torch.from_numpy(g["hidden_last"])[b:b + 1, :L]).item() > 0.998
        assert _cos(audio_feat[b:b + 1, :L], torch.from_numpy(g["audio_feat"])[b:b + 1, :L]).item() > 0.998
    assert_rows_match(loss_feats["image_feat"], torch.from_numpy(g["image_feat"]), 0.99, "image_feat")
    cc = assert_rows_match(loss_feats["parallel_audio_feat"], torch.from_numpy(g["parallel_audio_feat"]), 0.99, "parallel_audio_feat")
    print(tag, "parallel_audio_feat centred cosine per row:", cc.tolist())
    assert abs(loss - float(g["loss"])) < 2e-2, (loss, float(g["loss"]))
    assert abs(log_metrics["cl_temp"] - 1 / 0.07) < 1e-4


@pytest.mark.parametrize("tag,large", [("tiny_base_p", False), ("tiny_large_p", True)])
@pytest.mark.parametrize("method", ["method1", "method2"])
def test_normalize_type_methods_vs_reference_glue(tag, large, method):
    """audio_encoder.normalize_hiddenstates: true with normalize_type method1 / method2 (speech_encoder_plus.py:572-592): sc_hidden_normalize overwrites the
    stacked hidden states in place before they are mixed / returned; against what the reference's own encoder forward returned on the same weights and
    ragged waves (tests/golden/norm_methods.npz): the normalised states (hidden_0, hidden_last) and the mixed frames, per utterance over its own frames."""
    from speechclip_amd.model import KWClip_GeneralTransformer
    g = np.load(os.path.join(GOLD, f"e2e_{tag}.npz"))
    n = np.load(os.path.join(GOLD, "norm_methods.npz"))
    hc, cc = _tiny_cfgs(large)
    cfg = make_config(d_model=128, branch_heads=4, hubert_config=hc, clip_config=cc, hubert_name="hubert_large_ll60k" if large else "hubert",
                      normalize_hiddenstates=True)
    cfg.audio_encoder.normalize_type = method
    model = KWClip_GeneralTransformer(cfg)
    assert model.audio_encoder.weightedsum_layer.normalize_features is False
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")}
    model.load_state_dict(sd, strict=False)
    model = model.cuda().eval()
    wav, wav_len = torch.from_numpy(g["wav"]).cuda(), torch.from_numpy(g["wav_len"]).cuda()
    with torch.no_grad():
        feat, flen, hidden = model.forward_audio(wav, wav_len, return_hidden_states=True)
    assert np.array_equal(flen.cpu().numpy(), n[f"{tag}/{method}/feat_len"])
    for b, L in enumerate(n[f"{tag}/{method}/feat_len"]):
        for got, key in ((hidden[0], "hidden_0"), (hidden[-1], "hidden_last"), (feat, "feat")):
            ref = torch.from_numpy(n[f"{tag}/{method}/{key}"])
            assert _cos(got[b:b + 1, :L], ref[b:b + 1, :L]).item() > 0.998, (key, b)
            # the SCALE is what these methods set: unit frames (method1) / unit mean frame norm per utterance (method2)
            rn, gn = ref[b, :L].norm(dim=-1).mean().item(), got[b, :L].float().norm(dim=-1).mean().item()
            assert abs(gn - rn) < 2e-2 * rn, (key, b, gn, rn)
    if method == "method1":
        torch.testing.assert_close(hidden[-1].float().norm(dim=-1), torch.ones_like(hidden[-1][..., 0]).float(), atol=1e-2, rtol=0)


def test_tiny_cascaded_vs_reference_glue(tmp_path):
    vocab = np.array([0, 320, 510, 511] + list(range(5, 300, 3)))
    vp = str(tmp_path / "vocab.npy")
    np.save(vp, np.stack([vocab, np.arange(len(vocab))[::-1] + 1], axis=1))
    g, model, batch = _load_model("tiny_base_c", cascaded=True, vocab_path=vp)
    with torch.no_grad():
        loss_feats, log_metrics, others = model(batch)
        loss = model.compute_loss(loss_feats)["loss"].item()
    tg = others["vq_results"]["targets"].cpu().numpy()
    agree = (tg == g["vq_targets"]).mean()
    # Random sub-word tables: the reference's own top-1 / top-2 margins are ~1e-3, so a few arg-maxes flip under the bf16 towers.  The
    # embedding / loss of this configuration are asserted -- unconditionally -- on the decisive-margin fixture (test_cascaded_gpu.py:
    # test_decisive_cascaded_fixture_vs_reference_glue) and with the towers out of the picture (test_cascaded_head_isolated_on_oracle_frames).
    assert agree >= 0.9, agree
    assert np.isfinite(loss)
    np.testing.assert_allclose(others["vq_results"]["ent_per_t"].cpu().numpy(), g["vq_ent_per_t"], rtol=2e-2, atol=2e-2)
    assert abs(log_metrics["softmax_temp"] - 0.1) < 1e-6


def test_base_dims_vs_oracle():
    """Real P-base dimensions (HuBERT-base + ViT-B/32, 8-head branch), B=3, mixed lengths, shared seeded weights."""
    from oracle.clip_ref import ClipRefConfig
    from oracle.hubert_ref import HubertRefConfig
    from oracle.speechclip_ref import SpeechClipRef
    from speechclip_amd.model import KWClip_GeneralTransformer
    torch.manual_seed(1234)
    model = KWClip_GeneralTransformer(make_config()).eval()
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        model.audio_encoder.weightedsum_layer.weights.copy_(0.5 * torch.randn(13, generator=g))
        for m in model.modules():              # non-trivial norm affines / biases
            if isinstance(m, (torch.nn.LayerNorm, torch.nn.GroupNorm)):
                m.weight.add_(0.1 * torch.randn(m.weight.shape, generator=g))
                m.bias.add_(0.1 * torch.randn(m.bias.shape, generator=g))
    ref = SpeechClipRef(HubertRefConfig.base(), ClipRefConfig.vit_b32(), parallel=True, branch_heads=8).eval()
    sd = model.state_dict()
    ref.encoder.load_state_dict({k[len("audio_encoder.encoder."):]: v for k, v in sd.items() if k.startswith("audio_encoder.encoder.")})
    ref.clip.load_state_dict({k[len("clip.model."):]: v for k, v in sd.items() if k.startswith("clip.model.")})
    ref.parallel_branch.load_state_dict({k[len("parallel_branch."):]: v for k, v in sd.items() if k.startswith("parallel_branch.")})
    with torch.no_grad():
        ref.ws_weights.copy_(sd["audio_encoder.weightedsum_layer.weights"])
    lens = [32000, 20001, 9600]
    wav = torch.zeros(3, max(lens))
    for i, l in enumerate(lens):
        wav[i, :l] = 0.1 * torch.randn(l, generator=g)
    batch = {"wav": wav, "wav_len": torch.tensor(lens), "image": torch.randn(3, 3, 224, 224, generator=g), "id": torch.tensor([4, 4, 9])}
    o = ref(batch)
    ref_loss = ref.compute_loss(o)["loss"].item()
    model = model.cuda()
    with torch.no_grad():
        lf, _, _ = model({k: v.cuda() for k, v in batch.items()})
        loss = model.compute_loss(lf)["loss"].item()
    assert_rows_match(lf["image_feat"], o["image_feat"], 0.99, "image_feat")
    cc = assert_rows_match(lf["parallel_audio_feat"], o["parallel_audio_feat"], 0.99, "parallel_audio_feat")
    print("base dims: parallel_audio_feat centred cosine per row:", cc.tolist(), " raw cosine between different utterances (oracle):",
          _cos(o["parallel_audio_feat"][:1], o["parallel_audio_feat"][1:2]).item())
    logit_err = ((lf["parallel_audio_feat"].cpu() @ lf["image_feat"].cpu().t() - o["parallel_audio_feat"] @ o["image_feat"].t()) / 0.07).abs().max().item()
    assert logit_err < 5e-2, logit_err
    assert abs(loss - ref_loss) < 2e-2, (loss, ref_loss)


def _share_weights(model, ref, parallel=True, cascaded=False):
    sd = model.state_dict()
    ref.encoder.load_state_dict({k[len("audio_encoder.encoder."):]: v for k, v in sd.items() if k.startswith("audio_encoder.encoder.")})
    ref.clip.load_state_dict({k[len("clip.model."):]: v for k, v in sd.items() if k.startswith("clip.model.")})
    with torch.no_grad():
        ref.ws_weights.copy_(sd["audio_encoder.weightedsum_layer.weights"])
    if parallel:
        ref.parallel_branch.load_state_dict({k[len("parallel_branch."):]: v for k, v in sd.items() if k.startswith("parallel_branch.")})
    if cascaded:
        ref.cascaded_branch.load_state_dict({k[len("cascaded_branch."):]: v for k, v in sd.items()
                                             if k.startswith("cascaded_branch.") and not k.startswith("cascaded_branch.clip.")
                                             and "vector_quantizer" not in k})


def test_large_dims_vs_oracle():
    """BASELINE config 5 shapes: HuBERT-large (pre-LN, LN conv stack, wave layer-norm, normalised layer mix) + CLIP ViT-L/14,
    trainable temperature, B=2 mixed lengths."""
    from oracle.clip_ref import ClipRefConfig
    from oracle.hubert_ref import HubertRefConfig
    from oracle.speechclip_ref import SpeechClipRef
    from speechclip_amd.model import KWClip_GeneralTransformer
    torch.manual_seed(4321)
    cfg = make_config(d_model=1024, branch_heads=8, hubert_name="hubert_large_ll60k", clip_name="ViT-L/14", normalize_hiddenstates=True,
                      temperature_trainable=True)
    model = KWClip_GeneralTransformer(cfg).eval()
    g = torch.Generator().manual_seed(6)
    with torch.no_grad():
        model.audio_encoder.weightedsum_layer.weights.copy_(0.5 * torch.randn(25, generator=g))
    ref = SpeechClipRef(HubertRefConfig.large(), ClipRefConfig.vit_l14(), parallel=True, branch_heads=8, normalize_hiddenstates=True,
                        inv_temperature=model.criterion.current_temperature).eval()
    _share_weights(model, ref)
    lens = [24000, 17777]
    wav = torch.zeros(2, max(lens))
    for i, l in enumerate(lens):
        wav[i, :l] = 0.1 * torch.randn(l, generator=g) + 0.01
    batch = {"wav": wav, "wav_len": torch.tensor(lens), "image": torch.randn(2, 3, 224, 224, generator=g), "id": torch.tensor([1, 2])}
    o = ref(batch)
    ref_loss = ref.compute_loss(o)["loss"].item()
    model = model.cuda()
    with torch.no_grad():
        lf, lm, _ = model({k: v.cuda() for k, v in batch.items()})
        loss = model.compute_loss(lf)["loss"].item()
    assert lf["image_feat"].shape == (2, 768) and lf["parallel_audio_feat"].shape == (2, 768)
    assert_rows_match(lf["image_feat"], o["image_feat"], 0.99, "image_feat")
    cc = assert_rows_match(lf["parallel_audio_feat"], o["parallel_audio_feat"], 0.99, "parallel_audio_feat")
    print("large dims: parallel_audio_feat centred cosine per row:", cc.tolist())
    assert abs(loss - ref_loss) < 2e-2, (loss, ref_loss)
    assert abs(lm["cl_temp"] - 1 / 0.07) < 1e-3


def test_cascaded_base_dims_vs_oracle(tmp_path):
    """BASELINE config 3 shapes: cascaded base (8 keywords, 1-head attention, BatchNorm, cosine vs a reduced 8112-word
    vocabulary, hard VQ, CLIP text tower)."""
    from oracle.clip_ref import ClipRefConfig
    from oracle.hubert_ref import HubertRefConfig
    from oracle.speechclip_ref import SpeechClipRef
    from speechclip_amd.model import KWClip_GeneralTransformer
    g = torch.Generator().manual_seed(8)
    vocab = torch.cat([torch.tensor([0, 320, 49406, 49407]), torch.randperm(49000, generator=g)[:8108] + 321]).numpy()
    vp = str(tmp_path / "vocab.npy")
    np.save(vp, np.stack([vocab, np.arange(len(vocab))[::-1] + 1], axis=1))
    torch.manual_seed(99)
    model = KWClip_GeneralTransformer(make_config(parallel=False, cascaded=True, reduce_vocab=vp)).eval()
    with torch.no_grad():
        model.audio_encoder.weightedsum_layer.weights.copy_(0.5 * torch.randn(13, generator=g))
        bn = model.cascaded_branch.bn_layer.bn_layer
        bn.running_mean.copy_(0.05 * torch.randn(bn.running_mean.shape, generator=g))
        bn.running_var.copy_(1.0 + 0.2 * torch.rand(bn.running_var.shape, generator=g))
    ref = SpeechClipRef(HubertRefConfig.base(), ClipRefConfig.vit_b32(), parallel=False, cascaded=True,
                        reduced_vocab=torch.from_numpy(vocab)).eval()
    _share_weights(model, ref, parallel=False, cascaded=True)
    lens = [16000, 12000, 8000]
    wav = torch.zeros(3, max(lens))
    for i, l in enumerate(lens):
        wav[i, :l] = 0.1 * torch.randn(l, generator=g)
    batch = {"wav": wav, "wav_len": torch.tensor(lens), "image": torch.randn(3, 3, 224, 224, generator=g), "id": torch.tensor([5, 6, 6])}
    o = ref(batch)
    model = model.cuda()
    with torch.no_grad():
        lf, lm, others = model({k: v.cuda() for k, v in batch.items()})
    agree = (others["vq_results"]["targets"].cpu() == o["vq_results"]["targets"]).float().mean().item()
    assert agree >= 0.85, agree                    # arg-max over 8112 near-tied random embeddings; bf16 upstream features
    assert others["vq_results"]["subword_prob"].shape == (3, 8, 8112) and others["keywords"].shape == (3, 8, 512)
    # (embedding / loss of the cascaded head: asserted unconditionally in test_cascaded_gpu.py, where the arg-max is decisive)
    np.testing.assert_allclose(others["vq_results"]["ent_per_t"].cpu().numpy(), o["vq_results"]["ent_per_t"].numpy(), rtol=2e-2)
    assert abs(float(others["vq_results"]["prob_perplexity"]) - float(o["vq_results"]["prob_perplexity"])) / float(o["vq_results"]["prob_perplexity"]) < 2e-2


def test_full_size_properties():
    """BASELINE config-2 shape (10 s audio, 224^2 images, P-base) at B = 64, where the fp32 oracle is too slow to run:
    size-independent properties instead -- bitwise run-to-run determinism, batch-permutation equivariance of the embeddings,
    unit norms, and the HIP InfoNCE agreeing with the fp32 oracle loss evaluated on the SAME embeddings."""
    from oracle.speechclip_ref import masked_contrastive_loss
    from speechclip_amd.model import KWClip_GeneralTransformer
    torch.manual_seed(7)
    model = KWClip_GeneralTransformer(make_config()).eval().cuda()
    g = torch.Generator().manual_seed(11)
    B, L = 64, 160000
    lens = torch.full((B,), L)
    lens[::7] = 96000                                          # a few shorter utterances: exercises the key masks at full T
    wav = 0.1 * torch.randn(B, L, generator=g)
    for i in range(B):
        wav[i, lens[i]:] = 0
    batch = {"wav": wav.cuda(), "wav_len": lens.cuda(), "image": torch.randn(B, 3, 224, 224, generator=g).cuda(),
             "id": torch.arange(B).cuda() // 2}
    with torch.no_grad():
        lf1, _, _ = model(batch)
        a1, i1 = lf1["parallel_audio_feat"].clone(), lf1["image_feat"].clone()
        lf2, _, _ = model(batch)
        assert torch.equal(a1, lf2["parallel_audio_feat"]) and torch.equal(i1, lf2["image_feat"])      # deterministic
        perm = torch.randperm(B, generator=g).cuda()
        lf3, _, _ = model({k: v[perm] for k, v in batch.items()})
        loss = model.compute_loss(lf1)["loss"].item()
        loss_perm = model.compute_loss(lf3)["loss"].item()
    assert torch.isfinite(a1).all() and torch.isfinite(i1).all()
    torch.testing.assert_close(a1.norm(dim=-1), torch.ones(B, device="cuda"), atol=1e-5, rtol=0)
    torch.testing.assert_close(i1.norm(dim=-1), torch.ones(B, device="cuda"), atol=1e-5, rtol=0)
    # no cross-sample coupling except the shared batch max length (unchanged).  Not bitwise: the GEMM rotates its K loop by the
    # tile's M-panel index, so the fp32 summation order of a row depends (deterministically) on its position in the batch.
    assert _cos(lf3["image_feat"], i1[perm]).min().item() > 0.99999
    assert _cos(lf3["parallel_audio_feat"], a1[perm]).min().item() > 0.99999
    ref = masked_contrastive_loss(a1.cpu(), i1.cpu(), batch["id"].cpu()).item()
    assert abs(loss - ref) < 1e-4 and abs(loss_perm - loss) < 1e-4


def test_rccl_packed_gather_on_one_gpu():
    """The exchange step over RCCL (backend "nccl") on the real device: process group of size 1, packed all-gather of the features with the
    int64 ids bit-cast into two fp32 lanes, plus the flat-gradient all-reduce of FusedAdam.  (N > 1 is covered by the gloo tests on CPU.)"""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import os, sys, torch, torch.distributed as dist
        sys.path.insert(0, os.getcwd())
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
        from speechclip_amd import parallel
        g = torch.Generator().manual_seed(0)
        feats = {"id": torch.tensor([7, 2 ** 40 + 5, -3, 11]).cuda(), "image_feat": torch.randn(4, 512, generator=g).cuda(),
                 "parallel_audio_feat": torch.randn(4, 512, generator=g).cuda()}
        out = parallel.gather_loss_feats(feats, force=True)
        assert all(torch.equal(out[k], feats[k]) for k in feats), "packed gather changed the payload"
        assert out["id"].dtype == torch.int64
        t = torch.arange(1000, device="cuda", dtype=torch.float32)
        dist.all_reduce(t)
        assert torch.equal(t, torch.arange(1000, device="cuda", dtype=torch.float32))
        dist.destroy_process_group()
        print("RCCL_OK")
    """)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=os.path.dirname(os.path.dirname(__file__)))
    assert "RCCL_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_validation_hooks_recall_vs_oracle_pipeline():
    """Lightning's validation protocol end to end on the device (validation_step -> validation_step_end -> validation_epoch_end ->
    mutualRetrieval through sc_sgemm + sc_retrieval_ranks) against the oracle's features + argsort-based recall on the same batches:
    the score matrices must agree closely and the hooks' recall@K must be the argsort recall of their own score matrix.  (Recall parity
    against the oracle: tests/test_recall_gpu.py, a trained discriminative 1000-pair set.)"""
    from oracle.speechclip_ref import SpeechClipRef, mutual_retrieval
    from oracle.clip_ref import ClipRefConfig
    from oracle.hubert_ref import HubertRefConfig
    g, model, _ = _load_model("tiny_base_p", False)
    ref = SpeechClipRef(HubertRefConfig.tiny(), ClipRefConfig.tiny(), parallel=True, branch_heads=4).eval()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ref.encoder.load_state_dict({k[len("audio_encoder.encoder."):]: v for k, v in sd.items() if k.startswith("audio_encoder.encoder.")})
    ref.clip.load_state_dict({k[len("clip.model."):]: v for k, v in sd.items() if k.startswith("clip.model.")})
    ref.parallel_branch.load_state_dict({k[len("parallel_branch."):]: v for k, v in sd.items() if k.startswith("parallel_branch.")})
    with torch.no_grad():
        ref.ws_weights.copy_(sd["audio_encoder.weightedsum_layer.weights"])
    gen = torch.Generator().manual_seed(31)
    res = ClipRefConfig.tiny().image_resolution
    n_img, cap = 6, 2
    images = torch.randn(n_img, 3, res, res, generator=gen)
    batches, oa, oi, oid = [], [], [], []
    for bi in range(3):                                   # 3 batches x 4 captions; image j is paired with captions 2j, 2j+1
        ids = torch.tensor([(bi * 4 + k) // cap for k in range(4)])
        lens = [int(x) for x in torch.randint(3000, 8000, (4,), generator=gen)]
        wav = torch.zeros(4, max(lens))
        for i, l in enumerate(lens):
            wav[i, :l] = 0.3 * torch.randn(l, generator=gen)
        b = {"wav": wav, "wav_len": torch.tensor(lens), "image": images[ids], "id": ids}
        batches.append(b)
        o = ref(b)
        oa.append(o["parallel_audio_feat"]); oi.append(o["image_feat"]); oid.append(ids)
    outs = []
    with torch.no_grad():
        for i, b in enumerate(batches):
            outs.append(model.validation_step_end(model.validation_step({k: v.cuda() for k, v in b.items()}, i)))
        r_ab, r_ba, r_mean = model.validation_epoch_end(outs)
    # oracle pipeline (dedupe images by id exactly as validation_epoch_end does: last occurrence wins, first-seen order)
    all_ids = torch.cat(oid); all_img = torch.cat(oi); all_aud = torch.cat(oa)
    first = {}
    for i, _id in enumerate(all_ids.tolist()):
        first[_id] = i
    img_ids = torch.tensor(list(first.keys()))
    img_f = all_img[torch.tensor(list(first.values()))]
    score_ref = all_aud @ img_f.t()
    o_ab, o_ba, o_mean = mutual_retrieval(score_ref, score_ref.t().contiguous(), all_ids, img_ids, [1, 5, 10])
    # device scores recomputed from the hook outputs for the comparison of the matrices themselves
    aud_d = torch.cat([x["audio_feat"] for x in outs]).float()
    img_d = torch.cat([x["image_feat"] for x in outs]).float()[torch.tensor(list(first.values()))]
    score_dev = aud_d @ img_d.t()
    tol = (score_dev - score_ref).abs().max().item()
    assert tol < 1e-2, tol
    # This 12 x 6 random-init set is NOT discriminative (the whole score matrix spans ~0.1, adjacent candidates sit 1e-3 apart), so it pins the
    # plumbing only: the hooks' recall must equal an argsort recall of the hooks' OWN scores exactly, and the scores must match the oracle's.
    # Recall@K parity against the oracle is asserted on the trained 1000-pair set of tests/test_recall_gpu.py (+-0.5 pt, identical top-1
    # on margin > 0.05, no flip allowance).
    d_ab, d_ba, _ = mutual_retrieval(score_dev, score_dev.t().contiguous(), all_ids, img_ids, [1, 5, 10])
    for k in ("recall@1", "recall@5", "recall@10"):
        assert abs(r_ab[k] - d_ab[k]) < 1e-4 and abs(r_ba[k] - d_ba[k]) < 1e-4, (k, r_ab[k], d_ab[k], r_ba[k], d_ba[k])
    assert r_ab["recall@10"] == 100.0 and r_ba["recall@10"] == 100.0       # 6 images: everything is in the top 10


def test_samples_beyond_wav_len_are_ignored():
    """speech_encoder_plus.py:520-534 slices every utterance to wav[:wav_len] and re-pads with zeros, so whatever the caller left beyond
    wav_len (collate garbage, a previous batch) cannot reach the features -- same here, bitwise, for base (GroupNorm extractor) and for the
    large layout (per-utterance wave LayerNorm over the valid samples only)."""
    for tag, large in (("tiny_base_p", False), ("tiny_large_p", True)):
        g, model, batch = _load_model(tag, large)
        lens = batch["wav_len"].clone()
        wav = batch["wav"].clone()
        dirty = wav.clone()
        for i in range(wav.shape[0]):
            dirty[i, int(lens[i]):] = 3.0 * torch.randn(wav.shape[1] - int(lens[i]), device=wav.device) + 1.0
        assert not bool((lens == wav.shape[1]).all()), "fixture has no padded utterance"
        with torch.no_grad():
            a, _, _ = model(batch)
            b, _, _ = model(dict(batch, wav=dirty))
        assert torch.equal(a["parallel_audio_feat"], b["parallel_audio_feat"]), tag


def test_step_is_hip_graph_capturable():
    """The whole forward + loss step (every kernel behind the C ABI, hipBLASLt included) can be captured in a HIP graph and replayed with
    new inputs in the static buffers, bitwise equal to eager execution (length uploads come from ops.dev_ints' cache after warm-up)."""
    g, model, batch = _load_model("tiny_base_p", False)
    batch = dict(batch, wav_len=batch["wav_len"].cpu())          # host lengths: the encoder's length arithmetic is host logic

    def step():
        with torch.no_grad():
            lf, _, _ = model(batch)
            return model.compute_loss(lf)["loss"], lf["parallel_audio_feat"]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        g_loss, g_feat = step()
    graph.replay()
    e_loss, e_feat = step()
    assert torch.equal(g_feat, e_feat) and float(g_loss) == float(e_loss)
    batch["wav"].mul_(0.5)                                       # new samples through the same buffers
    graph.replay()
    e_loss2, e_feat2 = step()
    assert torch.equal(g_feat, e_feat2) and float(g_loss) == float(e_loss2) and not torch.equal(e_feat, e_feat2)


def test_image_tower_on_side_stream_is_bitwise_equal_to_serial():
    """KWClip_GeneralTransformer.forward runs the frozen image tower on a side HIP stream beside the speech tower (default); serialising the
    towers (SC_OVERLAP_VIT=0 / the module switch) must give bitwise identical features and loss."""
    import speechclip_amd.model.kwClip as K
    g, model, batch = _load_model("tiny_base_p", False)
    assert K._OVERLAP_IMAGE_TOWER or os.environ.get("SC_OVERLAP_VIT") == "0"
    outs = []
    for flag in (True, False, True):
        K._OVERLAP_IMAGE_TOWER = flag
        try:
            with torch.no_grad():
                lf, _, _ = model(batch)
                loss = model.compute_loss(lf)["loss"]
            torch.cuda.synchronize()
            outs.append((lf["image_feat"].clone(), lf["parallel_audio_feat"].clone(), float(loss)))
        finally:
            K._OVERLAP_IMAGE_TOWER = True
    for o in outs[1:]:
        assert torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1]) and o[2] == outs[0][2]


def test_forward_is_bitwise_reproducible_with_the_towers_overlapped():
    """Twenty forwards of one P-base batch (B = 64, 10 s audio), image tower on its side stream beside the speech tower's front end: every embedding bit-equal to
    the first run's.  Round 6 found the image features of 1-3 images differing by ~3e-4 in 30-60 % of the steps: the packed-fp32 code hipcc formed from the ViT's
    two-rows-per-wave LayerNorm kernel (layernorm768f_kernel) returned last-place-different statistics for the first row of a wave's pair when the kernel shared the chip
    with conv0 / the conv stack / the positional conv -- never on an idle GPU, never with the towers serialised (tools/vit_race_probe.py, tools/determinism_probe.py,
    profiles/r06_vit_layernorm_nondeterminism.txt).  rowops.hip is compiled without SLP vectorisation since (csrc/Makefile).  One repeat
    (test_full_size_properties) caught it once in five suite runs; twenty catch a 30 % event with 99.9 %."""
    from speechclip_amd.model import KWClip_GeneralTransformer
    torch.manual_seed(7)
    model = KWClip_GeneralTransformer(make_config()).eval().cuda()
    g = torch.Generator().manual_seed(12)
    B, L = 64, 160000
    lens = torch.full((B,), L)
    lens[::5] = 120000
    wav = 0.1 * torch.randn(B, L, generator=g)
    for i in range(B):
        wav[i, lens[i]:] = 0
    batch = {"wav": wav.cuda(), "wav_len": lens.cuda(), "image": torch.randn(B, 3, 224, 224, generator=g).cuda(), "id": torch.arange(B).cuda()}
    with torch.no_grad():
        lf0, _, _ = model(batch)
        a0, i0 = lf0["parallel_audio_feat"].clone(), lf0["image_feat"].clone()
        for rep in range(20):
            lf, _, _ = model(batch)
            assert torch.equal(lf["image_feat"], i0), ("image_feat", rep, int((lf["image_feat"] != i0).any(1).sum()))
            assert torch.equal(lf["parallel_audio_feat"], a0), ("parallel_audio_feat", rep)

"""Cascaded branch (kwClip.py:857-916, kw_bn.py:100-131, my_vector_quantizer.py:64-165, clip_official.py:220-264):
every forward kernel against the CPU oracle in isolation, the whole head on oracle inputs, and the decisive-margin fixture produced by
the reference's own modules (tests/golden/e2e_tiny_base_c2.npz) -- no conditional asserts, no skips."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import assert_rows_match, centred_cos, make_config

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _g(seed):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("B,K,D", [(5, 8, 512), (3, 8, 64), (256, 8, 512), (2, 1, 16)])
def test_kw_affine_vs_oracle_kw_batchnorm_eval(B, K, D):
    """sc_kw_affine (eval-mode Kw_BatchNorm) vs the oracle's _KwBatchNorm in eval mode: running statistics, affine parameters and the
    (dim, keyword) flattening of kw_bn.py:122-131."""
    from oracle.speechclip_ref import _KwBatchNorm
    from speechclip_amd.module.speechclip_c_modules.kw_bn import Kw_BatchNorm
    g = _g(B * 1000 + D)
    init_bias, init_scale = 0.1 * torch.randn(D, generator=g), 0.5 + torch.rand(D, generator=g)
    ref = _KwBatchNorm(K, D, init_bias, init_scale).eval()
    with torch.no_grad():
        ref.bn_layer.running_mean.copy_(0.3 * torch.randn(K * D, generator=g))
        ref.bn_layer.running_var.copy_(0.2 + torch.rand(K * D, generator=g))
        ref.bn_layer.weight.mul_(1 + 0.2 * torch.randn(K * D, generator=g))
        ref.bn_layer.bias.add_(0.2 * torch.randn(K * D, generator=g))
    ours = Kw_BatchNorm(K, D, "eachKw", init_bias, init_scale, 1.0, True, True).eval()
    ours.load_state_dict(ref.state_dict())
    ours = ours.cuda()
    x = 2 * torch.randn(B, K, D, generator=g) + 0.3
    with torch.no_grad():
        want = ref(x)
        got = ours(x.cuda())
    torch.testing.assert_close(got.cpu(), want, atol=2e-5, rtol=2e-5)


@pytest.mark.parametrize("R,V,E,exact", [(24, 8112, 512, True), (24, 8112, 512, None), (64, 103, 64, True), (7, 333, 32, None),
                                          (1024, 8112, 512, None), (256, 49408, 512, None)])
def test_cosine_scores_vs_fp32_cosine_similarity(R, V, E, exact):
    """sc_cosine_scores (fp32 SIMT) and the default dispatch (MFMA three-term split + sc_cosine_refine for large problems) against
    F.cosine_similarity in fp32 on the CPU, the reference's own expression (kwClip.py:889-897): values, and the arg-max on every row whose
    fp32 margin exceeds the fp32 round-off of the score itself."""
    from speechclip_amd import ops
    g = _g(R + V + E)
    a = torch.randn(R, E, generator=g) * (0.5 + torch.rand(R, 1, generator=g))
    emb = 0.02 * torch.randn(V, E, generator=g)
    emb[5] = 0                                                  # a zero row: the eps clamp of cosine_similarity
    a[1] = 3.0 * emb[17]                                        # an exact match
    if V > 200:                                                 # planted near-ties: two rows 1e-6 apart in cosine
        emb[101] = emb[100] * 1.5 + 1e-4 * torch.randn(E, generator=g) * emb[100].norm() / E ** 0.5
        a[2] = emb[100] + emb[101]
    want = torch.empty(R, V)
    for r0 in range(0, R, 64):                                  # chunked: [64, V, E] fp32 temporaries
        want[r0:r0 + 64] = F.cosine_similarity(a[r0:r0 + 64, None, :], emb[None, :, :], dim=-1)
    got = ops.cosine_scores(a.cuda(), emb.cuda(), exact=exact).cpu()
    big = (exact is None) and (E % 64 == 0 and V % 4 == 0 and R * V >= (1 << 22))
    tol = 2e-5 if big else 2e-6                                 # the MFMA path is ~1e-5 accurate away from the row maximum
    assert (got - want).abs().max().item() < tol, (got - want).abs().max().item()
    top2 = want.topk(2, dim=-1)
    decisive = (top2.values[:, 0] - top2.values[:, 1]) > 4e-7
    assert decisive.float().mean().item() > 0.8                 # (all rows but the planted near-tie)
    assert torch.equal(got.argmax(-1)[decisive], top2.indices[decisive, 0])
    # near the maximum the values themselves are fp32-accurate on both paths (the refine pass recomputes them)
    near = want >= (top2.values[:, :1] - 5e-4)
    assert (got - want)[near].abs().max().item() < 2e-6


@pytest.mark.parametrize("B,K,V", [(4, 8, 103), (3, 8, 8112), (16, 8, 49408), (1, 1, 7)])
def test_vq_fwd_vs_oracle_simple_vq(B, K, V):
    """sc_vq_fwd vs oracle.simple_vq (my_vector_quantizer.py:64-165, hard / non-gumbel): targets, both perplexities, per-keyword entropy,
    diversity loss; special ids 0/2/3 masked even when they hold the row maximum."""
    from oracle.speechclip_ref import simple_vq
    from speechclip_amd.module.speechclip_c_modules.vector_quantizers import SimpleVectorQuantizer
    g = _g(B + V)
    cos = (0.3 * torch.randn(B, K, V, generator=g)).clamp(-1, 1)
    cos[0, 0, 0] = 0.99                                         # masked ids carry the maximum
    if V > 3:
        cos[-1, -1, 3] = 0.98
        cos[0, -1, 2] = 0.97
    ref = simple_vq(cos, 0.1, training=False)
    vq = SimpleVectorQuantizer(temp="fixed=0.1", time_first=True, use_gumbel=False, hard=True).cuda().eval()
    out = vq(x=cos.cuda())
    assert torch.equal(out["targets"].cpu(), ref["targets"])
    assert out["targets"].dtype == torch.int64 and out["targets"].shape == (B, K, 1)
    for k in ("code_perplexity", "prob_perplexity", "diversity_loss"):
        assert abs(float(out[k]) - float(ref[k])) <= 2e-4 * max(1.0, abs(float(ref[k]))), (k, float(out[k]), float(ref[k]))
    torch.testing.assert_close(out["ent_per_t"].cpu(), ref["ent_per_t"], atol=1e-4, rtol=1e-4)
    assert out["num_vars"] == V and out["temp"] == 0.1
    assert torch.equal(out["subword_prob"].cpu(), ref["subword_prob"])


def test_gather_rows_is_exact_one_hot_matmul():
    """sc_gather_rows = `subword_prob @ token_embedding.weight` for a hard one-hot (kwClip.py:909): bitwise the selected rows."""
    from speechclip_amd import ops
    g = _g(3)
    emb = torch.randn(8112, 512, generator=g)
    idx = torch.randint(0, 8112, (2048,), generator=g)
    idx[:3] = torch.tensor([0, 8111, 8111])
    got = ops.gather_rows(emb.cuda(), idx.cuda()).cpu()
    assert torch.equal(got, emb[idx])
    onehot = torch.zeros(64, 8112).scatter_(-1, idx[:64, None], 1.0)
    assert torch.equal(got[:64], onehot @ emb)


@pytest.mark.parametrize("dims", ["tiny", "vit_b32"])
def test_encode_keywords_vs_oracle(dims, tmp_path):
    """ClipModel.encode_keywords ([SOT, kw_1..K, EOT] through the causal text tower, K + 2 live positions) vs oracle.encode_keywords (all
    77 positions, clip_official.py:220-264), shared weights, with a reduced vocabulary so SOT / EOT are re-mapped ids."""
    from oracle.clip_ref import ClipRef, ClipRefConfig
    from oracle.speechclip_ref import encode_keywords
    from speechclip_amd.module import ClipModel
    from speechclip_amd.module.clip_model import ClipConfig
    import dataclasses
    g = _g(17)
    rc = ClipRefConfig.tiny() if dims == "tiny" else ClipRefConfig.vit_b32()
    V = rc.vocab_size
    vocab = torch.cat([torch.tensor([0, 320, V - 2, V - 1]), torch.randperm(V - 400, generator=g)[:96] + 321]).numpy()
    vp = str(tmp_path / "vocab.npy")
    np.save(vp, np.stack([vocab, np.arange(len(vocab))[::-1] + 1], axis=1))
    torch.manual_seed(5)
    ours = ClipModel("ViT-B/32", reduce_subword_embbedding=vp, clip_config=ClipConfig(**dataclasses.asdict(rc)))
    ref = ClipRef(rc).eval()
    sd = {k: v for k, v in ours.model.state_dict().items()}
    ref.token_embedding = torch.nn.Embedding(*sd["token_embedding.weight"].shape)
    ref.load_state_dict(sd)
    with torch.no_grad():                                      # non-trivial norms so the positions matter
        for m in list(ours.model.modules()):
            if isinstance(m, torch.nn.LayerNorm):
                m.weight.add_(0.1 * torch.randn(m.weight.shape, generator=g)); m.bias.add_(0.1 * torch.randn(m.bias.shape, generator=g))
        ref.load_state_dict(ours.model.state_dict())
    B, K, W = 6, 8, rc.text_width
    emb = ours.model.token_embedding.weight
    kw = emb[torch.randint(4, emb.shape[0], (B, K), generator=g)].clone()
    kw[1] = kw[0]; kw[1, 3] = emb[50]                           # two sequences that differ in ONE keyword
    sot, eot = ours._special_ids()
    assert (sot, eot) == (2, 3)
    with torch.no_grad():
        want = encode_keywords(ref, kw, K, sot, eot)
    ours = ours.cuda()
    with torch.no_grad():
        got = ours.encode_keywords(kw.cuda(), K).cpu()
    assert got.shape == want.shape == (B, rc.embed_dim)
    rel = (got - want).norm(dim=-1) / want.norm(dim=-1)
    print("encode_keywords relative error per row:", rel.tolist())
    assert rel.max().item() < 2.5e-2, rel
    assert_rows_match(got, want, 0.99, "text-tower feature")


def _cascaded_pair(seed, tmp_path, V=8112):
    """Base-dims C-base model (reduced vocabulary of V sub-words) + the oracle with the same weights."""
    from oracle.clip_ref import ClipRefConfig
    from oracle.hubert_ref import HubertRefConfig
    from oracle.speechclip_ref import SpeechClipRef
    from speechclip_amd.model import KWClip_GeneralTransformer
    from test_e2e_gpu import _share_weights
    g = _g(seed)
    vocab = torch.cat([torch.tensor([0, 320, 49406, 49407]), torch.randperm(49000, generator=g)[:V - 4] + 321]).numpy()
    vp = str(tmp_path / "vocab.npy")
    np.save(vp, np.stack([vocab, np.arange(len(vocab))[::-1] + 1], axis=1))
    torch.manual_seed(seed)
    model = KWClip_GeneralTransformer(make_config(parallel=False, cascaded=True, reduce_vocab=vp)).eval()
    with torch.no_grad():
        model.audio_encoder.weightedsum_layer.weights.copy_(0.5 * torch.randn(13, generator=g))
        bn = model.cascaded_branch.bn_layer.bn_layer
        bn.running_mean.copy_(0.05 * torch.randn(bn.running_mean.shape, generator=g))
        bn.running_var.copy_(1.0 + 0.2 * torch.rand(bn.running_var.shape, generator=g))
    ref = SpeechClipRef(HubertRefConfig.base(), ClipRefConfig.vit_b32(), parallel=False, cascaded=True,
                        reduced_vocab=torch.from_numpy(vocab)).eval()
    _share_weights(model, ref, parallel=False, cascaded=True)
    return g, model, ref


def test_cascaded_head_isolated_on_oracle_frames(tmp_path):
    """The whole HIP KW_CascadedBranch (algebraic K-query pooling, out-proj, LN, Linear, eval BatchNorm, cosine vs 8112 sub-words, VQ, row
    gather, text tower, L2 norm) fed the ORACLE's fp32 frame features rounded to bf16 -- the towers are out of the picture -- at the real
    C-base dimensions.  Decisive sub-words are planted per keyword row (the reference's own margin between random sub-words is ~1e-3, below
    what any bf16 head can resolve; near-tie behaviour is pinned separately by test_cosine_scores_vs_fp32_cosine_similarity): the 40 VQ
    targets must then equal the oracle's EXACTLY, and embedding / keywords / loss / VQ statistics are asserted unconditionally."""
    from oracle import speechclip_ref as R
    g, model, ref = _cascaded_pair(8, tmp_path)
    B, T, D, K = 5, 120, 768, 8
    lens = torch.tensor([120, 77, 100, 31, 120])
    frames = torch.randn(B, T, D, generator=g)
    frames = frames * (0.5 + torch.rand(B, 1, 1, generator=g)) + 0.3 * torch.randn(B, 1, D, generator=g)
    frames16 = frames.to(torch.bfloat16)
    fr = frames16.float()
    cb_ref = ref.cascaded_branch
    # 1. Kw_BatchNorm's running statistics := the statistics of this batch (a trained model's running statistics match its data; with the
    #    fresh 0 / 1 buffers every utterance of a random-init model yields the same 8 keyword vectors to within cos 0.986, and the embedding
    #    check could not tell utterances apart).  2. one planted sub-word per keyword row, at the table's typical norm.
    with torch.no_grad():
        src = torch.cat([cb_ref.cls.expand(B, -1, -1), fr], dim=1)
        x = cb_ref.linear_proj(cb_ref.self_att(src, R.keypadding_mask(T + K, lens + K))[:, :K])
        flat = x.permute(0, 2, 1).reshape(B, -1)
        for bn in (cb_ref.bn_layer.bn_layer, model.cascaded_branch.bn_layer.bn_layer):
            bn.running_mean.copy_(flat.mean(0))
            bn.running_var.copy_(flat.var(0, unbiased=False))
        kwv = cb_ref.bn_layer(x)
        emb = ref.clip.token_embedding.weight
        typical = emb.norm(dim=-1).mean()
        kn = F.normalize(kwv, dim=-1)                                                # [B, K, E]
        ubar = F.normalize(kn.mean(1, keepdim=True), dim=-1)                         # the K rows of an utterance share a direction (cos ~0.9):
        emb[10:10 + B * K] = typical * F.normalize(kn - 0.9 * ubar, dim=-1).reshape(B * K, -1)   # plant the part that tells them apart
        model.clip.model.token_embedding.weight.copy_(emb)
        model.clip.model.invalidate_packed()
        feat_ref, vq_ref, kw_ref = cb_ref(fr, lens)
        cos_ref = F.cosine_similarity(kwv[:, :, None, :], emb[None, None, :, :], dim=-1)
        cos_ref[..., [0, 2, 3]] = float("-inf")
        t2 = cos_ref.topk(2, dim=-1).values
        margin = (t2[..., 0] - t2[..., 1]).min().item()
    print(f"smallest oracle arg-max margin {margin:.3f}")
    assert margin > 0.08
    assert torch.equal(vq_ref["targets"].reshape(-1), torch.arange(B * K) + 10)
    model = model.cuda()
    with torch.no_grad():
        feat, vq, kw = model.cascaded_branch(audio_feat=frames16.cuda(), audio_len=lens.cuda())
        from speechclip_amd import ops
        ca = ops.l2norm(feat)
    assert torch.equal(vq["targets"].cpu(), vq_ref["targets"])
    assert torch.equal(kw.cpu(), kw_ref)                                                 # gathered rows of the shared table: bitwise
    ca_ref = R.l2_normalize(feat_ref)
    cc = assert_rows_match(ca, ca_ref, 0.99, "cascaded_audio_feat")
    print("centred cosine per row:", cc.tolist())
    img = F.normalize(torch.randn(B, ca_ref.shape[1], generator=g), dim=-1)
    ids = torch.tensor([1, 2, 2, 3, 4])
    loss_ref = R.masked_contrastive_loss(ca_ref, img, ids).item()
    loss = model.criterion(feat_A=ca, feat_B=img.cuda(), index=ids.cuda()).item()
    logit_err = ((ca.cpu() @ img.t() - ca_ref @ img.t()) / 0.07).abs().max().item()
    assert logit_err < 5e-2, logit_err
    assert abs(loss - loss_ref) < 2e-2, (loss, loss_ref)
    torch.testing.assert_close(vq["ent_per_t"].cpu(), vq_ref["ent_per_t"], rtol=2e-2, atol=2e-2)
    assert abs(float(vq["prob_perplexity"]) / float(vq_ref["prob_perplexity"]) - 1) < 2e-2
    assert abs(float(vq["code_perplexity"]) - float(vq_ref["code_perplexity"])) < 1e-3 * float(vq_ref["code_perplexity"])


def _load_c2(tmp_path):
    from test_e2e_gpu import _load_model
    vocab = np.array([0, 320, 510, 511] + list(range(5, 300, 3)))
    vp = str(tmp_path / "vocab.npy")
    np.save(vp, np.stack([vocab, np.arange(len(vocab))[::-1] + 1], axis=1))
    return _load_model("tiny_base_c2", cascaded=True, vocab_path=vp)


def test_decisive_cascaded_fixture_vs_reference_glue(tmp_path):
    """tests/golden/e2e_tiny_base_c2.npz: outputs of the reference's OWN KWClip_GeneralTransformer (cascaded) on a model whose sub-word
    arg-max margins are >= 0.5 (make_golden.plant_decisive_keywords).  Everything is asserted unconditionally: exact VQ targets, exact
    keywords, embedding in centred cosine (with the rotated-rows negative control), loss, VQ statistics."""
    g, model, batch = _load_c2(tmp_path)
    assert float(g["min_margin"]) > 0.5
    with torch.no_grad():
        loss_feats, log_metrics, others = model(batch)
        loss = model.compute_loss(loss_feats)["loss"].item()
    assert np.array_equal(others["vq_results"]["targets"].cpu().numpy(), g["vq_targets"])
    np.testing.assert_array_equal(others["keywords"].cpu().numpy(), g["keywords"])
    cc = assert_rows_match(loss_feats["cascaded_audio_feat"], torch.from_numpy(g["cascaded_audio_feat"]), 0.99, "cascaded_audio_feat")
    print("centred cosine per row:", cc.tolist())
    assert_rows_match(loss_feats["image_feat"], torch.from_numpy(g["image_feat"]), 0.99, "image_feat")
    assert abs(loss - float(g["loss"])) < 2e-2, (loss, float(g["loss"]))
    np.testing.assert_allclose(others["vq_results"]["ent_per_t"].cpu().numpy(), g["vq_ent_per_t"], rtol=3e-2, atol=3e-2)
    assert abs(log_metrics["softmax_temp"] - 0.1) < 1e-6

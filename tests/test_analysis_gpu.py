"""Analysis surface of the cascaded model: KW_CascadedBranch.getAttentionMap (kwClip.py:918-1001),
MultiheadAttentionAndNorm.extract_attention_map (TransformerModels.py:130-135) and the keyword de-tokenisation of validation_epoch_end
(kwClip.py:277-466), on the HIP kernels (sc_attention_probs_fwd, sc_topk_rows_f32, sc_cosine_scores) against fixtures produced by the
reference's own code (tests/golden/analysis_*.npz) and against torch."""
import json
import os

import numpy as np
import pytest
import torch

from test_e2e_gpu import _load_model

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
VOCAB = np.array([0, 320, 510, 511] + list(range(5, 300, 3)))


def _model(tmp_path, tag):
    vp = str(tmp_path / "vocab.npy")
    np.save(vp, np.stack([VOCAB, np.arange(len(VOCAB))[::-1] + 1], axis=1))
    g, model, batch = _load_model(tag, cascaded=True, vocab_path=vp)
    return g, np.load(os.path.join(GOLD, f"analysis_{tag}.npz")), model, batch


def test_topk_rows_matches_a_stable_sort():
    from speechclip_amd import ops
    g = torch.Generator().manual_seed(3)
    for R, V, K in ((37, 1000, 10), (5, 49408, 10), (64, 103, 5), (3, 7, 7)):
        x = torch.randn(R, V, generator=g)
        x[:, ::5] = torch.round(x[:, ::5] * 4) / 4            # plenty of exact ties
        if V > 20:
            x[0, 3] = x[0, 17] = 9.0                          # a tie at the very top
        vals, idx = ops.topk_rows(x.cuda(), K)
        order = np.lexsort((np.arange(V)[None, :].repeat(R, 0), -x.numpy()), axis=-1)[:, :K]      # value descending, index ascending
        assert np.array_equal(idx.cpu().numpy(), order), (R, V, K)
        assert torch.equal(vals.cpu(), torch.gather(x, 1, torch.from_numpy(order)))
    x = torch.full((2, 9), float("-inf"))
    x[0, 4] = 1.0
    vals, idx = ops.topk_rows(x.cuda(), 3)
    assert idx[0].tolist() == [4, 0, 1] and vals[0, 0].item() == 1.0 and torch.isinf(vals[0, 1])


@pytest.mark.parametrize("heads,hd,L", [(1, 128, 32), (8, 96, 57), (4, 64, 300)])
def test_attention_probs_vs_torch_multihead_attention(heads, hd, L):
    """need_weights=True, average_attn_weights=False of torch.nn.MultiheadAttention on the same (bf16-rounded) q|k|v."""
    from speechclip_amd import ops
    B, D = 3, heads * hd
    g = torch.Generator().manual_seed(11)
    qkv = (torch.randn(B * L, 3 * D, generator=g) * 0.7).to(torch.bfloat16)
    lens = torch.tensor([L, L // 2 + 1, 9])
    mask = torch.arange(L)[None, :] >= lens[:, None]
    q, k, _ = qkv.float().view(B, L, 3, heads, hd).permute(2, 0, 3, 1, 4)
    s = (q * hd ** -0.5) @ k.transpose(-1, -2)
    ref = torch.softmax(s.masked_fill(mask[:, None, None, :], float("-inf")), dim=-1)
    for n_rows in (None, 8):
        p = ops.attention_probs(qkv.cuda(), B, L, heads, hd, mask.cuda(), n_rows=n_rows).cpu()
        r = ref if n_rows is None else ref[:, :, :n_rows]
        assert p.shape == r.shape
        assert (p - r).abs().max().item() < 2e-5
        assert torch.all(p.masked_select(mask[:, None, None, :].expand_as(p)) == 0)         # exactly 0 at padded keys
        assert (p.sum(-1) - 1).abs().max().item() < 1e-5
    p = ops.attention_probs(qkv.cuda(), B, L, heads, hd, None).cpu()
    assert (p - torch.softmax(s, dim=-1)).abs().max().item() < 2e-5


def test_extract_attention_map_matches_torch_module():
    """MultiheadAttentionAndNorm.extract_attention_map == torch's own module called as the reference calls it (TransformerModels.py:131-134)."""
    from speechclip_amd.module.kw_modules.TransformerModels import MultiheadAttentionAndNorm
    torch.manual_seed(5)
    m = MultiheadAttentionAndNorm(d_model=128, nhead=4).eval()
    B, L = 3, 41
    src = torch.randn(B, L, 128)
    mask = torch.arange(L)[None, :] >= torch.tensor([41, 20, 8])[:, None]
    with torch.no_grad():
        o, w = m.multihead_attn_layer(src, src, src, key_padding_mask=mask, average_attn_weights=False)
        o = m.attentionBlock_Norm(o + src)
    mg = m.cuda()
    out, probs = mg.extract_attention_map(src.cuda(), mask.cuda())
    assert probs.shape == (B, 4, L, L) and out.shape == (B, L, 128)
    assert (probs.cpu() - w).abs().max().item() < 5e-3                 # bf16 q / k
    valid = ~mask
    assert (out.cpu() - o)[valid].abs().max().item() < 5e-2
    _, p8 = mg.extract_attention_map(src.cuda(), mask.cuda(), query_rows=8)
    assert torch.equal(p8, probs[:, :, :8])


@pytest.mark.parametrize("tag", ["tiny_base_c", "tiny_base_c2"])
def test_get_attention_map_vs_reference(tmp_path, tag):
    g, a, model, batch = _model(tmp_path, tag)
    feat, feat_len = torch.from_numpy(g["audio_feat"]).cuda(), torch.from_numpy(g["feat_len"]).cuda()
    cls_weights, topk_kw, none = model.cascaded_branch.getAttentionMap(feat, feat_len)
    assert none is None and len(cls_weights) == 4 and len(topk_kw) == 4
    ref_names = json.loads(str(a["topk_kw"]))
    for i, w in enumerate(cls_weights):
        L = int(g["feat_len"][i]) + 8
        assert w.shape == (1, 8, L) and w.dtype == torch.float32
        assert (w.cpu() - torch.from_numpy(a["attn_map"][i, :, :, :L])).abs().max().item() < 3e-3
        assert (w.sum(-1) - 1).abs().max().item() < 1e-5
    same_top1 = np.mean([topk_kw[b][k][0] == ref_names[b][k][0] for b in range(4) for k in range(8)])
    overlap = np.mean([len(set(topk_kw[b][k]) & set(ref_names[b][k])) / 10 for b in range(4) for k in range(8)])
    print(tag, "top-1 agreement", same_top1, "top-10 overlap", overlap)
    assert all(len(topk_kw[b][k]) == 10 and all(t.startswith("<") and not t.endswith("</w>") for t in topk_kw[b][k]) for b in range(4) for k in range(8))
    assert not any(t in ("<0>", "<510>", "<511>") for b in range(4) for k in range(8) for t in topk_kw[b][k][:5])    # special ids pushed down
    assert overlap > 0.9 and same_top1 >= (1.0 if tag == "tiny_base_c2" else 0.85)      # c2: decisive margins (planted sub-words)


@pytest.mark.parametrize("tag", ["tiny_base_c", "tiny_base_c2"])
def test_validation_epoch_end_detokenises_keywords_like_the_reference(tmp_path, tag):
    """The reference's validation_epoch_end wrote kw_hit_ep0.json / keywords_ep0.json for these keyword embeddings and captions
    (make_golden.gen_analysis); the same outputs go through this build's validation_epoch_end (similarity + top-K on the device)."""
    from speechclip_amd.base import OrderedNamespace
    g, a, model, batch = _model(tmp_path, tag)
    model.config.trainer.default_root_dir = str(tmp_path / "run")
    model.config.data = OrderedNamespace({"dev_batch_size": 3})
    outputs = [{"id": torch.from_numpy(g["id"]), "audio_feat": torch.from_numpy(g["cascaded_audio_feat"]), "image_feat": torch.from_numpy(g["image_feat"]),
                "keywords": torch.from_numpy(a["keywords"]), "gold_text": torch.from_numpy(a["text"])}]
    model.validation_epoch_end(outputs)
    hit_rate, kw_top_ret, retok = model.last_kw_hit_rate
    root = tmp_path / "run" / "detokenizeText"
    assert json.load(open(root / "kw_hit_ep0.json")) == kw_top_ret == json.loads(str(a["kw_hit"]))
    np.testing.assert_allclose(hit_rate.numpy(), a["hits_per_keyword"] / 4 * 100)
    ref = json.loads(str(a["retok"]))
    mine = json.load(open(root / "keywords_ep0.json"))
    assert len(mine) == len(ref) == 4
    for r, m in zip(ref, mine):
        assert r["gold"] == m["gold"]
        for k in range(8):
            rn, mn = r["neighbors"][f"keyword_{k}"], m["neighbors"][f"keyword_{k}"]
            assert [t[0] for t in rn] == [t[0] for t in mn], (k, rn, mn)
            np.testing.assert_allclose([t[1] for t in mn], [t[1] for t in rn], atol=2e-6)


def test_detokenise_pseudo_inverse_readout_vs_oracle(tmp_path):
    from oracle import speechclip_ref as R
    from speechclip_amd.base import OrderedNamespace
    g, a, model, batch = _model(tmp_path, "tiny_base_c")
    model.config.trainer.default_root_dir = str(tmp_path / "run")
    model.config.data = OrderedNamespace({"dev_batch_size": 2})
    model.config.model_settings.cascaded_branch.keyword.retrieve_method = "pseudo_inverse"
    kw = torch.from_numpy(a["keywords"])
    outputs = [{"keywords": kw[:3], "gold_text": torch.from_numpy(a["text"][:3])}, {"keywords": kw[3:], "gold_text": torch.from_numpy(a["text"][3:])}]
    hit_rate, kw_top_ret, retok = model.detokenize_keywords(outputs)
    r2o = {n: int(o) for n, o in enumerate(VOCAB)}
    gold = [set(int(t) for t in row[0]) for row in a["text"]]
    hr, v, ix, fh = R.detokenize_keywords(kw.view(4, 8, -1), gold, model.clip.model.token_embedding.weight.detach().cpu(), K=5, method="pseudo_inverse",
                                          reduced_to_original=r2o, chunk=2)
    assert fh == kw_top_ret and torch.allclose(hr, hit_rate)
    for x in range(4):
        for k in range(8):
            mine = retok[x]["neighbors"][f"keyword_{k}"]
            assert [t[0] for t in mine] == ["<{}></w>".format(r2o[int(i)]) for i in ix[x, k]]
            np.testing.assert_allclose([t[1] for t in mine], v[x, k].numpy(), atol=1e-4, rtol=1e-4)

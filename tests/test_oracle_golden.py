"""CPU tests: the oracle (oracle/*.py) against the golden vectors produced by the reference's own glue
(tests/golden/make_golden.py -> tests/golden/*.npz).  No GPU, no /root/reference at run time."""
import os

import numpy as np
import pytest
import torch

from oracle import speechclip_ref as R
from oracle.clip_ref import ClipRefConfig
from oracle.hubert_ref import HubertRefConfig, conv_out_length, feat_lengths

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    return np.load(os.path.join(GOLD, name))


def test_loss_known_answers():
    g = _load("loss.npz")
    a, b = torch.from_numpy(g["anchor_a"]), torch.from_numpy(g["anchor_b"])
    lu = R.masked_contrastive_loss(a, b, torch.from_numpy(g["anchor_ids_u"])).item()
    ld = R.masked_contrastive_loss(a, b, torch.from_numpy(g["anchor_ids_d"])).item()
    # values produced by the reference's MaskedContrastiveLoss
    assert abs(lu - 2.978872299194336) < 1e-5
    assert abs(ld - 2.9537737369537354) < 1e-5


def test_loss_option_sweep():
    g = _load("loss.npz")
    for B, E, margin, dcl, a2b, b2a, inv_t, use_ids, val in g["cases"]:
        B = int(B)
        fa, fb, ids = (torch.from_numpy(g[f"{k}_{B}"]) for k in ("fa", "fb", "ids"))
        mine = R.masked_contrastive_loss(fa, fb, ids if use_ids else None, inv_t, margin, bool(dcl), bool(a2b), bool(b2a)).item()
        assert abs(mine - val) < 2e-5 * max(1.0, abs(val)), (B, margin, dcl, a2b, b2a, val, mine)


def test_retrieval_golden():
    g = _load("retrieval.npz")
    aud, img = torch.from_numpy(g["aud"]), torch.from_numpy(g["img"])
    s = aud @ img.t()
    ab, ba, mean = R.mutual_retrieval(s, s.t(), torch.from_numpy(g["aud_ids"]), torch.from_numpy(g["img_ids"]), [1, 5, 10])
    for i, k in enumerate((1, 5, 10)):
        assert abs(ab[f"recall@{k}"] - g["recall_ab"][i]) < 1e-4
        assert abs(ba[f"recall@{k}"] - g["recall_ba"][i]) < 1e-4
        assert abs(mean[f"recall@{k}"] - g["recall_mean"][i]) < 1e-4


def test_small_ops_golden():
    g = _load("small_ops.npz")
    for n in (13, 25):
        for norm in (0, 1):
            h = [torch.from_numpy(x) for x in g[f"ws_{n}_{norm}_h"]]
            y = R.weighted_sum(h, torch.from_numpy(g[f"ws_{n}_{norm}_w"]), bool(norm))
            np.testing.assert_allclose(y.numpy(), g[f"ws_{n}_{norm}_y"], atol=1e-6)
    m = R.keypadding_mask(10, torch.from_numpy(g["kpm_lens"]))
    assert np.array_equal(m.numpy(), g["kpm_mask"])


def test_feat_len_table():
    tab = _load("feat_len.npz")["table"]
    for lmax, l, T, flen, nvalid in tab:
        assert conv_out_length(int(lmax)) == T
        assert int(feat_lengths([int(l)], 320, int(T))[0]) == flen


def _build(tag, hub_cfg, cascaded, parallel, norm_hidden, vocab=None):
    g = _load(f"e2e_{tag}.npz")
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")}
    m = R.SpeechClipRef(hub_cfg, ClipRefConfig.tiny(), parallel=parallel, cascaded=cascaded, branch_heads=4,
                        normalize_hiddenstates=norm_hidden, reduced_vocab=vocab).eval()
    m.encoder.load_state_dict({k[len("audio_encoder.encoder."):]: v for k, v in sd.items() if k.startswith("audio_encoder.encoder.")})
    m.clip.load_state_dict({k[len("clip.model."):]: v for k, v in sd.items() if k.startswith("clip.model.")})
    with torch.no_grad():
        m.ws_weights.copy_(sd["audio_encoder.weightedsum_layer.weights"])
    if parallel:
        m.parallel_branch.load_state_dict({k[len("parallel_branch."):]: v for k, v in sd.items() if k.startswith("parallel_branch.")})
    if cascaded:
        m.cascaded_branch.load_state_dict({k[len("cascaded_branch."):]: v for k, v in sd.items()
                                           if k.startswith("cascaded_branch.") and "vector_quantizer" not in k})
    batch = {k: torch.from_numpy(g[k]) for k in ("wav", "wav_len", "image", "id")}
    return g, m, batch


@pytest.mark.parametrize("tag,large", [("tiny_base_p", False), ("tiny_large_p", True)])
def test_e2e_parallel_golden(tag, large):
    cfg = HubertRefConfig.tiny(layer_norm_first=large, extractor_mode="layer_norm" if large else "default", conv_bias=large)
    g, m, batch = _build(tag, cfg, cascaded=False, parallel=True, norm_hidden=large)
    o = m(batch)
    assert np.array_equal(o["audio_len"].numpy(), g["feat_len"])
    np.testing.assert_allclose(o["audio_feat"].numpy(), g["audio_feat"], atol=3e-5)
    np.testing.assert_allclose(o["image_feat"].numpy(), g["image_feat"], atol=1e-5)
    np.testing.assert_allclose(o["parallel_audio_feat"].numpy(), g["parallel_audio_feat"], atol=1e-5)
    assert abs(m.compute_loss(o)["loss"].item() - float(g["loss"])) < 1e-5


@pytest.mark.parametrize("tag,large", [("tiny_base_p", False), ("tiny_large_p", True)])
@pytest.mark.parametrize("method", ["method1", "method2"])
def test_normalize_type_methods_golden(tag, large, method):
    """`normalize_hiddenstates` with `normalize_type` method1 / method2 (speech_encoder_plus.py:572-592): the oracle's restatement against what the
    reference's own FairseqSpeechEncoder_Hubert.forward returned on the e2e fixtures' weights and waves (tests/golden/norm_methods.npz)."""
    cfg = HubertRefConfig.tiny(layer_norm_first=large, extractor_mode="layer_norm" if large else "default", conv_bias=large)
    g, m, batch = _build(tag, cfg, cascaded=False, parallel=True, norm_hidden=False)
    n = _load("norm_methods.npz")
    with torch.no_grad():
        _, flen, hidden = m.forward_audio(batch["wav"], batch["wav_len"])
        hs = R.normalize_hidden_states(hidden, method)
        feat = R.weighted_sum(hs, m.ws_weights, False)
    assert np.array_equal(flen.numpy(), n[f"{tag}/{method}/feat_len"])
    np.testing.assert_allclose(hs[0].numpy(), n[f"{tag}/{method}/hidden_0"], atol=2e-6)
    np.testing.assert_allclose(hs[-1].numpy(), n[f"{tag}/{method}/hidden_last"], atol=2e-6)
    np.testing.assert_allclose(feat.numpy(), n[f"{tag}/{method}/feat"], atol=2e-6)


@pytest.mark.parametrize("tag", ["tiny_base_c", "tiny_base_c2"])
def test_e2e_cascaded_golden(tag):
    vocab = torch.tensor([0, 320, 510, 511] + list(range(5, 300, 3)))
    g, m, batch = _build(tag, HubertRefConfig.tiny(), cascaded=True, parallel=False, norm_hidden=False, vocab=vocab)
    o = m(batch)
    assert np.array_equal(o["vq_results"]["targets"].numpy(), g["vq_targets"])
    np.testing.assert_allclose(o["cascaded_audio_feat"].numpy(), g["cascaded_audio_feat"], atol=1e-5)
    np.testing.assert_allclose(o["keywords"].numpy(), g["keywords"], atol=1e-6)
    assert abs(m.compute_loss(o, w_par=0.0, w_casc=1.0)["loss"].item() - float(g["loss"])) < 1e-5


class _IdDecoder(dict):
    def __missing__(self, i):
        return "<{}></w>".format(int(i))


@pytest.mark.parametrize("tag", ["tiny_base_c", "tiny_base_c2"])
def test_cascaded_analysis_golden(tag):
    """tests/golden/analysis_<tag>.npz: KW_CascadedBranch.getAttentionMap and the keyword de-tokenisation of validation_epoch_end run by the
    reference's OWN code (make_golden.gen_analysis); the oracle's restatements reproduce both."""
    import json
    vocab = [0, 320, 510, 511] + list(range(5, 300, 3))
    g, m, batch = _build(tag, HubertRefConfig.tiny(), cascaded=True, parallel=False, norm_hidden=False, vocab=torch.tensor(vocab))
    a = _load(f"analysis_{tag}.npz")
    r2o = {n: o for n, o in enumerate(vocab)}
    feat, feat_len = torch.from_numpy(g["audio_feat"]), torch.from_numpy(g["feat_len"])
    with torch.no_grad():
        cw, names, ids, _ = R.get_attention_map(m.cascaded_branch, feat, feat_len, decoder=_IdDecoder(), reduced_to_original=r2o)
    for i, w in enumerate(cw):
        L = int(feat_len[i]) + 8
        assert w.shape == (1, 8, L)
        np.testing.assert_allclose(w.numpy(), a["attn_map"][i, :, :, :L], atol=1e-6)
        assert np.all(a["attn_map"][i, :, :, L:] == 0)
        np.testing.assert_allclose(w.sum(-1).numpy(), 1.0, atol=1e-5)
    assert names == json.loads(str(a["topk_kw"]))
    gold = [set(int(t) for t in row[0]) for row in a["text"]]
    K = a["neighbor_ids"].shape[-1]
    hr, v, ix, first_hits = R.detokenize_keywords(torch.from_numpy(a["keywords"]).view(4, 8, -1), gold, m.clip.token_embedding.weight, K=K,
                                                  reduced_to_original=r2o, chunk=3)
    assert np.array_equal(np.vectorize(r2o.get)(ix.numpy()), a["neighbor_ids"])
    np.testing.assert_allclose(v.numpy(), a["neighbor_vals"], atol=1e-6)
    assert first_hits == json.loads(str(a["kw_hit"]))
    np.testing.assert_allclose(hr.numpy(), a["hits_per_keyword"] / 4 * 100)

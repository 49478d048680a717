#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's own glue code.

Runs ONLY in the build container (needs /root/reference, which is absent on the GPU box).
Recipe = SURVEY.md Appendix A: pre-seed sys.modules with stubs for the third-party packages the
reference imports but that are not installed (pytorch_lightning, fairseq, clip, s3prl, librosa,
torchvision), plug the oracle's restatements of the fairseq / openai backbones in *through those
stubs*, then import `avssl.*` from /root/reference and execute it.  Outputs are small .npz files:
inputs, (tiny) weights and the reference's outputs.  Every fixture is also recomputed with the
standalone oracle (oracle/speechclip_ref.py) and must agree, which pins the oracle.

Usage:  python tests/golden/make_golden.py
"""
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

# transformers first (Appendix A step 1), before the librosa stub exists
import transformers  # noqa: F401,E402

from oracle import clip_ref, hubert_ref, speechclip_ref  # noqa: E402

STATE = {"hubert_cfg": None, "clip_cfg": None}


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install_stubs():
    class LightningModule(nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, *a, **k):
            pass

        def log_dict(self, *a, **k):
            pass

        @property
        def device(self):
            return torch.device("cpu")

        logger = None
        global_step = 0

    class _Dummy:
        def __init__(self, *a, **k):
            pass

    pl = _mod("pytorch_lightning", LightningModule=LightningModule, Trainer=_Dummy, seed_everything=lambda s: torch.manual_seed(s))
    _mod("pytorch_lightning.callbacks", ModelCheckpoint=_Dummy, TQDMProgressBar=_Dummy, Callback=_Dummy)
    _mod("pytorch_lightning.loggers", TensorBoardLogger=_Dummy, WandbLogger=_Dummy, LightningLoggerBase=_Dummy)
    _mod("pytorch_lightning.loggers.wandb", WandbLogger=_Dummy)
    pl.loggers = sys.modules["pytorch_lightning.loggers"]

    class _IdDecoder(dict):
        def __missing__(self, i):
            return "<{}></w>".format(int(i))

    class SimpleTokenizer:
        """id-literal stand-in for openai's tokenizer (its BPE vocabulary is not available offline): sub-word i prints as "<i>"; decode /
        encode have the real tokenizer's call signatures and are inverses of each other on id lists."""

        def __init__(self):
            self.encoder = {"<|startoftext|>": 49406, "<|endoftext|>": 49407}
            self.decoder = _IdDecoder()

        def decode(self, tokens):
            return "".join(self.decoder[t] for t in tokens).replace("</w>", " ")

        def encode(self, text):
            return [int(w[1:-1]) for w in text.split()]

    def clip_load(name, device="cpu"):
        torch.manual_seed(1234)
        return clip_ref.ClipRef(STATE["clip_cfg"]).eval().float(), (lambda img: img)

    _mod("clip", load=clip_load, tokenize=lambda *a, **k: None)
    _mod("clip.simple_tokenizer", SimpleTokenizer=SimpleTokenizer)

    def load_model_ensemble_and_task(paths):
        torch.manual_seed(4321)
        cfg = STATE["hubert_cfg"]
        model = hubert_ref.HubertModelRef(cfg)
        hubert_ref.randomize_norm_affine(model, torch.Generator().manual_seed(99))
        task = types.SimpleNamespace(cfg=types.SimpleNamespace(normalize=cfg.normalize))
        return [model], cfg, task

    fs = _mod("fairseq")
    fs.checkpoint_utils = _mod("fairseq.checkpoint_utils", load_model_ensemble_and_task=load_model_ensemble_and_task)
    _mod("fairseq.models")
    _mod("fairseq.models.hubert")
    _mod("fairseq.models.hubert.hubert", HubertConfig=hubert_ref.HubertRefConfig, HubertModel=hubert_ref.HubertModelRef)
    _mod("fairseq.models.wav2vec")
    _mod("fairseq.models.wav2vec.wav2vec2", TransformerEncoder=hubert_ref.TransformerEncoderRef)

    def index_put(t, idx, v):
        t[idx] = v
        return t

    _mod("fairseq.utils", index_put=index_put)
    s3 = _mod("s3prl")
    s3.hub = _mod("s3prl.hub")
    _mod("s3prl.utility")
    _mod("s3prl.utility.download", _urls_to_filepaths=lambda url, refresh=False: "/nonexistent.pt")
    _mod("librosa")
    tv = _mod("torchvision")
    tv.transforms = _mod("torchvision.transforms", Compose=_Dummy, Resize=_Dummy, ToTensor=_Dummy)
    _mod("tqdm", tqdm=lambda x, *a, **k: x)
    import transformers.file_utils as fu
    if not hasattr(fu, "copy_func"):
        fu.copy_func = lambda f: f
    _mod("PIL", Image=types.SimpleNamespace())
    sys.modules["PIL.Image"] = sys.modules["PIL"].Image
    sys.path.insert(0, REF)
    # /root/reference/avssl has no __init__.py (namespace package), so this repo's regular `avssl/` alias package would win the import
    # regardless of path order: pin `avssl` to the reference tree explicitly.
    ref_pkg = types.ModuleType("avssl")
    ref_pkg.__path__ = [os.path.join(REF, "avssl")]
    sys.modules["avssl"] = ref_pkg


def np_state(sd):
    return {k: v.detach().cpu().numpy() for k, v in sd.items() if v.dtype != torch.bool}


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


# ----------------------------------------------------------------------------------------------
def gen_loss(losses_mod):
    out = {}
    # known-answer anchors recorded in BASELINE.md section 3
    torch.manual_seed(0)
    a = torch.nn.functional.normalize(torch.randn(16, 512), dim=-1)
    b = torch.nn.functional.normalize(torch.randn(16, 512), dim=-1)
    crit = losses_mod.MaskedContrastiveLoss()
    ids_u = torch.arange(16)
    ids_d = torch.tensor([0, 0, 0, 1, 1] + list(range(2, 13)))
    l_u, l_d = crit(a, b, ids_u).item(), crit(a, b, ids_d).item()
    print("anchors:", l_u, l_d)
    assert abs(l_u - 2.978872299194336) < 1e-6 and abs(l_d - 2.9537737369537354) < 1e-6
    out.update(anchor_a=a.numpy(), anchor_b=b.numpy(), anchor_ids_u=ids_u.numpy(), anchor_ids_d=ids_d.numpy(),
               anchor_loss_u=np.float64(l_u), anchor_loss_d=np.float64(l_d))
    # option sweep at B in {2, 16, 256}; B = 2048 with MAX_EYE patched before construction (Appendix A step 5)
    cases = []
    g = torch.Generator().manual_seed(5)
    for B, E in ((2, 8), (16, 64), (256, 512), (2048, 64)):
        losses_mod.MAX_EYE = max(256, B)
        fa = torch.nn.functional.normalize(torch.randn(B, E, generator=g), dim=-1)
        fb = torch.nn.functional.normalize(fa + 0.7 * torch.randn(B, E, generator=g), dim=-1)
        ids = torch.randint(0, max(2, B // 3), (B,), generator=g)
        for kw in (dict(), dict(margin=0.2), dict(dcl=True), dict(a2b=False), dict(b2a=False),
                   dict(temperature=0.05, temperature_trainable=True), dict(use_ids=False)):
            kw = dict(kw)
            use_ids = kw.pop("use_ids", True)
            c = losses_mod.MaskedContrastiveLoss(**kw)
            with torch.no_grad():
                val = c(fa.clone(), fb.clone(), ids if use_ids else None).item()
            inv_t = c.temperature.exp().item() if kw.get("temperature_trainable") else c.temperature
            mine = speechclip_ref.masked_contrastive_loss(fa, fb, ids if use_ids else None, inv_t, kw.get("margin", 0.0),
                                                          kw.get("dcl", False), kw.get("a2b", True), kw.get("b2a", True)).item()
            assert abs(val - mine) < 2e-5 * max(1, abs(val)), (B, kw, val, mine)
            cases.append((B, E, kw.get("margin", 0.0), int(kw.get("dcl", False)), int(kw.get("a2b", True)),
                          int(kw.get("b2a", True)), inv_t, int(use_ids), val))
        out[f"fa_{B}"], out[f"fb_{B}"], out[f"ids_{B}"] = fa.numpy(), fb.numpy(), ids.numpy()
    losses_mod.MAX_EYE = 256
    out["cases"] = np.array(cases, dtype=np.float64)
    # the documented limitation: B > 256 raises in the reference
    try:
        losses_mod.MaskedContrastiveLoss()(torch.randn(300, 8), torch.randn(300, 8), torch.arange(300))
        raised = False
    except IndexError:
        raised = True
    assert raised
    save("loss.npz", **out)


def gen_retrieval(retrieval_mod):
    g = torch.Generator().manual_seed(11)
    n_img, cap = 40, 5
    img = torch.nn.functional.normalize(torch.randn(n_img, 32, generator=g), dim=-1)
    img_ids = torch.randperm(1000, generator=g)[:n_img]
    aud = torch.nn.functional.normalize(img.repeat_interleave(cap, 0) + 0.9 * torch.randn(n_img * cap, 32, generator=g), dim=-1)
    aud_ids = img_ids.repeat_interleave(cap, 0)
    perm = torch.randperm(n_img * cap, generator=g)
    aud, aud_ids = aud[perm], aud_ids[perm]
    score = aud @ img.t()
    ab, ba, mean = retrieval_mod.mutualRetrieval(score, score.t().contiguous(), aud_ids, img_ids, [1, 5, 10])
    o_ab, o_ba, o_mean = speechclip_ref.mutual_retrieval(score, score.t(), aud_ids, img_ids, [1, 5, 10])
    for k in ab:
        assert abs(ab[k] - o_ab[k]) < 1e-4 and abs(ba[k] - o_ba[k]) < 1e-4 and abs(mean[k] - o_mean[k]) < 1e-4
    save("retrieval.npz", aud=aud.numpy(), img=img.numpy(), aud_ids=aud_ids.numpy(), img_ids=img_ids.numpy(),
         recall_ab=np.array([ab[f"recall@{k}"] for k in (1, 5, 10)]),
         recall_ba=np.array([ba[f"recall@{k}"] for k in (1, 5, 10)]),
         recall_mean=np.array([mean[f"recall@{k}"] for k in (1, 5, 10)]))


def gen_small_ops(ws_mod, du_mod):
    g = torch.Generator().manual_seed(3)
    out = {}
    for n in (13, 25):
        for norm in (False, True):
            layer = ws_mod.WeightedSumLayer(n, normalize_features=norm)
            with torch.no_grad():
                layer.weights.copy_(torch.randn(n, generator=g))
            hs = [torch.randn(2, 7, 16, generator=g) for _ in range(n)]
            with torch.no_grad():
                y = layer(hs)
            mine = speechclip_ref.weighted_sum(hs, layer.weights.detach(), norm)
            assert torch.allclose(y, mine, atol=1e-6)
            out[f"ws_{n}_{int(norm)}_w"] = layer.weights.detach().numpy()
            out[f"ws_{n}_{int(norm)}_h"] = torch.stack(hs).numpy()
            out[f"ws_{n}_{int(norm)}_y"] = y.numpy()
    lens = torch.tensor([1, 5, 9, 10])
    m = du_mod.get_keypadding_mask(10, lens)
    assert torch.equal(m, speechclip_ref.keypadding_mask(10, lens))
    out["kpm_lens"], out["kpm_mask"] = lens.numpy(), m.numpy()
    # LR schedules of the reference (avssl/optim/scheduler.py:10-47): learning rate after k scheduler steps
    import avssl.optim.scheduler as sched_mod
    steps = [0, 1, 2, 10, 11, 12, 40, 99, 100]
    for name, kw in (("linear_warmup_decay", dict(warmup=10, max_step=100, final_lr=1e-8)), ("noam", dict(warmup=10))):
        prm = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.Adam([prm], lr=1e-4)
        sch = sched_mod.get_scheduler(name, opt, **kw)
        lrs = []
        for k in range(max(steps) + 1):
            if k in steps:
                lrs.append(opt.param_groups[0]["lr"])
            opt.step()
            sch.step()
        out[f"sched_{name}_lr"] = np.array(lrs, dtype=np.float64)
    out["sched_steps"] = np.array(steps)
    # collate_general (avssl/data/collate_function.py:7-36) on ragged rows -- loaded from its file: the package __init__ pulls in the datasets
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_collate_function", f"{REF}/avssl/data/collate_function.py")
    cf = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cf)
    gc = torch.Generator().manual_seed(9)
    lens_c = [5, 17, 9, 17, 1]
    rows = [{"wav": torch.randn(n, generator=gc), "image": torch.randn(3, 4, 4, generator=gc), "id": 7 * i + 1} for i, n in enumerate(lens_c)]
    col = cf.collate_general(rows)
    assert list(col.keys()) == ["wav", "image", "id", "wav_len"]
    out["collate_lens"] = np.array(lens_c)
    out["collate_wav_flat"] = torch.cat([r["wav"] for r in rows]).numpy()
    out["collate_image_rows"] = torch.stack([r["image"] for r in rows]).numpy()
    for k, v in col.items():
        out["collate_out_" + k] = v.numpy()
    save("small_ops.npz", **out)


def tiny_config(OrderedNamespace, yaml_path, d_model, n_kw=8, cascaded=False, parallel=True, branch_heads=4,
                normalize_hiddenstates=False, temperature_trainable=False, reduce_vocab=None):
    import yaml
    cfg = yaml.load(open(yaml_path), Loader=yaml.FullLoader)
    ms = cfg["model_settings"]
    ms["cascaded_objective_weight"] = 1.0 if cascaded else 0.0
    ms["parallel_objective_weight"] = 1.0 if parallel else 0.0
    ms["parallel_branch"]["transformer_args"].update(d_model=d_model, nhead=branch_heads, dim_feedforward=4 * d_model)
    ms["cascaded_branch"]["transformer_args"].update(d_model=d_model, nhead=1, dim_feedforward=4 * d_model)
    ms["cascaded_branch"]["keyword"]["number"] = n_kw
    cfg["audio_encoder"]["normalize_hiddenstates"] = normalize_hiddenstates
    cfg["cl_loss"]["args"]["temperature_trainable"] = temperature_trainable
    cfg["clip"]["reduce_subword_embbedding"] = reduce_vocab
    return OrderedNamespace(cfg)


def plant_decisive_keywords(model, batch, gen):
    """Second cascaded fixture: make every arg-max of the sub-word retrieval decisive, so that a bf16 implementation must
    reproduce the reference's VQ targets EXACTLY and the embedding / loss / gradient checks need no `if targets agree` escape.
      1. Kw_BatchNorm's running statistics are set to the statistics of this batch (what a trained model's running statistics look like for
         in-distribution data; also makes the train-mode and eval-mode keyword vectors coincide);
      2. for every (utterance, keyword) row, one (non-special) row of the reduced sub-word table is replaced by a noisy copy of the
         reference's own keyword vector for that row, scaled to the table's typical norm.
    Everything is computed by the reference's own modules (hooks on model.cascaded_branch.bn_layer).  Returns the smallest top1 - top2 margin."""
    import torch.nn.functional as F
    cb = model.cascaded_branch
    cap = {}
    h = cb.bn_layer.register_forward_pre_hook(lambda m, inp: cap.__setitem__("x", inp[0].detach().clone()))
    with torch.no_grad():
        model.forward(batch)
    h.remove()
    x = cap["x"]                                                      # [B, K, D] = linear_proj output
    B, K, D = x.shape
    flat = x.permute(0, 2, 1).reshape(B, -1)                          # kw_bn.py:122-126 feature order
    bn = cb.bn_layer.bn_layer
    with torch.no_grad():
        bn.running_mean.copy_(flat.mean(0))
        bn.running_var.copy_(flat.var(0, unbiased=False))
    h = cb.bn_layer.register_forward_hook(lambda m, inp, out: cap.__setitem__("kw", out.detach().clone()))
    with torch.no_grad():
        model.forward(batch)
    h.remove()
    kw = cap["kw"].reshape(B * K, D)
    emb = model.clip.model.token_embedding.weight
    typical = emb.norm(dim=-1).mean().item()
    # In a random-init model the K keyword rows of one utterance nearly coincide after the BatchNorm (1-head attention over random frames is
    # close to uniform pooling: the utterance-dependent part is shared by all K queries, the query-dependent part is utterance-independent
    # and removed by the BatchNorm), while different utterances are decorrelated: one planted sub-word per utterance.
    kn = F.normalize(kw, dim=-1).view(B, K, D)
    assign = [b for b in range(B) for _ in range(K)]
    with torch.no_grad():
        for b in range(B):
            noise = F.normalize(torch.randn(D, generator=gen), dim=0)
            emb[4 + b] = typical * F.normalize(F.normalize(kn[b].mean(0), dim=0) + 0.1 * noise, dim=0)
    cos = F.cosine_similarity(kw[:, None, :], emb[None, :, :], dim=-1)
    cos[:, [0, 2, 3]] = float("-inf")
    top2 = cos.topk(2, dim=-1).values
    print(f"planted {B} sub-words for {B * K} keyword rows; own cos min {top2[:, 0].min():.3f}, runner-up max {top2[:, 1].max():.3f}")
    assert torch.equal(cos.argmax(-1), torch.tensor(assign) + 4)
    return (top2[:, 0] - top2[:, 1]).min().item()


def gen_analysis(model, batch, feat, feat_len, others, tag):
    """Analysis surface of the cascaded model, from the reference's own code: KW_CascadedBranch.getAttentionMap (kwClip.py:918-1001) and the
    keyword de-tokenisation of KWClipBase.validation_epoch_end (:277-466) -> tests/golden/analysis_<tag>.npz."""
    import json
    import tempfile
    cb, clip = model.cascaded_branch, model.clip
    with torch.no_grad():
        cls_weights, topk_kw, _ = cb.getAttentionMap(feat, feat_len)
    B, K = feat.shape[0], cb.keyword_num
    H = cls_weights[0].shape[0]
    amap = np.zeros((B, H, K, feat.shape[1] + K), dtype=np.float32)
    for i, w in enumerate(cls_weights):
        assert w.shape == (H, K, int(feat_len[i]) + K)
        amap[i, :, :, : w.shape[-1]] = w.numpy()
    # gold captions: SOT, six sub-words, EOT, padding; every second utterance holds the 3rd-nearest neighbour of keyword (x mod K)
    emb = clip.model.token_embedding.weight.detach()
    kw = others["keywords"].view(B, K, -1)
    nn_ids = torch.topk(torch.nn.functional.cosine_similarity(kw.reshape(-1, kw.shape[-1], 1), emb.T.unsqueeze(0), dim=1), 10)[1].view(B, K, 10)
    g = torch.Generator().manual_seed(4242)
    ids = np.array(clip.selected_text_emb_ids)
    text = torch.zeros(B, 1, 77, dtype=torch.long)
    for x in range(B):
        words = ids[4:][torch.randperm(len(ids) - 4, generator=g)[:6].numpy()].tolist()
        if x % 2 == 0:
            words[2] = clip.reducedl2Original[int(nn_ids[x, x % K, 2])]
        row = [clip.tokenizer.encoder["<|startoftext|>"]] + words + [clip.tokenizer.encoder["<|endoftext|>"]]
        text[x, 0, : len(row)] = torch.tensor(row)
    root = tempfile.mkdtemp()
    model.config.trainer.default_root_dir = root
    model.config.data.dev_batch_size = 3            # 4 utterances -> chunks of 3 + 1
    if not hasattr(type(model), "current_epoch"):
        type(model).current_epoch = 0
    with torch.no_grad():
        out = model.validation_step_end(model.validation_step(dict(batch, text=text), 0))
        try:
            model.validation_epoch_end([out])
        except Exception as e:                      # the retrieval half after the keyword logging is covered by its own fixtures
            print("validation_epoch_end after the keyword logs:", type(e).__name__, e)
    kw_hit = json.load(open(os.path.join(root, "detokenizeText", "kw_hit_ep0.json")))
    retok = json.load(open(os.path.join(root, "detokenizeText", "keywords_ep0.json")))
    assert len(retok) == B and len(kw_hit) == K
    nb_ids = np.array([[[int(t[0][1:-5]) for t in r["neighbors"]["keyword_%d" % k]] for k in range(K)] for r in retok])     # "<id></w>"
    nb_val = np.array([[[t[1] for t in r["neighbors"]["keyword_%d" % k]] for k in range(K)] for r in retok], dtype=np.float64)
    hits = np.array([len(h) for h in kw_hit])
    print(f"analysis_{tag}: hits per keyword {hits.tolist()}, attention map {amap.shape}")
    assert hits.sum() >= B // 2
    save(f"analysis_{tag}.npz", text=text.numpy(), attn_map=amap, topk_kw=np.array(json.dumps(topk_kw)), kw_hit=np.array(json.dumps(kw_hit)),
         retok=np.array(json.dumps(retok)), neighbor_ids=nb_ids, neighbor_vals=nb_val, hits_per_keyword=hits,
         keywords=others["keywords"].numpy())


def gen_end_to_end(kwclip_mod, OrderedNamespace, tag, hubert_cfg, clip_cfg, cascaded, parallel, lens, normalize_hiddenstates,
                   reduce_vocab_ids=None, plant_margins=False, analysis_only=False):
    STATE["hubert_cfg"], STATE["clip_cfg"] = hubert_cfg, clip_cfg
    vocab_path = None
    if reduce_vocab_ids is not None:
        vocab_path = os.path.join("/tmp", f"vocab_{tag}.npy")
        np.save(vocab_path, np.stack([reduce_vocab_ids, np.arange(len(reduce_vocab_ids))[::-1] + 1], axis=1))
    # the reference tokenizer stub reports 49406/49407; map them to the tiny vocab's last two ids
    import clip.simple_tokenizer as st
    st_enc = {"<|startoftext|>": clip_cfg.vocab_size - 2, "<|endoftext|>": clip_cfg.vocab_size - 1}
    _plain_init = getattr(st.SimpleTokenizer, "_plain_init", None) or st.SimpleTokenizer.__init__
    st.SimpleTokenizer._plain_init = _plain_init

    def _tok_init(self):
        _plain_init(self)
        self.encoder = st_enc
    st.SimpleTokenizer.__init__ = _tok_init
    d = hubert_cfg.encoder_embed_dim
    cfg = tiny_config(OrderedNamespace, f"{REF}/config/speechCLIP/model_base/spchclp_{'c' if cascaded else 'p'}.yaml",
                      d_model=d, cascaded=cascaded, parallel=parallel, branch_heads=4,
                      normalize_hiddenstates=normalize_hiddenstates, reduce_vocab=vocab_path)
    torch.manual_seed(2024)
    model = kwclip_mod.KWClip_GeneralTransformer(cfg).eval()
    g = torch.Generator().manual_seed(77)
    with torch.no_grad():
        model.audio_encoder.weightedsum_layer.weights.copy_(0.5 * torch.randn(hubert_cfg.encoder_layers + 1, generator=g))
    B, lmax = len(lens), max(lens)
    wav = torch.zeros(B, lmax)
    for i, l in enumerate(lens):
        wav[i, :l] = 0.3 * torch.randn(l, generator=g)
    res = clip_cfg.image_resolution
    batch = {"wav": wav, "wav_len": torch.tensor(lens), "image": torch.randn(B, 3, res, res, generator=g),
             "id": torch.tensor([7, 7, 3, 9, 11, 3][:B])}
    min_margin = None
    if plant_margins:
        min_margin = plant_decisive_keywords(model, batch, g)
        print(f"{tag}: smallest arg-max margin after planting = {min_margin:.3f}")
        assert min_margin > 0.2
    with torch.no_grad():
        losses, log_metrics, others = model.forward(batch)
        loss = model.compute_loss(losses)
        feat, feat_len, hidden = model.forward_audio(batch["wav"], batch["wav_len"], return_hidden_states=True)
    # standalone oracle on the same weights must agree
    sc = speechclip_ref.SpeechClipRef(hubert_cfg, clip_cfg, parallel=parallel, cascaded=cascaded, branch_heads=4,
                                      normalize_hiddenstates=normalize_hiddenstates,
                                      reduced_vocab=torch.tensor(reduce_vocab_ids) if reduce_vocab_ids is not None else None).eval()
    sd = model.state_dict()
    sc.encoder.load_state_dict({k[len("audio_encoder.encoder."):]: v for k, v in sd.items() if k.startswith("audio_encoder.encoder.")})
    sc.clip.load_state_dict({k[len("clip.model."):]: v for k, v in sd.items() if k.startswith("clip.model.")})
    with torch.no_grad():
        sc.ws_weights.copy_(sd["audio_encoder.weightedsum_layer.weights"])
    if parallel:
        sc.parallel_branch.load_state_dict({k[len("parallel_branch."):]: v for k, v in sd.items() if k.startswith("parallel_branch.")})
    if cascaded:
        sc.cascaded_branch.load_state_dict({k[len("cascaded_branch."):]: v for k, v in sd.items()
                                            if k.startswith("cascaded_branch.") and not k.startswith("cascaded_branch.clip.")
                                            and "vector_quantizer" not in k})
    o = sc(batch)
    assert torch.equal(o["audio_len"], feat_len), (o["audio_len"], feat_len)
    assert torch.allclose(o["audio_feat"], feat, atol=2e-5), (o["audio_feat"] - feat).abs().max()
    assert torch.allclose(o["image_feat"], losses["image_feat"], atol=1e-5)
    arrays = {"wav": wav.numpy(), "wav_len": np.array(lens), "image": batch["image"].numpy(), "id": batch["id"].numpy(),
              "audio_feat": feat.numpy(), "feat_len": feat_len.numpy(), "image_feat": losses["image_feat"].numpy(),
              "hidden_last": hidden[-1].numpy(), "hidden_0": hidden[0].numpy(), "loss": np.float64(loss["loss"].item())}
    if parallel:
        assert torch.allclose(o["parallel_audio_feat"], losses["parallel_audio_feat"], atol=1e-5)
        arrays["parallel_audio_feat"] = losses["parallel_audio_feat"].numpy()
    if cascaded:
        assert torch.allclose(o["cascaded_audio_feat"], losses["cascaded_audio_feat"], atol=1e-5), \
            (o["cascaded_audio_feat"] - losses["cascaded_audio_feat"]).abs().max()
        assert torch.equal(o["vq_results"]["targets"], others["vq_results"]["targets"])
        arrays["cascaded_audio_feat"] = losses["cascaded_audio_feat"].numpy()
        arrays["vq_targets"] = others["vq_results"]["targets"].numpy()
        arrays["vq_ent_per_t"] = others["vq_results"]["ent_per_t"].numpy()
        arrays["keywords"] = others["keywords"].numpy()
        if plant_margins:
            arrays["min_margin"] = np.float64(min_margin)
            assert int(others["vq_results"]["targets"].min()) >= 4 and int(others["vq_results"]["targets"].max()) < 4 + len(lens) * 8
    my_loss = sc.compute_loss(o, w_par=1.0 if parallel else 0.0, w_casc=1.0 if cascaded else 0.0)["loss"].item()
    assert abs(my_loss - loss["loss"].item()) < 1e-5
    if cascaded:
        gen_analysis(model, batch, feat, feat_len, others, tag)
        # the oracle's restatement of the two analysis functions agrees with the reference on the same weights
        ana = np.load(os.path.join(HERE, f"analysis_{tag}.npz"))
        r2o = model.clip.reducedl2Original
        cw, names, _, _ = speechclip_ref.get_attention_map(sc.cascaded_branch, feat, feat_len, decoder=model.clip.tokenizer.decoder,
                                                          reduced_to_original=r2o)
        for i, w in enumerate(cw):
            assert torch.allclose(w, torch.from_numpy(ana["attn_map"][i, :, :, : w.shape[-1]]), atol=1e-6)
        import json as _json
        assert names == _json.loads(str(ana["topk_kw"]))
        gold = [set(int(t) for t in row[0]) for row in ana["text"]]
        hr, v, ix, fh = speechclip_ref.detokenize_keywords(torch.from_numpy(ana["keywords"]).view(len(lens), -1, ana["keywords"].shape[-1]),
                                                          gold, model.clip.model.token_embedding.weight, K=ana["neighbor_ids"].shape[-1], reduced_to_original=r2o, chunk=3)
        assert np.array_equal(np.array([r2o[int(t)] for t in ix.reshape(-1)]).reshape(ix.shape), ana["neighbor_ids"])
        assert np.allclose(v.numpy(), ana["neighbor_vals"], atol=1e-6)
        assert fh == _json.loads(str(ana["kw_hit"])), (fh, _json.loads(str(ana["kw_hit"])))
    if analysis_only:
        saved = np.load(os.path.join(HERE, f"e2e_{tag}.npz"))
        for k, v in np_state(sd).items():           # same seeds -> the same weights the committed e2e fixture carries
            if "sd/" + k in saved.files:
                assert np.array_equal(saved["sd/" + k], v), k
        return
    if parallel and not cascaded:
        # gradients of the trainable tail from the reference's own modules (eval mode: dropout off; HuBERT / CLIP frozen):
        # loss.backward() as Lightning would call it after training_step_end (kwClip.py:147-191)
        model.zero_grad()
        losses_g, _, _ = model.forward(batch)
        model.compute_loss(losses_g)["loss"].backward()
        n_grad = 0
        for k, prm in model.named_parameters():
            if prm.grad is not None and (k.startswith("parallel_branch.") or k == "audio_encoder.weightedsum_layer.weights"):
                arrays["grad/" + k] = prm.grad.detach().numpy().copy()
                n_grad += 1
        assert n_grad == 18, n_grad   # 17 branch tensors + the layer-mix weights
        # and the oracle's autograd on the same weights must agree with them
        sc.zero_grad()
        feat_o = speechclip_ref.weighted_sum([h.detach() for h in sc.forward_audio(batch["wav"], batch["wav_len"])[2]], sc.ws_weights, normalize_hiddenstates)
        pa = speechclip_ref.l2_normalize(sc.parallel_branch(feat_o, o["audio_len"]))
        speechclip_ref.masked_contrastive_loss(pa, o["image_feat"], batch["id"], sc.inv_temperature).backward()
        for k, prm in sc.parallel_branch.named_parameters():
            assert torch.allclose(prm.grad, torch.from_numpy(arrays["grad/parallel_branch." + k]), atol=2e-5, rtol=1e-3), k
        assert torch.allclose(sc.ws_weights.grad, torch.from_numpy(arrays["grad/audio_encoder.weightedsum_layer.weights"]), atol=2e-5, rtol=1e-3)
    if parallel and not cascaded:
        # a short optimisation trajectory with the reference's own modules and optimizer recipe (Adam, weight decay 1e-6, clip_grad_norm_ 4:
        # spchclp_p.yaml:96-118,:125), dropout off (eval mode), on a deep copy so the saved weights stay the initial ones
        import copy
        m2 = copy.deepcopy(model)
        prm2 = [p for k, p in m2.named_parameters() if k.startswith("parallel_branch.") or k == "audio_encoder.weightedsum_layer.weights"]
        opt2 = torch.optim.Adam(prm2, lr=1e-3, weight_decay=1e-6)
        seq = []
        for _ in range(4):
            opt2.zero_grad()
            lg, _, _ = m2.forward(batch)
            l2 = m2.compute_loss(lg)["loss"]
            seq.append(l2.item())
            l2.backward()
            torch.nn.utils.clip_grad_norm_(prm2, 4.0)
            opt2.step()
        arrays["train/loss_seq"] = np.array(seq, dtype=np.float64)
        del m2
    if cascaded and not parallel:
        # TRAIN-mode gradients of the cascaded tail from the reference's own modules: batch-statistics Kw_BatchNorm, straight-through VQ
        # (my_vector_quantizer.py:133-141), gradients through the frozen CLIP text tower.  Attention dropout is set to 0 (its RNG stream is
        # not reproducible elsewhere); HuBERT / CLIP stay in eval mode and frozen.
        cb = model.cascaded_branch
        bn_before = {k: getattr(cb.bn_layer.bn_layer, k).clone() for k in ("running_mean", "running_var", "num_batches_tracked")}
        cb.train()
        model.clip.eval()
        cb.self_att.multihead_attn_layer.dropout = 0.0
        model.zero_grad()
        losses_g, _, others_g = model.forward(batch)
        loss_g = model.compute_loss(losses_g)["loss"]
        loss_g.backward()
        arrays["train/cascaded_audio_feat"] = losses_g["cascaded_audio_feat"].detach().numpy().copy()
        arrays["train/loss"] = np.float64(loss_g.item())
        arrays["train/vq_targets"] = others_g["vq_results"]["targets"].numpy().copy()
        if plant_margins:       # train-mode BatchNorm uses the same batch statistics the running buffers were set to: same decisive targets
            assert torch.equal(others_g["vq_results"]["targets"], others["vq_results"]["targets"])
        n_grad = 0
        for k, prm in model.named_parameters():
            if prm.grad is not None and ((k.startswith("cascaded_branch.") and not k.startswith("cascaded_branch.clip.")) or
                                         k == "audio_encoder.weightedsum_layer.weights"):
                arrays["grad/" + k] = prm.grad.detach().numpy().copy()
                n_grad += 1
        assert n_grad == 12, n_grad   # cls, in_proj w/b, out_proj w/b, norm w/b, linear_proj w/b, bn w/b + the layer-mix weights
        for k in ("running_mean", "running_var", "num_batches_tracked"):
            arrays["train/bn_" + k] = getattr(cb.bn_layer.bn_layer, k).detach().numpy().copy()
            getattr(cb.bn_layer.bn_layer, k).copy_(bn_before[k])      # state_dict() aliases the live buffers: keep the saved "sd/" arrays pre-step
        # the oracle's autograd in train mode (same weights, BN buffers as BEFORE the reference's step) must agree
        sc.zero_grad()
        sc.cascaded_branch.train()
        sc.clip.eval()
        sc.cascaded_branch.load_state_dict({k[len("cascaded_branch."):]: v for k, v in sd.items()
                                            if k.startswith("cascaded_branch.") and not k.startswith("cascaded_branch.clip.")
                                            and "vector_quantizer" not in k})
        feat_o = speechclip_ref.weighted_sum([h.detach() for h in sc.forward_audio(batch["wav"], batch["wav_len"])[2]], sc.ws_weights, normalize_hiddenstates)
        ca, _, _ = sc.cascaded_branch(feat_o, o["audio_len"])
        loss_o = speechclip_ref.masked_contrastive_loss(speechclip_ref.l2_normalize(ca), o["image_feat"], batch["id"], sc.inv_temperature)
        assert abs(loss_o.item() - loss_g.item()) < 1e-5, (loss_o.item(), loss_g.item())
        loss_o.backward()
        for k, prm in sc.cascaded_branch.named_parameters():
            if k.startswith("clip.") or prm.grad is None:
                continue
            ref_g = torch.from_numpy(arrays["grad/cascaded_branch." + k])
            assert torch.allclose(prm.grad, ref_g, atol=2e-5 + 1e-3 * ref_g.abs().max().item(), rtol=1e-3), (k, (prm.grad - ref_g).abs().max())
        sc.cascaded_branch.eval()
        cb.eval()
    for k, v in np_state(sd).items():
        if k.startswith("criterion.") or k.startswith("cascaded_branch.clip."):
            continue
        arrays["sd/" + k] = v
    save(f"e2e_{tag}.npz", **arrays)


def gen_feat_len_table(kwclip_mod, OrderedNamespace):
    """feat_len (round-half-even of len/320, clamp T) and the HuBERT-internal frame mask for mixed batches."""
    STATE["hubert_cfg"], STATE["clip_cfg"] = hubert_ref.HubertRefConfig.tiny(), clip_ref.ClipRefConfig.tiny()
    from avssl.module.speech_encoder_plus import FairseqSpeechEncoder_Hubert
    enc = FairseqSpeechEncoder_Hubert("hubert", pretrained=True, feat_select_idx="weighted_sum").eval()
    rows = []
    for lens in ([400, 480, 16000], [800, 1120, 1440, 1760, 8160, 15840, 16000], [80000, 102400, 159999, 160000],
                 [480, 480], [4000, 12345, 33333, 40000]):
        wav = torch.zeros(len(lens), max(lens))
        with torch.no_grad():
            feat, flen = enc(wav, torch.tensor(lens))
            padded, mask = enc.preprocess_input([wav[i, :l] for i, l in enumerate(lens)])
            T = feat.shape[1]
            fm = enc.encoder.forward_padding_mask(torch.zeros(len(lens), T, 1), mask)
        for i, l in enumerate(lens):
            rows.append((max(lens), l, T, int(flen[i]), int((~fm[i]).sum())))
    save("feat_len.npz", table=np.array(rows, dtype=np.int64))


def gen_norm_methods():
    """`normalize_hiddenstates: true` with `normalize_type: method1 / method2` (speech_encoder_plus.py:572-592; no shipped YAML uses them, the shipped large
    configs use "s3prl" = per-feature layer_norm inside WeightedSumLayer): the reference's OWN FairseqSpeechEncoder_Hubert.forward on the weights and waves of
    the e2e_tiny_{base,large}_p fixtures.  Stored: the normalised hidden states the reference returns (first / last) and the mixed frames."""
    from avssl.module.speech_encoder_plus import FairseqSpeechEncoder_Hubert
    out = {}
    for tag, large in (("tiny_base_p", False), ("tiny_large_p", True)):
        g = np.load(os.path.join(HERE, f"e2e_{tag}.npz"))
        STATE["hubert_cfg"] = hubert_ref.HubertRefConfig.tiny(layer_norm_first=large, extractor_mode="layer_norm" if large else "default", conv_bias=large)
        STATE["clip_cfg"] = clip_ref.ClipRefConfig.tiny()
        sd = {k[len("sd/audio_encoder."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/audio_encoder.")}
        wav, wav_len = torch.from_numpy(g["wav"]), torch.from_numpy(g["wav_len"])
        for method in ("method1", "method2"):
            enc = FairseqSpeechEncoder_Hubert("hubert_large_ll60k" if large else "hubert", pretrained=True, feat_select_idx="weighted_sum",
                                              normalize_hiddenstates=True, normalize_type=method).eval()
            missing, unexpected = enc.load_state_dict(sd, strict=False)
            assert not unexpected and all("mask_emb" in k or "final_proj" in k or "label_embs" in k for k in missing), (missing, unexpected)
            assert enc.weightedsum_layer.normalize_features is False          # only "s3prl" normalises inside the layer mix (:472-476)
            with torch.no_grad():
                feat, flen, hidden = enc(wav, wav_len, return_hidden_states=True)
            # the standalone oracle restatement must agree
            hs = speechclip_ref.normalize_hidden_states([h.clone() for h in hidden_raw(enc, wav, wav_len)], method)
            for a, b in zip(hs, hidden):
                assert torch.allclose(a, b, atol=1e-6), method
            out[f"{tag}/{method}/feat"] = feat.numpy()
            out[f"{tag}/{method}/hidden_0"] = hidden[0].numpy()
            out[f"{tag}/{method}/hidden_last"] = hidden[-1].numpy()
            out[f"{tag}/{method}/feat_len"] = flen.numpy()
    save("norm_methods.npz", **out)


def hidden_raw(enc, wav, wav_len):
    """The un-normalised hidden states of the same encoder (normalisation switched off for one call)."""
    enc.normalize_hiddenstates = False
    try:
        with torch.no_grad():
            return enc(wav, wav_len, return_hidden_states=True)[2]
    finally:
        enc.normalize_hiddenstates = True


def main():
    only_analysis = "--only-analysis" in sys.argv
    install_stubs()
    if "--only-norm-methods" in sys.argv:      # adds tests/golden/norm_methods.npz without touching the other fixtures
        return gen_norm_methods()
    import avssl.module.losses as losses_mod
    import avssl.module.retrieval as retrieval_mod
    import avssl.module.weighted_sum as ws_mod
    import avssl.util.data_utils as du_mod
    from avssl.base import OrderedNamespace
    import avssl.model.kwClip as kwclip_mod

    if not only_analysis:
        gen_loss(losses_mod)
        gen_retrieval(retrieval_mod)
        gen_small_ops(ws_mod, du_mod)
        gen_feat_len_table(kwclip_mod, OrderedNamespace)
    tiny_b = hubert_ref.HubertRefConfig.tiny()
    tiny_l = hubert_ref.HubertRefConfig.tiny(layer_norm_first=True, extractor_mode="layer_norm", conv_bias=True)
    tiny_clip = clip_ref.ClipRefConfig.tiny()
    if not only_analysis:
        gen_end_to_end(kwclip_mod, OrderedNamespace, "tiny_base_p", tiny_b, tiny_clip, cascaded=False, parallel=True,
                       lens=[8000, 5000, 6777, 1200], normalize_hiddenstates=False)
        gen_end_to_end(kwclip_mod, OrderedNamespace, "tiny_large_p", tiny_l, tiny_clip, cascaded=False, parallel=True,
                       lens=[6400, 8000, 3999], normalize_hiddenstates=True)
    vocab = np.array([0, 320, 510, 511] + list(range(5, 300, 3)))
    gen_end_to_end(kwclip_mod, OrderedNamespace, "tiny_base_c", tiny_b, tiny_clip, cascaded=True, parallel=False,
                   lens=[8000, 5000, 6777, 1200], normalize_hiddenstates=False, reduce_vocab_ids=vocab, analysis_only=only_analysis)
    gen_end_to_end(kwclip_mod, OrderedNamespace, "tiny_base_c2", tiny_b, tiny_clip, cascaded=True, parallel=False,
                   lens=[8000, 7600, 7777, 7100], normalize_hiddenstates=False, reduce_vocab_ids=vocab, plant_margins=True,
                   analysis_only=only_analysis)


if __name__ == "__main__":
    main()

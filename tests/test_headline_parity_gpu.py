"""Oracle parity AT THE BENCHMARKED SIZES (VERDICT r3 "weak 1": until round 4 the fp32 oracle was only compared at B = 3, T <= 99; everything at
T = 499 / B = 256 was property-only, and garbage embeddings also give loss ~ ln 256).

Each test runs the HIP path on the FULL batch of the configuration exactly as bench.py builds it (bench.build_model(), bench.make_batch()),
then runs the fp32 CPU oracle -- same weights -- on a SUBSET of utterances spread over the batch (first, last, both sides of every 64-row
boundary, the longest utterance so that the padded length, hence the GroupNorm statistics and the frame-mask chunking, are the batch's:
speech_encoder_plus.py:506-518 and fairseq's forward_padding_mask), and asserts, per utterance of the subset,

  * `hidden_last` (last encoder layer, frames below feat_len): cosine >= 0.998                                  (speech_encoder_plus.py:29-64)
  * `parallel_audio_feat` / `image_feat`: CENTRED cosine >= 0.99 with the rotated-rows negative control (tests/helpers.py)    (kwClip.py:1385-1478)
    (P-large audio: >= 0.985 -- 24 bf16 layers on ragged utterances measure 0.9899-0.9988 per row; the CLS-row head itself is fp32-grade since
    round 4 (HIP head on the oracle's frames: 1.0000), what is left is the bf16 tower: feeding the HIP frames to the ORACLE's head gives the same
    0.9966 the whole HIP path reaches at P-base: tools/parity_diag.py)
  * masked InfoNCE of the HIP embeddings of the subset == oracle loss on the oracle's embeddings (<= 2e-2)          (losses.py:185-245)
  * C-base: the keyword scores ahead of the arg-max, VQ targets agreement, and the embedding where all 8 keywords agree.

Configurations: BASELINE.json configs[1] P-base B = 256 x 160000 samples (the bench line's workload), configs[4] P-large at B_local = 64
(model_large/coco/spchclp_p.yaml:10: batch 256 over 4 GPUs) with ragged 2-15 s utterances, configs[2] C-base B = 256."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from helpers import assert_rows_match, centred_cos  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _oracle_threads():
    """torch's CPU kernels collapse at 256 threads on the GPU box (bench.py's probe picks 16)."""
    old = torch.get_num_threads()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    yield
    torch.set_num_threads(old)


def _share(model, ref, parallel=True, cascaded=False):
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
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


def _spread(B, lens):
    """first, last, both sides of every 64-row boundary, + the longest (keeps the batch's padded length) and the shortest utterance."""
    idx = {0, 1, B - 2, B - 1, max(range(B), key=lambda i: lens[i]), min(range(B), key=lambda i: lens[i])}
    for b in range(64, B, 64):
        idx |= {b - 1, b}
    return sorted(idx)


def _nontrivial_(model, seed):
    """Random-init leaves every norm affine at (1, 0) and the layer-mix weights at 0: perturb them so a dropped affine / a wrong mix shows."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        w = model.audio_encoder.weightedsum_layer.weights
        w.copy_(0.5 * torch.randn(w.shape, generator=g))
        for m in model.modules():
            if isinstance(m, (torch.nn.LayerNorm, torch.nn.GroupNorm)) and m.weight is not None:
                m.weight.add_(0.1 * torch.randn(m.weight.shape, generator=g))
                m.bias.add_(0.1 * torch.randn(m.bias.shape, generator=g))


def _hip_forward(model, batch):
    from speechclip_amd import ops
    ops.bump_param_epoch()
    with torch.no_grad():
        _, flen, hidden = model.forward_audio(batch["wav"], batch["wav_len"], return_hidden_states=True)
        last = hidden[-1]
        lf, lm, others = model(batch)
    return flen, last, lf, lm, others


def _check_subset(tag, idx, flen_hip, last_hip, lf, key, o_feat, o_flen, o_last, o_img, ids, inv_t, hip_loss_fn, ref_loss, min_ccos=0.99):
    assert torch.equal(flen_hip[idx].cpu().long(), o_flen.long()), (tag, "feat_len", flen_hip[idx].tolist(), o_flen.tolist())
    worst_h = 1.0
    for j, b in enumerate(idx):
        n = int(o_flen[j])
        c = F.cosine_similarity(last_hip[b, :n].float().cpu().reshape(1, -1), o_last[j, :n].reshape(1, -1)).item()
        worst_h = min(worst_h, c)
        assert c >= 0.998, (tag, "hidden_last cosine", b, c)
    cc_i = assert_rows_match(lf["image_feat"][idx], o_img, 0.99, f"{tag} image_feat")
    cc_a = assert_rows_match(lf[key][idx], o_feat, min_ccos, f"{tag} {key}")
    hip_loss = hip_loss_fn(lf[key][idx].float().contiguous(), lf["image_feat"][idx].float().contiguous(), ids)
    assert abs(hip_loss - ref_loss) <= 2e-2, (tag, hip_loss, ref_loss)
    logit_err = ((lf[key][idx].float().cpu() @ lf["image_feat"][idx].float().cpu().t() - o_feat @ o_img.t()) * inv_t).abs().max().item()
    print(f"{tag}: utterances {idx}: hidden_last min cos {worst_h:.5f}; centred cos audio min {cc_a.min().item():.4f} image min {cc_i.min().item():.4f}; "
          f"subset loss hip {hip_loss:.5f} oracle {ref_loss:.5f}; max |logit diff| {logit_err:.4f}")
    return cc_a, cc_i


def test_p_base_headline_b256_vs_oracle():
    """BASELINE.json configs[1]: THE bench line's batch -- 256 pairs x 160000 samples (T = 499) -- against the fp32 oracle on 10 utterances."""
    import bench
    from oracle.clip_ref import ClipRefConfig
    from oracle.hubert_ref import HubertRefConfig
    from oracle.speechclip_ref import SpeechClipRef, l2_normalize
    from speechclip_amd import ops
    model = bench.build_model()
    _nontrivial_(model, 11)
    ref = SpeechClipRef(HubertRefConfig.base(), ClipRefConfig.vit_b32(), parallel=True, branch_heads=8).eval()
    _share(model, ref)
    model = model.cuda()
    B, L = 256, 160000
    batch, lens = bench.make_batch(B, L, 0, "cuda")
    flen, last, lf, lm, _ = _hip_forward(model, batch)
    assert last.shape == (B, 499, 768) and lf["parallel_audio_feat"].shape == (B, 512)
    loss_full = model.compute_loss(lf)["loss"].item()
    idx = _spread(B, lens)
    assert len(idx) >= 8
    sub = {k: v[idx].cpu() for k, v in batch.items()}
    with torch.no_grad():
        feat, o_flen, hidden = ref.forward_audio(sub["wav"], sub["wav_len"])
        o_img = l2_normalize(ref.clip.encode_image(sub["image"]))
        o_par = l2_normalize(ref.parallel_branch(feat, o_flen))
    ref_loss = ref.compute_loss({"parallel_audio_feat": o_par, "image_feat": o_img, "id": sub["id"]})["loss"].item()
    _check_subset("P-base B=256 T=499", idx, flen, last, lf, "parallel_audio_feat", o_par, o_flen, hidden[-1], o_img, sub["id"].cuda(), 1 / 0.07,
                  lambda a, i, ids: ops.infonce(a, i, ids)[0].item(), ref_loss)
    # the FULL-batch loss is the InfoNCE of the full-batch embeddings (fp32 oracle loss on the HIP embeddings), not merely "about ln 256"
    from oracle.speechclip_ref import masked_contrastive_loss
    want = masked_contrastive_loss(lf["parallel_audio_feat"].float().cpu(), lf["image_feat"].float().cpu(), batch["id"].cpu()).item()
    assert abs(loss_full - want) < 1e-4, (loss_full, want)


def test_p_large_b64_ragged_vs_oracle():
    """BASELINE.json configs[4] at the per-GPU batch the reference's large config implies (model_large/coco/spchclp_p.yaml:10: 256 over 4 GPUs =
    64): HuBERT-large + ViT-L/14, ragged 2-15 s utterances on the padding-free engine, against the fp32 oracle on 8+ utterances."""
    import bench
    from oracle.clip_ref import ClipRefConfig
    from oracle.hubert_ref import HubertRefConfig
    from oracle.speechclip_ref import SpeechClipRef, l2_normalize
    from speechclip_amd import ops
    model = bench.build_model(large=True)
    _nontrivial_(model, 12)
    inv_t = float(model.criterion.current_temperature)
    ref = SpeechClipRef(HubertRefConfig.large(), ClipRefConfig.vit_l14(), parallel=True, branch_heads=8, normalize_hiddenstates=True,
                        inv_temperature=inv_t).eval()
    _share(model, ref)
    model = model.cuda()
    B = 64
    g = torch.Generator().manual_seed(21)
    lens = [int(x) for x in torch.randint(32000, 240001, (B,), generator=g)]
    lens[37] = 240000
    wav = torch.zeros(B, max(lens))
    for i, n in enumerate(lens):
        wav[i, :n] = 0.1 * torch.randn(n, generator=g) + 0.01
    batch = {"wav": wav.cuda(), "wav_len": torch.tensor(lens).cuda(), "image": torch.randn(B, 3, 224, 224, generator=g).cuda(),
             "id": (torch.arange(B) // 2).cuda()}
    flen, last, lf, lm, _ = _hip_forward(model, batch)
    assert lf["parallel_audio_feat"].shape == (B, 768)
    idx = sorted(set(_spread(B, lens)) | {15, 16, 31, 32, 47, 48})
    sub = {k: v[idx].cpu() for k, v in batch.items()}
    with torch.no_grad():
        feat, o_flen, hidden = ref.forward_audio(sub["wav"], sub["wav_len"])
        o_img = l2_normalize(ref.clip.encode_image(sub["image"]))
        o_par = l2_normalize(ref.parallel_branch(feat, o_flen))
    ref_loss = ref.compute_loss({"parallel_audio_feat": o_par, "image_feat": o_img, "id": sub["id"]})["loss"].item()
    _check_subset("P-large B=64 ragged", idx, flen, last, lf, "parallel_audio_feat", o_par, o_flen, hidden[-1], o_img, sub["id"].cuda(), inv_t,
                  lambda a, i, ids: ops.infonce(a, i, ids, inv_temperature=inv_t)[0].item(), ref_loss, min_ccos=0.99)
    # (round 6: the suite-wide floor.  Rounds 4-5 asserted 0.985 here -- 24 pre-LN layers of bf16 GEMM / attention operands measured 0.9899-0.9901 -- until the
    #  layers moved to IEEE-half operands, the reference's own GPU precision for this model (spchclp_p.yaml:122): 0.9960 on this batch; SC_PRELN_F16=0 restores bf16.)


def test_c_base_b256_vs_oracle():
    """BASELINE.json configs[2]: Cascaded SpeechCLIP base, reduced vocabulary of 8112 sub-words, B = 256, 10 s audio with some shorter utterances."""
    import bench
    from oracle.clip_ref import ClipRefConfig
    from oracle.hubert_ref import HubertRefConfig
    from oracle.speechclip_ref import SpeechClipRef
    from speechclip_amd import ops
    model = bench.build_model(cascaded=True, vocab=8112)
    _nontrivial_(model, 13)
    g = torch.Generator().manual_seed(31)
    with torch.no_grad():
        bn = model.cascaded_branch.bn_layer.bn_layer
        bn.running_mean.copy_(0.05 * torch.randn(bn.running_mean.shape, generator=g))
        bn.running_var.copy_(1.0 + 0.2 * torch.rand(bn.running_var.shape, generator=g))
    ref = SpeechClipRef(HubertRefConfig.base(), ClipRefConfig.vit_b32(), parallel=False, cascaded=True,
                        reduced_vocab=bench.bench_vocab_ids(8112)).eval()
    _share(model, ref, parallel=False, cascaded=True)
    model = model.cuda()
    B, L = 256, 160000
    batch, lens = bench.make_batch(B, L, 0, "cuda")
    lens = list(lens)
    for i in range(3, B, 17):                      # some shorter utterances (zero right-padding, as collate_general hands them over)
        lens[i] = 48000 + 400 * i
        batch["wav"][i, lens[i]:] = 0
    batch["wav_len"] = torch.tensor(lens)
    batch["id"] = (torch.arange(B) // 2).cuda()
    flen, last, lf, lm, others = _hip_forward(model, batch)
    idx = sorted(set(_spread(B, lens)) | {3, 20})
    sub = {k: v[idx].cpu() for k, v in batch.items()}
    with torch.no_grad():
        o = ref(sub)
        _, o_flen, hidden = ref.forward_audio(sub["wav"], sub["wav_len"])
    tg_hip = others["vq_results"]["targets"][idx].reshape(len(idx), -1).cpu()
    tg_ref = o["vq_results"]["targets"].reshape(len(idx), -1)
    agree = (tg_hip == tg_ref)
    # hidden states, image tower, feat_len: exactly as in the parallel tests
    assert torch.equal(flen[idx].cpu().long(), o_flen.long())
    for j, b in enumerate(idx):
        n = int(o_flen[j])
        c = F.cosine_similarity(last[b, :n].float().cpu().reshape(1, -1), hidden[-1][j, :n].reshape(1, -1)).item()
        assert c >= 0.998, ("C-base hidden_last", b, c)
    assert_rows_match(lf["image_feat"][idx], o["image_feat"], 0.99, "C-base image_feat")
    # keywords ahead of the arg-max: the keyword embeddings that enter the text tower agree wherever the arg-max does; the arg-max over 8112 random
    # near-tied sub-words (top-1/top-2 margins ~1e-3 in the oracle itself) must agree on most keywords
    assert agree.float().mean().item() >= 0.85, agree.float().mean().item()
    same = agree.all(dim=1)
    assert int(same.sum()) >= 2, same.tolist()
    kw_hip, kw_ref = others["keywords"][idx].float().cpu(), o["keywords"]
    assert torch.allclose(kw_hip[agree], kw_ref[agree], atol=2e-2, rtol=2e-2)
    rows = [j for j in range(len(idx)) if same[j]]
    cc = centred_cos(lf["cascaded_audio_feat"][idx][rows], o["cascaded_audio_feat"][rows]) if len(rows) > 1 else torch.ones(1)
    raw = F.cosine_similarity(lf["cascaded_audio_feat"][idx][rows].float().cpu(), o["cascaded_audio_feat"][rows], dim=-1)
    assert raw.min().item() >= 0.999 and cc.min().item() >= 0.98, (raw.tolist(), cc.tolist())
    if len(rows) >= 2:
        ids = sub["id"][rows].cuda()
        hip_loss = ops.infonce(lf["cascaded_audio_feat"][idx][rows].float().contiguous(), lf["image_feat"][idx][rows].float().contiguous(), ids)[0].item()
        from oracle.speechclip_ref import masked_contrastive_loss
        ref_loss = masked_contrastive_loss(o["cascaded_audio_feat"][rows], o["image_feat"][rows], sub["id"][rows]).item()
        assert abs(hip_loss - ref_loss) <= 2e-2, (hip_loss, ref_loss)
    print(f"C-base B=256: utterances {idx}: VQ targets agree {agree.float().mean().item():.3f}, all-8 on {int(same.sum())}/{len(idx)}; "
          f"cascaded emb raw cos min {raw.min().item():.5f}, centred {cc.min().item():.4f}")

"""Padding-free (packed) engine vs the uniform padded layout, and vs the oracle.

The reference pads every utterance to the batch maximum and runs the conv stack and all transformer GEMMs on B x T_max rows
(avssl/data/collate_function.py:18-30, avssl/module/speech_encoder_plus.py:506-518, :540-556); only frames below each utterance's own
length reach an output (:604-611).  The packed engine (speechclip_amd/module/hubert.py: packed_geometry) runs the same kernels on
sum_b (frames_b + 1) rows.  Checked here:
  * packed == padded on every frame a head may read (all hidden states, the mixed features, the final embeddings), base and large layouts;
  * packed vs the fp32 oracle at the real base dimensions on a ragged batch (the padded path's own tolerance);
  * the train-mode form (frozen-encoder dropouts, packed attention with probability dropout) runs and keeps the valid frames finite;
  * a batch with one utterance much longer than the rest, single-utterance batches, and equal lengths under SC_VARLEN_PACK=1."""
import dataclasses
import os

import pytest
import torch
import torch.nn.functional as F

from helpers import assert_rows_match, make_config

pytestmark = pytest.mark.gpu


def _cos(a, b):
    return F.cosine_similarity(a.float().reshape(1, -1), b.float().reshape(1, -1)).item()


def _tiny(large=False, cascaded=False):
    from oracle.clip_ref import ClipRefConfig
    from oracle.hubert_ref import HubertRefConfig
    from speechclip_amd.model import KWClip_GeneralTransformer
    from speechclip_amd.module.clip_model import ClipConfig
    from speechclip_amd.module.hubert import HubertConfig
    href = HubertRefConfig.tiny(layer_norm_first=large, extractor_mode="layer_norm" if large else "default", conv_bias=large)
    cref = ClipRefConfig.tiny()
    cfg = make_config(d_model=128, branch_heads=4, hubert_config=HubertConfig(**dataclasses.asdict(href)),
                      clip_config=ClipConfig(**dataclasses.asdict(cref)), hubert_name="hubert_large_ll60k" if large else "hubert",
                      normalize_hiddenstates=large)
    torch.manual_seed(3)
    model = KWClip_GeneralTransformer(cfg)
    g = torch.Generator().manual_seed(4)
    with torch.no_grad():
        model.audio_encoder.weightedsum_layer.weights.copy_(0.5 * torch.randn(href.encoder_layers + 1, generator=g))
        for m in model.audio_encoder.encoder.modules():
            if isinstance(m, torch.nn.LayerNorm):
                m.weight.add_(0.2 * torch.randn(m.weight.shape, generator=g)); m.bias.add_(0.1 * torch.randn(m.bias.shape, generator=g))
    return model.cuda().eval(), href, cref


def _batch(lens, res=64, seed=7, amp=0.3):
    g = torch.Generator().manual_seed(seed)
    wav = torch.zeros(len(lens), max(lens))
    for i, l in enumerate(lens):
        wav[i, :l] = amp * torch.randn(l, generator=g)
    return {"wav": wav.cuda(), "wav_len": torch.tensor(lens).cuda(), "image": torch.randn(len(lens), 3, res, res, generator=g).cuda(),
            "id": torch.arange(len(lens)).cuda()}


def _run(model, batch, pack):
    old = os.environ.get("SC_VARLEN_PACK")
    os.environ["SC_VARLEN_PACK"] = pack
    try:
        with torch.no_grad():
            feat, flen, hidden = model.forward_audio(batch["wav"], batch["wav_len"], return_hidden_states=True)
            mixed_only, flen2 = model.forward_audio(batch["wav"], batch["wav_len"])
            lf, _, _ = model(batch)
        return feat.float().cpu(), flen.cpu(), [h.float().cpu() for h in hidden], mixed_only.float().cpu(), {k: v.float().cpu() for k, v in lf.items()}
    finally:
        if old is None:
            os.environ.pop("SC_VARLEN_PACK")
        else:
            os.environ["SC_VARLEN_PACK"] = old


@pytest.mark.parametrize("large", [False, True])
@pytest.mark.parametrize("lens", [[8000, 6000, 3000, 8000], [16000, 1200, 900, 2000, 700, 5000], [5000], [4000, 4000, 4000]])
def test_packed_engine_equals_padded_engine(large, lens):
    model, _, _ = _tiny(large)
    batch = _batch(lens)
    enc = model.audio_encoder
    geo = enc.encoder.packed_geometry(lens, max(lens))
    f0, l0, h0, m0, e0 = _run(model, batch, "0")
    f1, l1, h1, m1, e1 = _run(model, batch, "1")
    assert torch.equal(l0, l1) and f0.shape == f1.shape and len(h0) == len(h1)
    for b, L in enumerate(l0.tolist()):
        assert L <= geo["rows"][b] - 1
        for li, (a, c) in enumerate(zip(h0, h1)):
            assert _cos(a[b, :L], c[b, :L]) > 0.9999, (b, li, _cos(a[b, :L], c[b, :L]))
            assert (a[b, :L] - c[b, :L]).abs().max().item() <= 2e-2 * a[b, :L].abs().max().item() + 1e-3
        assert _cos(f0[b, :L], f1[b, :L]) > 0.9999 and _cos(m0[b, :L], m1[b, :L]) > 0.9999
        assert torch.equal(f1[b, :L], m1[b, :L])                      # mixing packed rows == mixing unpacked states
    if len(lens) > 1:
        assert_rows_match(e1["parallel_audio_feat"], e0["parallel_audio_feat"], 0.999, "packed vs padded parallel_audio_feat")
    else:
        assert _cos(e1["parallel_audio_feat"], e0["parallel_audio_feat"]) > 0.99999
    assert (e1["parallel_audio_feat"] - e0["parallel_audio_feat"]).abs().max().item() < 5e-3


def test_packed_engine_at_base_dims_vs_oracle():
    """Real P-base dimensions, ragged batch (0.4 - 2.6 s): the packed engine against the fp32 oracle, same tolerances as the padded path."""
    from oracle.clip_ref import ClipRefConfig
    from oracle.hubert_ref import HubertRefConfig
    from oracle.speechclip_ref import SpeechClipRef
    from speechclip_amd.model import KWClip_GeneralTransformer
    torch.manual_seed(1)
    model = KWClip_GeneralTransformer(make_config()).eval()
    ref = SpeechClipRef(HubertRefConfig(), ClipRefConfig(), parallel=True, branch_heads=8).eval()
    sd = model.state_dict()
    ref.encoder.load_state_dict({k[len("audio_encoder.encoder."):]: v for k, v in sd.items() if k.startswith("audio_encoder.encoder.")})
    ref.clip.load_state_dict({k[len("clip.model."):]: v for k, v in sd.items() if k.startswith("clip.model.")})
    ref.parallel_branch.load_state_dict({k[len("parallel_branch."):]: v for k, v in sd.items() if k.startswith("parallel_branch.")})
    lens = [41000, 6500, 23000, 16000, 9000]
    batch = _batch(lens, res=224, seed=2, amp=0.2)
    o = ref({k: v.cpu() for k, v in batch.items()})
    model = model.cuda()
    geo = model.audio_encoder.encoder.packed_geometry(lens, max(lens))
    assert geo["total"] < 0.6 * geo["padded_rows"]
    f1, l1, h1, m1, e1 = _run(model, batch, "1")
    assert l1.tolist() == o["audio_len"].tolist()
    for b, L in enumerate(l1.tolist()):
        assert _cos(f1[b, :L], o["audio_feat"][b, :L]) > 0.998, (b, _cos(f1[b, :L], o["audio_feat"][b, :L]))
    cc = assert_rows_match(e1["parallel_audio_feat"], o["parallel_audio_feat"], 0.99, "packed parallel_audio_feat vs oracle")
    print("packed engine, base dims: centred cosine per row", cc.tolist(), "rows", geo["total"], "of", geo["padded_rows"])


def test_packed_engine_train_mode_dropouts():
    """Lightning's model.train() puts the frozen encoder's dropouts on (speech_encoder_plus.py:42, :87): the packed attention's dropout form
    and the row-wise dropouts run on packed rows; outputs stay finite, differ from eval, and two seeds differ."""
    model, _, _ = _tiny(False)
    batch = _batch([8000, 3000, 6000, 2500])
    old = {k: os.environ.get(k) for k in ("SC_VARLEN_PACK", "SC_FROZEN_DROPOUT")}
    os.environ["SC_VARLEN_PACK"] = "1"
    os.environ["SC_FROZEN_DROPOUT"] = "1"
    try:
        with torch.no_grad():
            ev = model.forward_audio(batch["wav"], batch["wav_len"])[0].float()
            model.audio_encoder.train()
            torch.manual_seed(0)
            a = model.forward_audio(batch["wav"], batch["wav_len"])[0].float()
            torch.manual_seed(1)
            b = model.forward_audio(batch["wav"], batch["wav_len"])[0].float()
            torch.manual_seed(0)
            a2 = model.forward_audio(batch["wav"], batch["wav_len"])[0].float()
    finally:
        model.audio_encoder.eval()
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert torch.isfinite(a).all() and torch.isfinite(b).all()
    assert torch.equal(a, a2), "same seed, same masks"
    assert (a - b).abs().max().item() > 1e-3 and (a - ev).abs().max().item() > 1e-3
    assert _cos(a, ev) > 0.8                          # dropout 0.1 perturbs, it does not destroy


def test_unpack_rows_kernel():
    from speechclip_amd import ops
    g = torch.Generator().manual_seed(0)
    rows = [5, 1, 9, 3]
    off = [0]
    for r in rows:
        off.append(off[-1] + r)
    src = torch.randn(3, off[-1], 64, generator=g).to(torch.bfloat16).cuda()
    out = ops.unpack_rows(src, torch.tensor(off, dtype=torch.int32).cuda(), 4, 7)
    assert out.shape == (3, 4, 7, 64)
    for b, r in enumerate(rows):
        n = min(r, 7)
        assert torch.equal(out[:, b, :n], src[:, off[b]:off[b] + n])
        assert out[:, b, n:].abs().max().item() == 0 if n < 7 else True
    one = ops.unpack_rows(src[0].float().contiguous(), torch.tensor(off, dtype=torch.int32).cuda(), 4, 9)
    assert one.shape == (4, 9, 64) and torch.equal(one[2, :9], src[0, off[2]:off[2] + 9].float())
    # halo = 1 (what the encoder passes): the last row of every utterance -- the packed engine's receptive-field row -- is NOT exposed
    hal = ops.unpack_rows(src, torch.tensor(off, dtype=torch.int32).cuda(), 4, 7, halo=1)
    for b, r in enumerate(rows):
        n = min(r - 1, 7)
        assert torch.equal(hal[:, b, :n], src[:, off[b]:off[b] + n])
        assert n == 7 or hal[:, b, n:].abs().max().item() == 0


def test_returned_hidden_states_do_not_alias_the_engine_workspace():
    """The reference returns fresh tensors (speech_encoder_plus.py:596-602): hidden states handed to the caller must survive a second forward on a
    different batch (the engine's `hidden` buffer is a reused workspace; until round 4 the padded path returned views of it)."""
    model, _, _ = _tiny(False)
    g = torch.Generator().manual_seed(3)
    w1, w2 = (0.1 * torch.randn(3, 6000, generator=g)).cuda(), (0.1 * torch.randn(3, 6000, generator=g)).cuda()
    ln = torch.tensor([6000, 6000, 6000]).cuda()
    with torch.no_grad():
        _, _, hid1 = model.forward_audio(w1, ln, return_hidden_states=True)
        keep = [h.clone() for h in hid1]
        _, _, hid2 = model.forward_audio(w2, ln, return_hidden_states=True)
    for a, b in zip(hid1, keep):
        assert torch.equal(a, b), "a later forward rewrote hidden states that were already returned"
    assert not torch.equal(hid1[-1], hid2[-1])


def test_workspaces_do_not_multiply_with_the_batch_maximum():
    """A job whose longest utterance differs from batch to batch must not keep a workspace set per distinct frame count (round 3: the padded
    engine's shape-keyed buffers grew by 107 GB over ten distinct T at B = 256): every large buffer is one growing flat allocation per name."""
    model, _, _ = _tiny(False)
    enc = model.audio_encoder.encoder
    sizes = []
    for pack in ("0", "1"):
        for lmax in (8000, 7000, 7680, 6100, 5000, 7990, 6400):
            batch = _batch([lmax, lmax - 900, 2500, lmax - 10], seed=lmax)
            old = os.environ.get("SC_VARLEN_PACK")
            os.environ["SC_VARLEN_PACK"] = pack
            try:
                with torch.no_grad():
                    model.forward_audio(batch["wav"], batch["wav_len"])
            finally:
                if old is None:
                    os.environ.pop("SC_VARLEN_PACK")
                else:
                    os.environ["SC_VARLEN_PACK"] = old
            sizes.append((len(enc._ws), sum(t.numel() * t.element_size() for t in enc._ws.values())))
    n_first, b_first = sizes[0]
    assert sizes[-1][0] <= n_first + 2, sizes                     # no new entries per shape
    assert sizes[-1][1] <= 1.3 * max(s[1] for s in sizes[:1]) + 4096, sizes      # the largest batch came first: capacity never had to grow

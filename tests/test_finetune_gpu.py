"""Fine-tuning HuBERT transformer layers (speech_encoder_plus.py:416-446 `trainable` + `unfreeze_layers` /
`reinit_layers`): every backward kernel against torch autograd in fp32 on the same bf16 operands, then the whole chain -- loss ->
pooling head -> layer mix -> trained encoder layers -- against the oracle's autograd, and a short fine-tuning run."""
import dataclasses

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _g(seed):
    return torch.Generator().manual_seed(seed)


def _cos(a, b):
    return F.cosine_similarity(a.double().reshape(1, -1).cpu(), b.double().reshape(1, -1).cpu()).item()


@pytest.mark.parametrize("R,C,Rp,Z", [(500, 64, 512, 3), (70, 768, 128, 1), (64, 64, 64, 2), (1, 5, 64, 1)])
def test_transpose_bf16_strided_batched_padded(R, C, Rp, Z):
    from speechclip_amd import ops
    g = _g(R + C)
    ld = C + 16
    src = torch.randn(Z, R, ld, generator=g).to(BF).cuda()
    out = ops.transpose_bf16(src, ld, R * ld, R, C, Z, rows_padded=Rp)
    assert out.shape == (Z, C, Rp)
    assert torch.equal(out[:, :, :R], src[:, :, :C].transpose(1, 2)) and bool((out[:, :, R:] == 0).all())
    if C >= 8:          # a view that starts 2 elements in (4-byte aligned only): the scalar form must take over, same result
        sub = src.view(-1)[2:]
        o2 = ops.transpose_bf16(sub, ld, R * ld, R, C - 4, Z, rows_padded=Rp)
        assert torch.equal(o2[:, :, :R], src[:, :, 2:C - 2].transpose(1, 2))
    # an overlapping-row (sliding-window) view: ld_in < cols
    if R >= 8:
        flat = src[0].reshape(-1)[: (R + 16) * 8].contiguous()
        ov = ops.transpose_bf16(flat, 8, 0, R, 64, 1, rows_padded=Rp)[0]                 # row r = flat[8 r : 8 r + 64]
        ref = torch.stack([flat[8 * r: 8 * r + 64] for r in range(R)], 1)
        assert torch.equal(ov[:, :R], ref)


def test_gelu_bwd_colsum_axpy_bf16():
    from speechclip_amd import ops
    g = _g(3)
    u = (2 * torch.randn(1000, 256, generator=g)).to(BF)
    dh = torch.randn(1000, 256, generator=g).to(BF)
    uf = u.float().requires_grad_(True)
    F.gelu(uf).backward(dh.float())
    torch.testing.assert_close(ops.gelu_bwd_bf16(u.cuda(), dh.cuda()).float().cpu(), uf.grad, atol=2e-2, rtol=2e-2)
    ut, dt = u.reshape(-1)[:4102].contiguous(), dh.reshape(-1)[:4102].contiguous()          # 4102 = 8 * 512 + 6: the pairwise tail
    torch.testing.assert_close(ops.gelu_bwd_bf16(ut.cuda(), dt.cuda()).float().cpu(), uf.grad.reshape(-1)[:4102], atol=2e-2, rtol=2e-2)
    x = torch.randn(70000, 96, generator=g).to(BF)
    torch.testing.assert_close(ops.colsum_bf16(x.cuda()).cpu(), x.float().sum(0), atol=0.05, rtol=1e-3)
    y = torch.randn(4096, generator=g).to(BF).cuda()
    y0 = y.clone()
    ops.axpy_bf16(y, x.cuda().reshape(-1)[:4096].contiguous(), 0.37)
    torch.testing.assert_close(y.float(), y0.float() + 0.37 * x.reshape(-1)[:4096].float().cuda(), atol=2e-2, rtol=2e-2)


@pytest.mark.parametrize("rows,D", [(301, 768), (70000, 128), (7, 64), (70000, 512), (5000, 1024), (33, 256)])
def test_layernorm_bwd_bf16_vs_autograd(rows, D):
    from speechclip_amd import ops
    g = _g(rows)
    x = (1.5 * torch.randn(rows, D, generator=g) + 0.3).to(BF)
    dy = torch.randn(rows, D, generator=g).to(BF)
    gamma = 1 + 0.3 * torch.randn(D, generator=g)
    xf = x.float().requires_grad_(True)
    gm = gamma.clone().requires_grad_(True)
    bt = torch.zeros(D, requires_grad=True)
    F.layer_norm(xf, (D,), gm, bt, 1e-5).backward(dy.float())
    dx, dg, db = ops.layernorm_bwd_bf16(x.cuda(), dy.cuda(), gamma.cuda())
    torch.testing.assert_close(dx.float().cpu(), xf.grad, atol=3e-2, rtol=3e-2)
    assert _cos(dg, gm.grad) > 0.9999 and abs(dg.norm().item() / gm.grad.norm().item() - 1) < 2e-3
    assert _cos(db, bt.grad) > 0.9999 and abs(db.norm().item() / bt.grad.norm().item() - 1) < 2e-3


@pytest.mark.parametrize("M,N,K", [(128000 // 16, 768, 768), (5000, 2304, 128), (300, 64, 256), (48, 128, 128)])
def test_weight_gradient_split_k(M, N, K):
    from speechclip_amd.train_hubert import wgrad
    g = _g(M + N)
    dy = (0.3 * torch.randn(M, N, generator=g)).to(BF)
    x = (0.5 * torch.randn(M, K, generator=g)).to(BF)
    got = wgrad(dy.cuda(), x.cuda()).cpu()
    ref = dy.float().t() @ x.float()
    assert got.shape == (N, K) and got.dtype == torch.float32
    torch.testing.assert_close(got, ref, atol=2e-3 * ref.abs().max().item() + 1e-4, rtol=2e-3)


@pytest.mark.parametrize("B,T,H,lens", [(3, 70, 2, [70, 33, 7]), (2, 500, 12, [500, 321]), (2, 25, 2, [25, 25])])
def test_attention_backward_vs_autograd(B, T, H, lens):
    """dqkv of softmax(Q K^T / 8 + key mask) V from (qkv, dO): recomputed S / dP (batched MFMA GEMMs), sc_attn_softmax_bwd, TN products over
    transposed operands -- against fp32 autograd on the same bf16 qkv."""
    from speechclip_amd import ops
    from speechclip_amd.train_hubert import attention_bwd
    g = _g(B * T + H)
    d = H * 64
    Lp = -(-T // 64) * 64
    qkv = torch.zeros(B * T + (Lp - T), 3 * d, dtype=BF)
    qkv[:B * T] = torch.randn(B * T, 3 * d, generator=g).to(BF)
    dO = torch.randn(B * T, d, generator=g).to(BF)
    for b, n in enumerate(lens):
        dO[b * T + n:(b + 1) * T] = 0                       # padded query rows carry no gradient (nothing reads them)
    klens = torch.tensor(lens, dtype=torch.int32)
    x = qkv[:B * T].float().clone().requires_grad_(True)
    xx = x.view(B, T, 3, H, 64)
    q, k, v = xx[:, :, 0].transpose(1, 2), xx[:, :, 1].transpose(1, 2), xx[:, :, 2].transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * 0.125
    mask = torch.arange(T)[None, :] >= klens[:, None]
    s = s.masked_fill(mask[:, None, None, :], float("-inf"))
    o = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * T, d)
    o.backward(dO.float())
    att = ops.attention(qkv[:B * T].cuda().contiguous(), B, T, H, klens.cuda())
    torch.testing.assert_close(att.float().cpu()[dO.float().abs().sum(1) > 0], o.detach()[dO.float().abs().sum(1) > 0], atol=2e-2, rtol=2e-2)
    dqkv = attention_bwd(qkv.cuda(), att, dO.cuda(), B, T, H, klens.cuda()).float().cpu()
    ref = x.grad
    valid_rows = torch.cat([torch.arange(T) < n for n in lens])
    err = (dqkv - ref)[valid_rows].abs().max().item()
    assert err < 3e-2 * max(1.0, ref.abs().max().item()), err
    assert _cos(dqkv[valid_rows], ref[valid_rows]) > 0.999
    # keys beyond the mask receive no gradient; their query rows had dO = 0
    assert dqkv[~valid_rows][:, d:].abs().max().item() < 1e-6 if (~valid_rows).any() else True


def test_cls_pool_dz_matches_autograd():
    """Gradient of the mixed frames out of the pooling head's backward (sc_cls_pool_dz on sc_cls_pool_bwd's workspaces) vs autograd of the explicit
    pooling reference."""
    from speechclip_amd import ops
    from test_train_kernels_gpu import _pool_reference
    g = _g(12)
    B, T, D, NQ, H = 4, 37, 768, 1, 8
    R = NQ * H
    lens = torch.tensor([37, 20, 5, 31])
    x16 = torch.randn(B, T, D, generator=g).to(BF)
    cls = torch.randn(NQ, D, generator=g)
    u = 0.05 * torch.randn(R, D, generator=g)
    beta = 0.1 * torch.randn(R, generator=g)
    xf = x16.float().clone().requires_grad_(True)
    _, zbar_ref = _pool_reference(xf, cls, u, beta, lens, NQ, H)
    dzbar = torch.randn(B, R, D, generator=g)
    (zbar_ref * dzbar).sum().backward()
    rows = x16.cuda().view(B * T, D)
    lens_i = lens.to(torch.int32).cuda()
    scores = ops.gemm(rows, u.to(BF).cuda().contiguous(), beta.cuda(), out_f32=True)
    cls_scores = (cls @ u.t() + beta).cuda().contiguous()
    p, zbar = ops.cls_pool_train_fwd(rows, cls.cuda(), scores, cls_scores, lens_i, B, T, NQ, R, D)
    du, dck, dalpha, ds_ws, pp_ws = ops.cls_pool_bwd(rows, cls.cuda(), None, p, dzbar.cuda(), u.cuda(), lens_i, B, T, NQ, R, D, return_ws=True)
    dz = ops.cls_pool_dz(pp_ws, ds_ws, dzbar.cuda(), u.cuda(), lens_i, B, T, NQ, R, D).float().cpu().view(B, T, D)
    ref = xf.grad
    for b, n in enumerate(lens.tolist()):
        assert _cos(dz[b, :n], ref[b, :n]) > 0.999 and (dz[b, :n] - ref[b, :n]).abs().max().item() < 3e-2 * ref.abs().max().item() + 1e-3
        assert dz[b, n:].abs().max().item() == 0 if n < T else True


def _finetune_pair(train_layers, reinit=False, everything=False, large=False):
    """Tiny P-base model with the listed encoder layers trainable + the oracle with the same weights."""
    from helpers import make_config
    from oracle.clip_ref import ClipRefConfig
    from oracle.hubert_ref import HubertRefConfig
    from oracle.speechclip_ref import SpeechClipRef
    from speechclip_amd.model import KWClip_GeneralTransformer
    from speechclip_amd.module.clip_model import ClipConfig
    from speechclip_amd.module.hubert import HubertConfig
    tiny = HubertRefConfig.tiny(layer_norm_first=True, extractor_mode="layer_norm", conv_bias=True) if large else HubertRefConfig.tiny()
    href, cref = dataclasses.replace(tiny, encoder_layers=3), ClipRefConfig.tiny()
    hc = HubertConfig(**dataclasses.asdict(href))
    if large:       # the released large checkpoint: no dropouts, feature_grad_mult 1
        hc = dataclasses.replace(hc, dropout=0.0, attention_dropout=0.0, dropout_input=0.0, encoder_layerdrop=0.0, feature_grad_mult=1.0)
    cfg = make_config(d_model=128, branch_heads=4, hubert_config=hc, clip_config=ClipConfig(**dataclasses.asdict(cref)),
                      hubert_name="hubert_large_ll60k" if large else "hubert", normalize_hiddenstates=large)
    cfg.audio_encoder.trainable = True
    if everything:
        pass                                                          # bare trainable: true -- no layer lists
    elif reinit:
        cfg.audio_encoder.reinit_layers = list(train_layers)
    else:
        cfg.audio_encoder.unfreeze_layers = list(train_layers)
    torch.manual_seed(5)
    model = KWClip_GeneralTransformer(cfg)
    g = _g(9)
    with torch.no_grad():
        model.audio_encoder.weightedsum_layer.weights.copy_(0.5 * torch.randn(4, generator=g))
        for m in model.audio_encoder.encoder.modules():
            if isinstance(m, torch.nn.LayerNorm):
                m.weight.add_(0.2 * torch.randn(m.weight.shape, generator=g)); m.bias.add_(0.1 * torch.randn(m.bias.shape, generator=g))
    ref = SpeechClipRef(href, cref, parallel=True, branch_heads=4, normalize_hiddenstates=large)
    sd = model.state_dict()
    ref.encoder.load_state_dict({k[len("audio_encoder.encoder."):]: v for k, v in sd.items() if k.startswith("audio_encoder.encoder.")})
    ref.clip.load_state_dict({k[len("clip.model."):]: v for k, v in sd.items() if k.startswith("clip.model.")})
    ref.parallel_branch.load_state_dict({k[len("parallel_branch."):]: v for k, v in sd.items() if k.startswith("parallel_branch.")})
    with torch.no_grad():
        ref.ws_weights.copy_(sd["audio_encoder.weightedsum_layer.weights"])
    lens = [8000, 5200, 8000, 3100]
    wav = torch.zeros(4, 8000)
    for i, l in enumerate(lens):
        wav[i, :l] = 0.3 * torch.randn(l, generator=g)
    batch = {"wav": wav, "wav_len": torch.tensor(lens), "image": torch.randn(4, 3, 64, 64, generator=g), "id": torch.tensor([1, 2, 3, 4])}
    return model, ref, batch


def test_trainable_flags_follow_the_reference():
    model, _, _ = _finetune_pair([1, 2])
    enc = model.audio_encoder
    names = [k for k, p in enc.encoder.named_parameters() if p.requires_grad]
    assert names and all(k.startswith(("encoder.layers.1.", "encoder.layers.2.")) for k in names)        # pos_conv, layer_norm, extractor, proj, layer 0: frozen
    assert len(names) == 2 * 16 and enc.encoder.feature_grad_mult == 0
    tp = model.getTrainableParams()
    assert sum(any(p is q for q in tp) for p in enc.encoder.parameters() if p.requires_grad) == 32
    from speechclip_amd.module import FairseqSpeechEncoder_Hubert
    full = FairseqSpeechEncoder_Hubert("hubert", trainable=True, hubert_config=enc.encoder.cfg)     # bare trainable: everything trains (train_front.py)
    assert full.train_front and full.train_layers == [0, 1, 2]
    with pytest.raises(AssertionError):
        FairseqSpeechEncoder_Hubert("hubert", trainable=False, unfreeze_layers=[1], hubert_config=enc.encoder.cfg)


@pytest.mark.parametrize("train_layers", [[2], [1, 2], [1]])
def test_finetune_gradients_vs_oracle_autograd(train_layers):
    """loss.backward() through head -> layer mix -> encoder layers L0.. on the HIP kernels vs the fp32 oracle's autograd (same weights, same batch):
    every trained encoder tensor, the branch and the mix weights."""
    from oracle import hubert_ref as HR
    from oracle import speechclip_ref as R
    model, ref, batch = _finetune_pair(train_layers)
    model = model.cuda().eval()                                  # eval(): no dropout in the branch; gradients still flow (grad mode is on)
    feats, _, _ = model({k: v.cuda() for k, v in batch.items()})
    loss = model.compute_loss(feats)["loss"]
    loss.backward()
    # oracle with autograd through the encoder layers
    for p in ref.parameters():
        p.requires_grad_(False)
    lys = ref.encoder.encoder.layers
    for i in train_layers:
        for p in lys[i].parameters():
            p.requires_grad_(True)
    for p in ref.parallel_branch.parameters():
        p.requires_grad_(True)
    ref.ws_weights.requires_grad_(True)
    wavs = [batch["wav"][b, :int(batch["wav_len"][b])] for b in range(4)]
    padded, mask = HR.preprocess_input(wavs, ref.hubert_cfg.normalize)
    with torch.enable_grad():
        out = HR.hubert_forward.__wrapped__(ref.encoder, padded, mask)
        hidden = out["layer_results"]
        flen = HR.feat_lengths([len(w) for w in wavs], 320, hidden[-1].shape[1])
        mixed = R.weighted_sum(hidden, ref.ws_weights, False)
        pa = R.l2_normalize(ref.parallel_branch(mixed, flen))
        with torch.no_grad():
            img = R.l2_normalize(ref.clip.encode_image(batch["image"]))
        ref_loss = R.masked_contrastive_loss(pa, img, batch["id"], ref.inv_temperature)
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) < 2e-2
    mine = dict(model.named_parameters())
    checked = 0
    for i in train_layers:
        for k, p in lys[i].named_parameters():
            got = mine[f"audio_encoder.encoder.encoder.layers.{i}.{k}"].grad
            assert got is not None, (i, k)
            if p.grad.norm().item() < 1e-7:
                assert got.norm().item() < 1e-4, (i, k)
                continue
            c, ratio = _cos(got, p.grad), got.norm().item() / p.grad.norm().item()
            assert c > 0.98 and abs(ratio - 1) < 0.1, (i, k, c, ratio)
            checked += 1
    assert checked >= 12 * len(train_layers)
    for k, p in ref.parallel_branch.named_parameters():
        got = mine["parallel_branch." + k].grad
        if p.grad.norm().item() > 1e-6:
            assert _cos(got, p.grad) > 0.98, k
    assert _cos(mine["audio_encoder.weightedsum_layer.weights"].grad, ref.ws_weights.grad) > 0.98
    # frozen parts have no gradient
    assert all(p.grad is None for k, p in mine.items() if k.startswith("audio_encoder.encoder.") and not any(f".layers.{i}." in k for i in train_layers))


@pytest.mark.parametrize("train_layers", [[2], [1, 2]])
def test_finetune_pre_ln_layers_vs_oracle_autograd(train_layers):
    """HuBERT-large style encoder (pre-LN layers on an fp32 residual stream, LayerNorm extractor, wave normalisation, normalize_hiddenstates in
    front of the mix): `unfreeze_layers` gradients vs the oracle's autograd."""
    from oracle import hubert_ref as HR
    from oracle import speechclip_ref as R
    model, ref, batch = _finetune_pair(train_layers, large=True)
    model = model.cuda().eval()
    feats, _, _ = model({k: v.cuda() for k, v in batch.items()})
    loss = model.compute_loss(feats)["loss"]
    loss.backward()
    for p in ref.parameters():
        p.requires_grad_(False)
    lys = ref.encoder.encoder.layers
    for i in train_layers:
        for p in lys[i].parameters():
            p.requires_grad_(True)
    for p in ref.parallel_branch.parameters():
        p.requires_grad_(True)
    ref.ws_weights.requires_grad_(True)
    wavs = [batch["wav"][b, :int(batch["wav_len"][b])] for b in range(4)]
    padded, mask = HR.preprocess_input(wavs, ref.hubert_cfg.normalize)
    with torch.enable_grad():
        hidden = HR.hubert_forward.__wrapped__(ref.encoder, padded, mask)["layer_results"]
        flen = HR.feat_lengths([len(w) for w in wavs], 320, hidden[-1].shape[1])
        pa = R.l2_normalize(ref.parallel_branch(R.weighted_sum(hidden, ref.ws_weights, True), flen))
        with torch.no_grad():
            img = R.l2_normalize(ref.clip.encode_image(batch["image"]))
        ref_loss = R.masked_contrastive_loss(pa, img, batch["id"], ref.inv_temperature)
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) < 2e-2
    mine = dict(model.named_parameters())
    checked, worst = 0, (1.0, "")
    for i in train_layers:
        for k, p in lys[i].named_parameters():
            got = mine[f"audio_encoder.encoder.encoder.layers.{i}.{k}"].grad
            assert got is not None, (i, k)
            if p.grad.norm().item() < 1e-7:
                assert got.norm().item() < 1e-4, (i, k)
                continue
            c, ratio = _cos(got, p.grad), got.norm().item() / p.grad.norm().item()
            worst = min(worst, (c, f"{i}.{k}"))
            assert c > 0.97 and abs(ratio - 1) < 0.12, (i, k, c, ratio)
            checked += 1
    print("pre-LN gradients checked:", checked, "worst cosine:", worst)
    assert checked >= 12 * len(train_layers)
    assert _cos(mine["audio_encoder.weightedsum_layer.weights"].grad, ref.ws_weights.grad) > 0.97
    assert all(p.grad is None for k, p in mine.items() if k.startswith("audio_encoder.encoder.") and not any(f".layers.{i}." in k for i in train_layers))


@pytest.mark.parametrize("large", [False, True])
def test_full_encoder_training_gradients_vs_oracle_autograd(large):
    """`audio_encoder.trainable: true` with no layer lists (speech_encoder_plus.py:399-401): loss.backward() reaches EVERY encoder tensor the forward
    uses -- conv feature extractor (x feature_grad_mult 0.1), feature LayerNorm, post_extract_proj, positional conv (weight-norm g and v), encoder
    LayerNorm, all transformer layers -- and matches the fp32 oracle's autograd on the same weights and batch."""
    from oracle import hubert_ref as HR
    from oracle import speechclip_ref as R
    model, ref, batch = _finetune_pair([], everything=True, large=large)
    model = model.cuda().eval()
    assert model.audio_encoder.train_front
    feats, _, _ = model({k: v.cuda() for k, v in batch.items()})
    loss = model.compute_loss(feats)["loss"]
    loss.backward()
    for p in ref.parameters():
        p.requires_grad_(False)
    for k, p in ref.encoder.named_parameters():
        p.requires_grad_(not k.startswith(("mask_emb", "final_proj", "label_embs_concat") + (("encoder.layer_norm",) if large else ())))
    for p in ref.parallel_branch.parameters():
        p.requires_grad_(True)
    ref.ws_weights.requires_grad_(True)
    ref.encoder.feature_grad_mult = 1.0 if large else 0.1
    wavs = [batch["wav"][b, :int(batch["wav_len"][b])] for b in range(4)]
    padded, mask = HR.preprocess_input(wavs, ref.hubert_cfg.normalize)
    with torch.enable_grad():
        hidden = HR.hubert_forward.__wrapped__(ref.encoder, padded, mask)["layer_results"]
        flen = HR.feat_lengths([len(w) for w in wavs], 320, hidden[-1].shape[1])
        pa = R.l2_normalize(ref.parallel_branch(R.weighted_sum(hidden, ref.ws_weights, large), flen))
        with torch.no_grad():
            img = R.l2_normalize(ref.clip.encode_image(batch["image"]))
        ref_loss = R.masked_contrastive_loss(pa, img, batch["id"], ref.inv_temperature)
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) < 2e-2
    mine = dict(model.named_parameters())
    checked, worst = 0, (1.0, "")
    for k, p in ref.encoder.named_parameters():
        if not p.requires_grad:
            assert mine["audio_encoder.encoder." + k].grad is None, k
            continue
        got = mine["audio_encoder.encoder." + k].grad
        assert got is not None and p.grad is not None, k
        if p.grad.norm().item() < 1e-7:
            assert got.norm().item() < 1e-4, k
            continue
        c, ratio = _cos(got, p.grad), got.norm().item() / p.grad.norm().item()
        worst = min(worst, (c, k))
        assert c > 0.97 and abs(ratio - 1) < 0.12, (k, c, ratio)
        checked += 1
    print("full-encoder gradients checked:", checked, "worst cosine:", worst)
    assert checked >= (35 if large else 18) + 3 * 12
    assert _cos(mine["audio_encoder.weightedsum_layer.weights"].grad, ref.ws_weights.grad) > 0.97


def test_short_full_encoder_run_lowers_the_loss_and_moves_the_conv_stack():
    model, _, batch = _finetune_pair([], everything=True)
    model = model.cuda().train()
    batch = {k: v.cuda() for k, v in batch.items()}
    model.config.audio_encoder.optim.args.lr = 3e-4
    model.config.audio_encoder.scheduler.warmup = 1
    (opt,), (sch,) = model.configure_optimizers()
    conv3 = getattr(model.audio_encoder.encoder.feature_extractor.conv_layers[3], "0").weight
    pos_v = getattr(model.audio_encoder.encoder.encoder.pos_conv, "0").weight_v
    c0, v0 = conv3.detach().clone(), pos_v.detach().clone()
    with torch.no_grad():
        model.eval()
        before = model(batch)[0]["parallel_audio_feat"].clone()
        model.train()
    losses = []
    torch.manual_seed(0)
    for step in range(20):
        opt.zero_grad()
        loss = model.training_step_end(model.training_step(batch, step))["loss"]
        loss.backward()
        opt.step()
        sch["scheduler"].step()
        losses.append(loss.item())
    assert all(np.isfinite(losses)) and np.mean(losses[-4:]) < 0.8 * np.mean(losses[:3]), losses
    assert not torch.equal(c0, conv3.detach()) and not torch.equal(v0, pos_v.detach())
    model.eval()
    with torch.no_grad():
        after = model(batch)[0]["parallel_audio_feat"]
    assert (after - before).abs().max().item() > 1e-3           # the eval engine repacked the trained front-end weights (parameter epoch)


def test_short_finetuning_run_moves_the_encoder_layer_and_the_eval_path_sees_it():
    model, _, batch = _finetune_pair([2], reinit=True)
    model = model.cuda().train()
    batch = {k: v.cuda() for k, v in batch.items()}
    model.config.audio_encoder.optim.args.lr = 1e-3
    model.config.audio_encoder.scheduler.warmup = 1
    (opt,), (sch,) = model.configure_optimizers()
    w0 = model.audio_encoder.encoder.encoder.layers[2].fc1.weight.detach().clone()
    frozen0 = model.audio_encoder.encoder.encoder.layers[0].fc1.weight.detach().clone()
    with torch.no_grad():
        model.eval()
        before = model(batch)[0]["parallel_audio_feat"].clone()
        model.train()
    losses = []
    torch.manual_seed(0)
    for step in range(25):
        opt.zero_grad()
        loss = model.training_step_end(model.training_step(batch, step))["loss"]
        loss.backward()
        opt.step()
        sch["scheduler"].step()
        losses.append(loss.item())
    assert all(np.isfinite(losses)) and np.mean(losses[-5:]) < 0.7 * np.mean(losses[:3]), losses
    assert not torch.equal(w0, model.audio_encoder.encoder.encoder.layers[2].fc1.weight) and torch.equal(frozen0, model.audio_encoder.encoder.encoder.layers[0].fc1.weight)
    model.eval()
    with torch.no_grad():
        after = model(batch)[0]["parallel_audio_feat"]
    assert (after - before).abs().max().item() > 1e-3           # the eval path repacked the trained layer's weights (parameter epoch)


def test_eval_path_repacks_after_a_plain_torch_optimizer_step():
    """configure_optimizers falls back to a torch optimizer for optim.name != "Adam"; that bumps the tensors' `_version`, not
    ops.param_epoch.  The engine's packed bf16 operands must follow either: after one AdamW step on a fine-tuned layer the eval forward
    equals the oracle run on the UPDATED weights (and differs from the pre-step forward)."""
    model, ref, batch = _finetune_pair([2])
    model = model.cuda().train()
    batch = {k: v.cuda() for k, v in batch.items()}
    with torch.no_grad():
        model.eval()
        before = model(batch)[0]["parallel_audio_feat"].float().cpu().clone()       # packs the operands
        model.train()
    enc_params = [p for p in model.audio_encoder.encoder.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(enc_params, lr=2e-2)
    loss = model.training_step_end(model.training_step(batch, 0))["loss"]
    loss.backward()
    opt.step()
    model.eval()
    with torch.no_grad():
        after = model(batch)[0]["parallel_audio_feat"].float().cpu()
    assert (after - before).abs().max().item() > 1e-3, "the eval path still runs on the pre-step packed weights"
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    ref.encoder.load_state_dict({k[len("audio_encoder.encoder."):]: v for k, v in sd.items() if k.startswith("audio_encoder.encoder.")})
    want = ref.eval()({k: v.cpu() for k, v in batch.items()})["parallel_audio_feat"]
    from helpers import assert_rows_match
    assert_rows_match(after, want, 0.99, "eval forward after a torch optimizer step")
    stale = torch.nn.functional.cosine_similarity(before - want.mean(0, keepdim=True), want - want.mean(0, keepdim=True), dim=-1)
    print("centred cosine of the STALE forward vs the updated oracle:", stale.tolist())


def test_cascaded_branch_passes_the_frame_gradient_down_too(tmp_path):
    """C-base layout with a fine-tuned encoder layer: the cascaded head's backward (K keyword queries, BatchNorm with batch statistics, straight-through
    VQ, frozen text tower) also returns d loss / d frames (sc_cls_pool_dz), the encoder layer receives finite, non-zero gradients, and a few
    FusedAdam steps on one batch lower the loss."""
    from helpers import make_config
    from oracle.clip_ref import ClipRefConfig
    from oracle.hubert_ref import HubertRefConfig
    from speechclip_amd.model import KWClip_GeneralTransformer
    from speechclip_amd.module.clip_model import ClipConfig
    from speechclip_amd.module.hubert import HubertConfig
    vocab = np.array([0, 320, 510, 511] + list(range(5, 300, 3)))
    vp = str(tmp_path / "vocab.npy")
    np.save(vp, np.stack([vocab, np.arange(len(vocab))[::-1] + 1], axis=1))
    href, cref = HubertRefConfig.tiny(), ClipRefConfig.tiny()
    cfg = make_config(d_model=128, branch_heads=4, parallel=False, cascaded=True, reduce_vocab=vp, hubert_config=HubertConfig(**dataclasses.asdict(href)),
                      clip_config=ClipConfig(**dataclasses.asdict(cref)))
    cfg.audio_encoder.trainable = True
    cfg.audio_encoder.unfreeze_layers = [1]
    torch.manual_seed(3)
    model = KWClip_GeneralTransformer(cfg).cuda().train()
    g = _g(4)
    lens = [8000, 6100, 8000, 4000]
    wav = torch.zeros(4, 8000)
    for i, l in enumerate(lens):
        wav[i, :l] = 0.3 * torch.randn(l, generator=g)
    batch = {"wav": wav.cuda(), "wav_len": torch.tensor(lens).cuda(), "image": torch.randn(4, 3, 64, 64, generator=g).cuda(), "id": torch.arange(4).cuda()}
    model.config.audio_encoder.optim.args.lr = 1e-3
    model.config.audio_encoder.scheduler.warmup = 1
    (opt,), (sch,) = model.configure_optimizers()
    losses = []
    torch.manual_seed(0)
    for step in range(20):
        opt.zero_grad()
        loss = model.training_step_end(model.training_step(batch, step))["loss"]
        loss.backward()
        if step == 0:
            lyr = model.audio_encoder.encoder.encoder.layers[1]
            for k, p in lyr.named_parameters():
                assert p.grad is not None and torch.isfinite(p.grad).all(), k
            assert lyr.fc1.weight.grad.abs().max().item() > 0 and lyr.self_attn.q_proj.weight.grad.abs().max().item() > 0
            assert model.audio_encoder.encoder.encoder.layers[0].fc1.weight.grad is None
        opt.step()
        sch["scheduler"].step()
        losses.append(loss.item())
    assert all(np.isfinite(losses)) and np.mean(losses[-4:]) < np.mean(losses[:3]), losses


@pytest.mark.parametrize("everything", [False, True])
def test_single_utterance_batch_goes_through_the_training_path(everything):
    """B = 1: the pooling head's frames view has no batch stride to read the padded frame count from (it uses T), while the hidden states keep
    their padded rows -- the layer-mix gradient has to reconcile the two (a reshape used to fail here).  One pair has zero InfoNCE loss; the
    step must still run end to end and leave finite (zero) gradients."""
    model, _, batch = _finetune_pair([] if everything else [2], everything=everything)
    model = model.cuda().train()
    one = {k: v[:1].cuda() for k, v in batch.items()}
    one["wav"] = one["wav"][:, :6777]
    one["wav_len"] = torch.tensor([6777]).cuda()
    loss = model.training_step_end(model.training_step(one, 0))["loss"]
    loss.backward()
    assert abs(loss.item()) < 1e-5
    g = model.audio_encoder.weightedsum_layer.weights.grad
    assert g is not None and torch.isfinite(g).all()
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)


@pytest.mark.parametrize("drop", [None, (0.2, 777)])
def test_fused_attention_probs_kernel_vs_the_unfused_form(drop):
    """sc_attn_bwd_probs (S, dP on MFMA fragments from the packed rows + softmax backward in registers) against the first, unfused form: the two
    products as fp32 images (computed here by torch on the same bf16 operands) through sc_attn_softmax_bwd_heads -- same P, same dS, same mask."""
    from speechclip_amd import ops
    B, T, H = 3, 70, 2
    d, Lp = H * 64, 128
    g = _g(17)
    qkv = (0.7 * torch.randn(B * T, 3 * d, generator=g)).to(BF)
    att = torch.randn(B * T, d, generator=g).to(BF)
    datt = torch.randn(B * T, d, generator=g).to(BF)
    klens = torch.tensor([70, 33, 7], dtype=torch.int32)
    x = qkv.float().view(B, T, 3, H, 64)
    S = torch.zeros(B * H, Lp, Lp)
    dP = torch.zeros(B * H, Lp, Lp)
    S[:, :T, :T] = torch.einsum("bihd,bjhd->bhij", x[:, :, 0], x[:, :, 1]).reshape(B * H, T, T)
    dP[:, :T, :T] = torch.einsum("bihd,bjhd->bhij", datt.float().view(B, T, H, 64), x[:, :, 2]).reshape(B * H, T, T)
    P0, dS0 = ops.attn_softmax_bwd_heads(S.cuda(), dP.cuda(), datt.cuda(), att.cuda(), T, klens.cuda(), T, 0.125, B, H, drop)
    P1, dS1 = ops.attn_bwd_probs(qkv.cuda(), datt.cuda(), att.cuda(), klens.cuda(), B, T, H, drop)
    assert P1.shape == (B * H, Lp, Lp)
    torch.testing.assert_close(P1.float(), P0.float(), atol=4e-3, rtol=2e-2)
    torch.testing.assert_close(dS1.float(), dS0.float(), atol=2e-2, rtol=3e-2)
    assert torch.equal(P1 == 0, P0 == 0) or ((P1 == 0) != (P0 == 0)).float().mean().item() < 1e-4      # same zeros: mask, key padding, row padding
    for z in range(B * H):
        n = int(klens[z // H])
        assert P1[z, T:].abs().max().item() == 0 and P1[z, :, n:].abs().max().item() == 0 and dS1[z, :, n:].abs().max().item() == 0

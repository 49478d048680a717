"""Training the HuBERT front end (bare `audio_encoder.trainable: true`, speech_encoder_plus.py:399-401): conv feature extractor, feature
LayerNorm, post_extract_proj, positional conv and encoder LayerNorm backward on the HIP kernels (speechclip_amd/train_front.py) against torch
autograd / the oracle's autograd on the same weights."""
import dataclasses

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _cos(a, b):
    return F.cosine_similarity(a.double().reshape(1, -1).cpu(), b.double().reshape(1, -1).cpu()).item()


def _close(name, mine, ref, cos=0.995, ratio=0.05):
    mine, ref = mine.float().cpu(), ref.float().cpu()
    assert mine.shape == ref.shape, (name, mine.shape, ref.shape)
    c = _cos(mine, ref)
    r = (mine.norm() / ref.norm()).item()
    assert c > cos and abs(r - 1) < ratio, (name, c, r)
    return c


@pytest.mark.parametrize("B,C,L", [(3, 64, 4000), (2, 512, 16000)])
def test_conv0_backward_vs_autograd(B, C, L):
    from speechclip_amd import ops
    g = torch.Generator().manual_seed(C + L)
    wav = 0.3 * torch.randn(B, L, generator=g)
    w = (0.3 * torch.randn(C, 1, 10, generator=g)).requires_grad_(True)
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).requires_grad_(True)
    beta = (0.1 * torch.randn(C, generator=g)).requires_grad_(True)
    T0 = (L - 10) // 5 + 1
    P = -(-T0 // 64) * 64
    y = F.gelu(F.group_norm(F.conv1d(wav[:, None], w, stride=5), C, gamma, beta, 1e-5))      # [B, C, T0]
    dy = torch.randn(B, T0, C, generator=g).to(BF)
    y.backward(dy.float().permute(0, 2, 1))
    dyp = torch.zeros(B, P, C, dtype=BF)
    dyp[:, :T0] = dy
    dw, dg, db = ops.conv0_bwd(wav.cuda(), w.detach().reshape(C, 10).cuda().contiguous(), gamma.detach().cuda(), beta.detach().cuda(),
                               dyp.view(B * P, C).cuda(), T0, P)
    _close("dw", dw, w.grad.view(C, 10), 0.999, 0.02)
    _close("dgamma", dg, gamma.grad, 0.999, 0.02)
    _close("dbeta", db, beta.grad, 0.999, 0.02)


def test_reverse_rows_and_posconv_adjoint():
    """<posconv(x; W), y> == <x, dgrad(y)> for random x, y: the input gradient of the grouped conv as the build computes it (time-reversed conv with
    swapped channel roles) is the adjoint of the forward conv -- checked against torch's conv1d autograd too."""
    from speechclip_amd import ops
    from speechclip_amd.train_front import _pos_operands
    B, Tp, D, G, Kw = 3, 40, 128, 4, 16
    cg = D // G
    g = torch.Generator().manual_seed(2)
    w = 0.2 * torch.randn(D, cg, Kw, generator=g)
    x = torch.randn(B, Tp, D, generator=g).to(BF)
    dy = torch.randn(B, Tp, D, generator=g).to(BF)
    xf = x.float().requires_grad_(True)
    wf = w.to(BF).float()
    y = F.conv1d(xf.permute(0, 2, 1), wf, padding=Kw // 2, groups=G)[:, :, :-1].permute(0, 2, 1)
    (y * dy.float()).sum().backward()
    fwd, adj = _pos_operands(w.cuda(), G, Kw)
    full = torch.full((B,), Tp, dtype=torch.int32).cuda()
    mine_y = ops.posconv_conv(x.cuda().view(B * Tp, D), full, fwd, B, Tp, D, G, Kw).view(B, G, Tp, cg).permute(0, 2, 1, 3).reshape(B, Tp, D)
    assert _cos(mine_y, y.detach()) > 0.9995
    rev = ops.reverse_rows_bf16(dy.cuda().view(B * Tp, D), B, Tp, D)
    assert torch.equal(rev.view(B, Tp, D).cpu(), dy.flip(1))
    convT = ops.posconv_conv(rev, full, adj, B, Tp, D, G, Kw)
    dx = ops.posconv_dgrad_finish(convT, torch.zeros(B * Tp, D, dtype=BF).cuda(), full, B, Tp, D, G).view(B, Tp, D)
    _close("dx", dx, xf.grad, 0.999, 0.02)


def _front_pair(lens, L):
    from oracle.hubert_ref import HubertModelRef, HubertRefConfig, randomize_norm_affine
    from speechclip_amd.module.hubert import HubertConfig, HubertModel
    href = HubertRefConfig.tiny()
    torch.manual_seed(11)
    ref = HubertModelRef(href)
    randomize_norm_affine(ref, torch.Generator().manual_seed(5))
    ref.feature_grad_mult = 0.1
    enc = HubertModel(HubertConfig(**dataclasses.asdict(href)))
    enc.load_state_dict(ref.state_dict())
    g = torch.Generator().manual_seed(3)
    wav = torch.zeros(len(lens), L)
    for i, l in enumerate(lens):
        wav[i, :l] = 0.3 * torch.randn(l, generator=g)
    return href, ref, enc.cuda(), wav


def test_front_forward_and_backward_vs_oracle_autograd():
    from speechclip_amd import ops
    from speechclip_amd.train_front import HubertFrontTrainFn, front_params
    lens, L = [8000, 5200, 8000, 3100], 8000
    href, ref, enc, wav = _front_pair(lens, L)
    B = len(lens)
    T0, T, P0, Tp = enc.frame_geometry(L)
    valid = enc.valid_frames(lens, L, T)
    cfg = enc.cfg
    meta = dict(conv_layers=[tuple(c) for c in cfg.conv_layers], T0=T0, P0=P0, Tp=Tp, d=cfg.encoder_embed_dim, G=cfg.conv_pos_groups, Kw=cfg.conv_pos,
                grad_mult=0.1)
    prm = front_params(enc)
    for p in prm:
        p.requires_grad_(True)
    h0 = HubertFrontTrainFn.apply(meta, wav.cuda(), ops.dev_ints(valid, torch.int32, torch.device("cuda")), *prm)
    # the oracle's own path to the same tensor, with autograd (customFunc_hubert_forward up to layer_results[0], speech_encoder_plus.py:75-101, :29-47)
    ref.train(False)
    pad = torch.arange(L)[None, :] >= torch.tensor(lens)[:, None]
    feats = ref.forward_features(wav).transpose(1, 2)
    feats = ref.layer_norm(feats)
    pm = ref.forward_padding_mask(feats, pad)
    x = ref.post_extract_proj(feats)
    x = x.masked_fill(pm[:, :, None], 0.0)
    x = x + ref.encoder.pos_conv(x.transpose(1, 2)).transpose(1, 2)
    h0_ref = ref.encoder.layer_norm(x)                                     # [B, T, d]
    assert h0_ref.shape[1] == T and [int((~pm[b]).sum()) for b in range(B)] == valid
    g = torch.Generator().manual_seed(8)
    dh = torch.randn(B, T, cfg.encoder_embed_dim, generator=g)
    dh = dh * (~pm)[:, :, None]                                           # no gradient comes back from padded frames (attention never reads them)
    for b in range(B):
        n = valid[b]
        assert _cos(h0.view(B, Tp, -1)[b, :n], h0_ref[b, :n].detach()) > 0.999
    dhp = torch.zeros(B, Tp, cfg.encoder_embed_dim)
    dhp[:, :T] = dh
    h0.backward(dhp.view(B * Tp, -1).to(BF).cuda())
    h0_ref.backward(dh.to(BF).float())
    rp = dict(ref.named_parameters())
    names = (["feature_extractor.conv_layers.0.0.weight", "feature_extractor.conv_layers.0.2.weight", "feature_extractor.conv_layers.0.2.bias"] +
             [f"feature_extractor.conv_layers.{i}.0.weight" for i in range(1, 7)] +
             ["layer_norm.weight", "layer_norm.bias", "post_extract_proj.weight", "post_extract_proj.bias", "encoder.pos_conv.0.weight_g",
              "encoder.pos_conv.0.weight_v", "encoder.pos_conv.0.bias", "encoder.layer_norm.weight", "encoder.layer_norm.bias"])
    report = {}
    for name, p in zip(names, prm):
        assert p.grad is not None, name
        report[name] = round(_close(name, p.grad, rp[name].grad, 0.98, 0.1), 4)
    print(report)

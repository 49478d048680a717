"""Train-mode dropouts of the FROZEN encoder (Lightning's model.train() re-enables them in the reference: speech_encoder_plus.py:42, :87, the
fairseq layers' dropout modules): counter-based masks on HIP kernels -- sc_dropout_bf16, sc_attention_fwd_dropout -- and their wiring."""
import dataclasses

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def test_dropout_kernel_statistics_residual_and_determinism():
    from speechclip_amd import ops
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(4096, 768, generator=g) + 3.0).to(BF).cuda()          # no exact zeros in the input
    res = torch.randn(4096, 768, generator=g).to(BF).cuda()
    for p in (0.1, 0.5):
        y = ops.dropout_bf16(x, p, 1234)
        kept = y != 0
        frac = 1 - kept.float().mean().item()
        assert abs(frac - p) < 3e-3, (p, frac)
        assert torch.allclose(y[kept].float(), (x[kept].float() / (1 - p)).to(BF).float(), rtol=1e-2)
        assert abs(y.float().mean().item() / x.float().mean().item() - 1) < 5e-3                    # unbiased
        assert torch.equal(y, ops.dropout_bf16(x, p, 1234)) and not torch.equal(y, ops.dropout_bf16(x, p, 1235))
        z = ops.dropout_bf16(x, p, 1234, residual=res)
        want = (x.float() / (1 - p) * kept + res.float()).to(BF).float()                           # one rounding, as the kernel does
        assert torch.allclose(z.float(), want, rtol=8e-3, atol=1e-3)
        # masks of rows are not correlated with the row index (a different row has a different pattern)
        assert not torch.equal(kept[0], kept[1])
    assert torch.equal(ops.dropout_bf16(x, 0.0, 7), x)
    inplace = x.clone()
    ops.dropout_bf16(inplace, 0.3, 9, out=inplace)
    assert torch.equal(inplace, ops.dropout_bf16(x, 0.3, 9))


@pytest.mark.parametrize("T,lens", [(64, [64, 40]), (300, [300, 123])])
def test_attention_probability_dropout(T, lens):
    """With V = identity blocks the attention output IS the (dropped) probability row: zeros at rate p, survivors = P / (1 - p), the softmax
    normalisation untouched; p = 0 equals the plain kernel bit for bit; the mean over seeds approaches the plain output."""
    from speechclip_amd import ops
    B, H, D = len(lens), 2, 128
    g = torch.Generator().manual_seed(T)
    qkv = (0.5 * torch.randn(B * T, 3 * D, generator=g)).to(BF)
    kl = torch.tensor(lens, dtype=torch.int32).cuda()
    base = ops.attention(qkv.cuda(), B, T, H, kl)
    assert torch.equal(ops.attention_dropout(qkv.cuda(), B, T, H, kl, 0.0, 5), base)
    acc = torch.zeros_like(base, dtype=torch.float32)
    n = 48
    for s in range(n):
        acc += ops.attention_dropout(qkv.cuda(), B, T, H, kl, 0.1, 100 + s).float()
    valid = torch.cat([torch.arange(T) < l for l in lens]).cuda()
    err = ((acc / n - base.float())[valid]).abs().mean().item() / base.float()[valid].abs().mean().item()
    assert err < 0.12, err                                                  # unbiased: the seed average approaches the plain output
    if T == 64:
        v = torch.zeros(B, T, H, 64)
        for j in range(64):
            v[:, j, :, j] = 1.0                                             # V_h = I: out[i, h, :] = dropped P[i, h, :]
        q2 = qkv.clone().view(B * T, 3, H, 64)
        q2[:, 2] = v.view(B * T, H, 64).to(BF)
        q2 = q2.view(B * T, 3 * D).contiguous().cuda()
        p0 = ops.attention(q2, B, T, H, kl).float().view(B, T, H, 64)       # plain probabilities
        pd = ops.attention_dropout(q2, B, T, H, kl, 0.25, 77).float().view(B, T, H, 64)
        for b, l in enumerate(lens):
            P0, Pd = p0[b, :l, :, :l], pd[b, :l, :, :l]
            dropped = (Pd == 0) & (P0 > 1e-3)
            frac = dropped.float().sum().item() / (P0 > 1e-3).float().sum().item()
            assert abs(frac - 0.25) < 0.03, frac
            keep = Pd != 0
            assert torch.allclose(Pd[keep], P0[keep] / 0.75, rtol=3e-2, atol=2e-3)
            assert (p0[b, :l, :, l:].abs().max().item() if l < T else 0.0) == 0.0


def test_frozen_encoder_applies_its_dropouts_in_train_mode_only(monkeypatch):
    from oracle.hubert_ref import HubertRefConfig
    from speechclip_amd.module import FairseqSpeechEncoder_Hubert
    from speechclip_amd.module.hubert import HubertConfig
    monkeypatch.setenv("SC_FROZEN_DROPOUT", "1")
    hc = HubertConfig(**dataclasses.asdict(dataclasses.replace(HubertRefConfig.tiny(), encoder_layers=3)))
    assert (hc.dropout, hc.attention_dropout, hc.activation_dropout, hc.dropout_input) == (0.1, 0.1, 0.0, 0.1)
    torch.manual_seed(0)
    enc = FairseqSpeechEncoder_Hubert("hubert", feat_select_idx="hidden_states", max_audio_len=100000, hubert_config=hc).cuda()
    g = torch.Generator().manual_seed(2)
    wav = (0.3 * torch.randn(3, 8000, generator=g)).cuda()
    lens = torch.tensor([8000, 6000, 3000])
    def run():                          # the hidden states are views of the engine's reused workspace: copy them out
        hs, fl_ = enc(wav, lens)
        return [h.clone() for h in hs], fl_
    enc.eval()
    with torch.no_grad():
        e1, fl = run()
        e2, _ = run()
    assert all(torch.equal(a, b) for a, b in zip(e1, e2))                   # eval: deterministic, no dropout
    enc.train()
    with torch.no_grad():
        torch.manual_seed(5)
        t1, _ = run()
        torch.manual_seed(5)
        t2, _ = run()
        torch.manual_seed(6)
        t3, _ = run()
    assert all(torch.equal(a, b) for a, b in zip(t1, t2)) and not torch.equal(t1[-1], t3[-1])          # seeded by torch's generator
    n = int(fl[0])
    z = (t1[0][0, :n] == 0).float().mean().item()
    assert abs(z - 0.1) < 0.02, z                                           # hidden state 0 IS the dropped tensor (layer_results[0], :47)
    cos = torch.nn.functional.cosine_similarity(t1[-1][0, :n].float().reshape(1, -1), e1[-1][0, :n].float().reshape(1, -1)).item()
    assert 0.5 < cos < 0.999, cos                                           # perturbed, not unrelated
    monkeypatch.setenv("SC_FROZEN_DROPOUT", "0")
    with torch.no_grad():
        off, _ = run()
    assert all(torch.equal(a, b) for a, b in zip(off, e1))                  # the switch restores the eval arithmetic in train mode
    large = HubertConfig.from_name("hubert_large_ll60k")
    assert (large.dropout, large.attention_dropout, large.dropout_input) == (0.0, 0.0, 0.0)


def _hash32(x):
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7feb352d)) & np.uint64(0xffffffff)
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846ca68b)) & np.uint64(0xffffffff)
    x ^= x >> np.uint64(16)
    return x


def _keep(seed, idx, p):
    """csrc/common.h keep_elem, restated: element idx survives iff hash32(seed ^ hash32(idx + 0x9e3779b9)) >= p * 2^32."""
    idx = (np.asarray(idx, dtype=np.uint64) + np.uint64(0x9e3779b9)) & np.uint64(0xffffffff)
    h = _hash32(np.uint64(seed & 0xffffffff) ^ _hash32(idx))
    return torch.from_numpy((h >= np.uint64(int(p * 4294967296.0))).astype(np.float32))


def _keep_attn(seed, B, H, T, p):
    """The attention-probability mask (csrc/common.h hash_pair): one hash per pair of adjacent keys of a query row, 16 bits each;
    pair index = ((b*H + h)*T + query) * ceil(T/2) + (key >> 1); keep iff the half's 16 bits >= p * 2^16."""
    rows = np.arange(B * H * T, dtype=np.uint64)[:, None]
    keys = np.arange(T, dtype=np.uint64)[None, :]
    pair = (rows * np.uint64((T + 1) // 2) + (keys >> np.uint64(1))) & np.uint64(0xffffffff)
    h = _hash32(((pair * np.uint64(0x9E3779B1)) + np.uint64(seed & 0xffffffff)) & np.uint64(0xffffffff))
    bits = np.where((keys & np.uint64(1)) == 1, h >> np.uint64(16), h & np.uint64(0xffff))
    return torch.from_numpy((bits >= np.uint64(int(p * 65536.0))).astype(np.float32)).view(B, H, T, T)


def test_trained_layer_with_dropout_matches_autograd_with_the_same_masks():
    """One post-LN layer through train_hubert.HubertLayersTrainFn with meta["drop"]: forward and every gradient against an fp32 torch
    re-statement of the fairseq layer (dropout1 / dropout3 / attention dropout) that uses the SAME counter-based masks, regenerated in numpy."""
    from speechclip_amd import ops
    from speechclip_amd.train_hubert import HubertLayersTrainFn
    B, Tp, H, d, ffn = 2, 64, 2, 128, 256
    M = B * Tp
    lens = [64, 40]
    g = torch.Generator().manual_seed(4)
    P = [0.08 * torch.randn(d, d, generator=g), 0.02 * torch.randn(d, generator=g)] * 0
    shapes = [(d, d), (d,), (d, d), (d,), (d, d), (d,), (d, d), (d,), (d,), (d,), (ffn, d), (ffn,), (d, ffn), (d,), (d,), (d,)]
    params = []
    for i, sh in enumerate(shapes):
        t = 0.08 * torch.randn(*sh, generator=g) if len(sh) == 2 else 0.05 * torch.randn(*sh, generator=g)
        if i in (8, 14):
            t = 1.0 + t                                               # LayerNorm gains
        params.append(t)
    h_in = torch.randn(M, d, generator=g).to(BF)
    dout = torch.randn(1, M, d, generator=g).to(BF)
    for b, n in enumerate(lens):
        dout[0, b * Tp + n:(b + 1) * Tp] = 0                         # nothing flows back from padded frames
    p_h, p_a, seed = 0.1, 0.1, 987654
    s0, seeds = seed & 0x7fffffff, []
    for _ in range(4):
        s0 = (s0 * 1103515245 + 12345) & 0x7fffffff
        seeds.append(s0)
    sa, s1, _, s3 = seeds
    # ---- the HIP node
    dev_p = [p.clone().cuda().requires_grad_(True) for p in params]
    hin_d = h_in.clone().cuda().requires_grad_(True)
    meta = dict(B=B, Tp=Tp, H=H, eps=1e-5, train=[True], drop=dict(hidden=p_h, attention=p_a, activation=0.0, seed=seed))
    out = HubertLayersTrainFn.apply(meta, hin_d, torch.tensor(lens, dtype=torch.int32).cuda(), *dev_p)
    out.backward(dout.cuda())
    # ---- fp32 torch re-statement with the same masks
    rp = [p.clone().requires_grad_(True) for p in params]
    qw, qb, kw, kb, vw, vb, ow, ob, g1, b1n, w1, b1, w2, b2, g2, b2n = rp
    x = h_in.float().clone().requires_grad_(True)
    q = (x @ qw.t() + qb).view(B, Tp, H, 64).transpose(1, 2)
    k = (x @ kw.t() + kb).view(B, Tp, H, 64).transpose(1, 2)
    v = (x @ vw.t() + vb).view(B, Tp, H, 64).transpose(1, 2)
    s = (q * 0.125) @ k.transpose(-1, -2)
    kmask = torch.arange(Tp)[None, :] >= torch.tensor(lens)[:, None]
    pr = torch.softmax(s.masked_fill(kmask[:, None, None, :], float("-inf")), dim=-1)
    am = _keep_attn(sa, B, H, Tp, p_a) / (1 - p_a)
    att = ((pr * am) @ v).transpose(1, 2).reshape(M, d)
    m1 = _keep(s1, np.arange(M * d), p_h).view(M, d) / (1 - p_h)
    m3 = _keep(s3, np.arange(M * d), p_h).view(M, d) / (1 - p_h)
    y1 = (att @ ow.t() + ob) * m1 + x
    x1 = torch.nn.functional.layer_norm(y1, (d,), g1, b1n, 1e-5)
    hm = torch.nn.functional.gelu(x1 @ w1.t() + b1)
    y2 = (hm @ w2.t() + b2) * m3 + x1
    ref = torch.nn.functional.layer_norm(y2, (d,), g2, b2n, 1e-5)
    ref.backward(dout[0].float())
    cosf = torch.nn.functional.cosine_similarity
    for b, n in enumerate(lens):
        rows = slice(b * Tp, b * Tp + n)
        assert cosf(out[0, rows].float().cpu().reshape(1, -1), ref[rows].detach().reshape(1, -1)).item() > 0.999
        assert cosf(hin_d.grad[rows].float().cpu().reshape(1, -1), x.grad[rows].reshape(1, -1)).item() > 0.995
    names = "q_w q_b k_w k_b v_w v_b o_w o_b ln1_w ln1_b fc1_w fc1_b fc2_w fc2_b ln2_w ln2_b".split()
    for nme, mine, r in zip(names, dev_p, rp):
        if r.grad.norm().item() < 1e-5 * max(1.0, rp[0].grad.norm().item()):      # k_b: exactly zero in theory (softmax rows are shift invariant)
            assert mine.grad.float().norm().item() < 1e-2 * dev_p[0].grad.float().norm().item(), nme
            continue
        c = cosf(mine.grad.float().cpu().reshape(1, -1), r.grad.reshape(1, -1)).item()
        ratio = mine.grad.float().norm().item() / r.grad.norm().item()
        assert c > 0.99 and abs(ratio - 1) < 0.06, (nme, c, ratio)
    # and the masks really were applied: without them the output differs
    meta0 = dict(B=B, Tp=Tp, H=H, eps=1e-5, train=[False])
    with torch.no_grad():
        plain = HubertLayersTrainFn.apply(meta0, h_in.cuda(), torch.tensor(lens, dtype=torch.int32).cuda(), *[p.cuda() for p in params])
    assert cosf(plain[0, :64].float().reshape(1, -1), out[0, :64].detach().float().reshape(1, -1)).item() < 0.999


def test_full_encoder_training_step_with_dropout_runs_and_is_seeded(monkeypatch):
    from test_finetune_gpu import _finetune_pair
    monkeypatch.setenv("SC_FROZEN_DROPOUT", "1")
    model, _, batch = _finetune_pair([], everything=True)
    model = model.cuda().train()
    batch = {k: v.cuda() for k, v in batch.items()}

    def grads(seed):
        model.zero_grad()
        torch.manual_seed(seed)
        np.random.seed(0)
        loss = model.training_step_end(model.training_step(batch, 0))["loss"]
        loss.backward()
        c = getattr(model.audio_encoder.encoder.feature_extractor.conv_layers[2], "0").weight.grad.clone()
        return loss.item(), c
    l1, g1 = grads(3)
    l2, g2 = grads(3)
    l3, g3 = grads(4)
    assert l1 == l2 and torch.equal(g1, g2)                           # same torch seed -> same masks -> same step
    assert l1 != l3 and not torch.equal(g1, g3)
    assert np.isfinite(l1) and torch.isfinite(g1).all() and g1.abs().max().item() > 0


def test_fused_dropout_residual_layernorm_matches_the_two_step_form():
    """sc_dropout_add_layernorm_bf16 (D = 768): LayerNorm(residual + dropout(x)) in one pass, same mask as sc_dropout_bf16 (re-stated in numpy)."""
    from speechclip_amd import ops
    g = torch.Generator().manual_seed(12)
    rows, D, p, seed = 517, 768, 0.1, 4242
    x = torch.randn(rows, D, generator=g).to(BF)
    res = torch.randn(rows, D, generator=g).to(BF)
    gamma, beta = 1 + 0.2 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
    out = ops.dropout_add_layernorm(x.clone().cuda(), res.cuda(), gamma.cuda(), beta.cuda(), p, seed).float().cpu()
    m = _keep(seed, np.arange(rows * D), p).view(rows, D) / (1 - p)
    ref = torch.nn.functional.layer_norm(res.float() + x.float() * m, (D,), gamma, beta, 1e-5)
    assert (out - ref).abs().max().item() < 3e-2 and torch.nn.functional.cosine_similarity(out.reshape(1, -1), ref.reshape(1, -1)).item() > 0.9999
    two = ops.layernorm(ops.dropout_bf16(x.cuda(), p, seed, residual=res.cuda()), gamma.cuda(), beta.cuda()).float().cpu()
    assert (out - two).abs().max().item() < 6e-2                                  # the two-step form rounds the sum to bf16 first
    # other widths fall back to the two-step form with the same result as calling it by hand
    x2, r2 = x[:, :128].contiguous(), res[:, :128].contiguous()
    fb = ops.dropout_add_layernorm(x2.clone().cuda(), r2.cuda(), gamma[:128].cuda().contiguous(), beta[:128].cuda().contiguous(), p, seed)
    hand = ops.layernorm(ops.dropout_bf16(x2.cuda(), p, seed, residual=r2.cuda()), gamma[:128].cuda().contiguous(), beta[:128].cuda().contiguous())
    assert torch.equal(fb, hand)

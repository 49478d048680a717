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

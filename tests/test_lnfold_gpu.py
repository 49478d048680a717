"""LayerNorm folded into the GEMMs around it (sc_gemm_bf16_ln / sc_ln_stats_finalize / sc_weighted_sum_ln_fwd): the eval path of the post-LN
HuBERT layers writes no LayerNorm output at all.  Each piece against fp32 torch on the same bf16 operands, then the
whole encoder against the unfolded path and (through tests/test_e2e_gpu.py, which takes this path at base dims) against the oracle."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _g(seed):
    return torch.Generator().manual_seed(seed)


def _fold(w, b, gamma, beta):
    wp = (w * gamma[None, :]).to(BF)
    return wp, wp.float().sum(1), w @ beta + b


@pytest.mark.parametrize("M,N,K,act", [(300, 2304, 768, 0), (1000, 3072, 768, 1), (70000, 768, 768, 0), (256, 256, 64, 1)])
def test_gemm_with_folded_input_layernorm(M, N, K, act):
    """mode 1: act(LN(y) W^T + b) computed as act(rstd (y W'^T - mean c) + (W beta + b)) from the PRE-norm rows y."""
    from speechclip_amd import ops
    g = _g(M + N + K)
    y = (torch.randn(M, K, generator=g) * (0.5 + 2 * torch.rand(M, 1, generator=g)) + torch.randn(M, 1, generator=g)).to(BF)   # rows of different scale / offset
    w = torch.randn(N, K, generator=g) * K ** -0.5
    b = 0.1 * torch.randn(N, generator=g)
    gamma, beta = 1 + 0.3 * torch.randn(K, generator=g), 0.3 * torch.randn(K, generator=g)
    yf = y.float()
    mean, var = yf.mean(1), yf.var(1, unbiased=False)
    stats = torch.stack([mean, torch.rsqrt(var + 1e-5)], 1).contiguous()
    wp, c, d = _fold(w, b, gamma, beta)
    out = ops.gemm_ln(y.cuda(), wp.cuda(), d.cuda(), 1, act, ln_stats=stats.cuda(), ln_c=c.cuda())
    assert out is not None and out.shape == (M, N) and out.dtype == BF
    ref = F.layer_norm(yf, (K,), gamma, beta, 1e-5) @ w.t() + b
    if act == 1:
        ref = F.gelu(ref)
    torch.testing.assert_close(out.float().cpu(), ref, atol=4e-2, rtol=3e-2)
    # and it agrees with the unfused sequence (layernorm kernel -> bf16 -> gemm) to bf16 round-off
    x16 = ops.layernorm(y.cuda(), gamma.cuda(), beta.cuda())
    unf = ops.gemm(x16, w.to(BF).cuda(), b.cuda(), act)
    assert (out.float() - unf.float()).abs().max().item() < 6e-2


@pytest.mark.parametrize("M,N,K", [(300, 768, 768), (70001, 768, 3072), (512, 256, 64)])
def test_gemm_with_rebuilt_residual_layernorm_and_output_stats(M, N, K):
    """mode 2: C = A W^T + b + bf16(LN(resid)) with the residual's LayerNorm rebuilt from its row statistics, plus the partial statistics of C."""
    from speechclip_amd import ops
    g = _g(M + N + K + 1)
    a = (0.5 * torch.randn(M, K, generator=g)).to(BF)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(BF)
    b = 0.1 * torch.randn(N, generator=g)
    resid = (torch.randn(M, N, generator=g) * (0.5 + torch.rand(M, 1, generator=g)) + 0.5 * torch.randn(M, 1, generator=g)).to(BF)
    gamma, beta = 1 + 0.3 * torch.randn(N, generator=g), 0.3 * torch.randn(N, generator=g)
    rf = resid.float()
    rstats = torch.stack([rf.mean(1), torch.rsqrt(rf.var(1, unbiased=False) + 1e-5)], 1).contiguous()
    part = torch.full((M, N // 64, 2), float("nan"), device="cuda")
    out = ops.gemm_ln(a.cuda(), w.cuda(), b.cuda(), 2, residual=resid.cuda(), res_stats=rstats.cuda(), res_gamma=gamma.cuda(), res_beta=beta.cuda(),
                      ln_partial=part)
    assert out is not None
    x = F.layer_norm(rf, (N,), gamma, beta, 1e-5).to(BF).float()
    ref = a.float() @ w.float().t() + b + x
    torch.testing.assert_close(out.float().cpu(), ref, atol=4e-2, rtol=3e-2)
    assert torch.isfinite(part).all()                                   # every (row, strip) slot was written exactly by its owner
    st = ops.ln_stats_finalize(part, N).cpu()
    of = out.float().cpu()
    torch.testing.assert_close(st[:, 0], of.mean(1), atol=2e-4, rtol=1e-4)
    torch.testing.assert_close(st[:, 1], torch.rsqrt(of.var(1, unbiased=False) + 1e-5), atol=1e-4, rtol=1e-3)
    # run-to-run bitwise stable (no atomics in the statistics)
    part2 = torch.empty_like(part)
    out2 = ops.gemm_ln(a.cuda(), w.cuda(), b.cuda(), 2, residual=resid.cuda(), res_stats=rstats.cuda(), res_gamma=gamma.cuda(), res_beta=beta.cuda(),
                       ln_partial=part2)
    assert torch.equal(out, out2) and torch.equal(part, part2)


def test_gemm_ln_declines_shapes_outside_the_tile_kernel():
    from speechclip_amd import ops
    a = torch.zeros(100, 128, device="cuda", dtype=BF)
    w = torch.zeros(128, 128, device="cuda", dtype=BF)
    z = torch.zeros(128, device="cuda")
    assert ops.gemm_ln(a, w, z, 1, ln_stats=torch.zeros(100, 2, device="cuda"), ln_c=z) is None


@pytest.mark.parametrize("n,rows,D", [(13, 301, 768), (3, 7, 128)])
def test_weighted_sum_over_pre_norm_rows(n, rows, D):
    from speechclip_amd import ops
    g = _g(n + rows)
    h0 = torch.randn(rows, D, generator=g).to(BF)
    ypre = (2 * torch.randn(n - 1, rows, D, generator=g) + 0.3).to(BF)
    gamma, beta = 1 + 0.2 * torch.randn(n - 1, D, generator=g), 0.2 * torch.randn(n - 1, D, generator=g)
    w = torch.randn(n, generator=g)
    out = ops.weighted_sum_ln(h0.cuda(), ypre.cuda(), gamma.cuda(), beta.cuda(), w.cuda()).float().cpu()
    layers = [h0.float()] + [F.layer_norm(ypre[i].float(), (D,), gamma[i], beta[i], 1e-5).to(BF).float() for i in range(n - 1)]
    ref = (torch.softmax(w, 0).view(-1, 1, 1) * torch.stack(layers)).sum(0)
    torch.testing.assert_close(out, ref, atol=1.5e-2, rtol=1.5e-2)
    # identical (up to the LayerNorm's own rounding) to the unfused kernels: layernorm -> stacked states -> weighted_sum
    hid = torch.stack([h0.cuda()] + [ops.layernorm(ypre[i].cuda(), gamma[i].cuda(), beta[i].cuda()) for i in range(n - 1)])
    unf = ops.weighted_sum(hid, w.cuda()).float().cpu()
    assert (out - unf).abs().max().item() < 2e-2


def test_encoder_with_folded_layernorms_matches_the_unfolded_path(monkeypatch):
    """HuBERT-base dims, mixed lengths: the eval fast path (no LayerNorm kernel, pre-norm states, layer mix over them) against the path that
    materialises every hidden state -- same weights, same batch.  Also checks that the fast path really ran without LayerNorm launches on the
    transformer rows, and that asking for the hidden states falls back to the materialising path."""
    from helpers import assert_rows_match, make_config
    from speechclip_amd import ops
    from speechclip_amd.model import KWClip_GeneralTransformer
    torch.manual_seed(77)
    model = KWClip_GeneralTransformer(make_config()).eval()
    g = _g(5)
    with torch.no_grad():
        model.audio_encoder.weightedsum_layer.weights.copy_(0.5 * torch.randn(13, generator=g))
        for m in model.audio_encoder.encoder.modules():
            if isinstance(m, torch.nn.LayerNorm):
                m.weight.add_(0.2 * torch.randn(m.weight.shape, generator=g)); m.bias.add_(0.2 * torch.randn(m.bias.shape, generator=g))
    model = model.cuda()
    lens = [48000, 30001, 16000, 48000]
    wav = torch.zeros(4, max(lens))
    for i, l in enumerate(lens):
        wav[i, :l] = 0.1 * torch.randn(l, generator=g)
    wav, wl = wav.cuda(), torch.tensor(lens).cuda()
    enc = model.audio_encoder
    monkeypatch.setenv("SC_FOLD_LN", "1")                                 # the folded path is opt-in (see HubertModel.fold_ln_supported)
    assert enc.encoder.fold_ln_supported(4, max(lens))
    calls = []
    real_ln = ops.layernorm
    monkeypatch.setattr(ops, "layernorm", lambda x, *a, **k: (calls.append(tuple(x.shape)), real_ln(x, *a, **k))[1])
    with torch.no_grad():
        fast, flen = enc(wav, wl)
        n_fast = len([c for c in calls if c[-1] == 768])
        calls.clear()
        slow, flen2, hidden = enc(wav, wl, return_hidden_states=True)
        n_slow = len([c for c in calls if c[-1] == 768])
    assert n_fast == 0 and n_slow == 24                                  # 2 LayerNorm passes per layer only on the materialising path
    assert torch.equal(flen, flen2) and fast.shape == slow.shape
    for b, n in enumerate(flen.tolist()):
        cos = F.cosine_similarity(fast[b, :n].float().reshape(1, -1), slow[b, :n].float().reshape(1, -1)).item()
        assert cos > 0.9995, (b, cos)
    monkeypatch.undo()
    monkeypatch.setenv("SC_FOLD_LN", "1")
    batch = {"wav": wav, "wav_len": wl, "image": torch.randn(4, 3, 224, 224, generator=g).cuda(), "id": torch.arange(4).cuda()}
    with torch.no_grad():
        lf_fast, _, _ = model(batch)
        monkeypatch.setenv("SC_FOLD_LN", "0")
        lf_slow, _, _ = model(batch)
    assert_rows_match(lf_fast["parallel_audio_feat"], lf_slow["parallel_audio_feat"], 0.99, "folded vs unfolded parallel_audio_feat")

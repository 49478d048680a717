"""Backward / optimizer kernels of the trainable tail vs torch autograd (fp32) on the same inputs.  Tolerances are fp32 round-off
(different summation orders), except where bf16 frames enter (stated inline)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda", 0)


@pytest.mark.parametrize("transa,transb", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K", [(256, 768, 3072), (1, 768, 96), (37, 130, 19), (768, 3072, 256), (8, 768, 128000 // 50)])
def test_sgemm(transa, transb, M, N, K):
    from speechclip_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N * 3 + K)
    a = torch.randn((K, M) if transa else (M, K), generator=g).to(dev())
    b = torch.randn((N, K) if transb else (K, N), generator=g).to(dev())
    bias = torch.randn(N, generator=g).to(dev())
    c0 = torch.randn(M, N, generator=g).to(dev())
    ref = 0.5 * ((a.t() if transa else a).double() @ (b.t() if transb else b).double()) + 2.0 * c0.double() + bias.double()
    out = ops.sgemm(a, b, transa, transb, alpha=0.5, beta=2.0, out=c0.clone(), bias=bias)
    tol = 1e-5 * math.sqrt(K) * 4
    assert (out.double() - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item() / 10)
    # strided operands (row slices of wider matrices)
    wide = torch.randn(a.shape[0], a.shape[1] + 8, generator=g).to(dev())
    av = wide[:, 4:4 + a.shape[1]]
    out2 = ops.sgemm(av, b, transa, transb)
    ref2 = (av.t() if transa else av).double() @ (b.t() if transb else b).double()
    assert (out2.double() - ref2).abs().max().item() < tol * max(1.0, ref2.abs().max().item() / 10)


def test_sgemm_split_k_weight_grad_shape():
    """dW = dY^T X with many rows: the split-K (atomic) path."""
    from speechclip_amd import ops
    g = torch.Generator().manual_seed(3)
    dy, x = torch.randn(4096, 48, generator=g).to(dev()), torch.randn(4096, 96, generator=g).to(dev())
    out = ops.sgemm(dy, x, transa=True)
    ref = dy.double().t() @ x.double()
    assert (out.double() - ref).abs().max().item() < 2e-3


def test_layernorm_bwd_matches_autograd():
    from speechclip_amd import ops
    g = torch.Generator().manual_seed(11)
    for rows, D in [(256, 768), (5, 1024), (33, 128)]:
        x = (torch.randn(rows, D, generator=g) * 2 + 0.5).to(dev()).requires_grad_(True)
        gamma = torch.randn(D, generator=g).to(dev()).requires_grad_(True)
        beta = torch.randn(D, generator=g).to(dev()).requires_grad_(True)
        dy = torch.randn(rows, D, generator=g).to(dev())
        torch.nn.functional.layer_norm(x, (D,), gamma, beta, 1e-5).backward(dy)
        dg, db = torch.zeros(D, device=dev()), torch.zeros(D, device=dev())
        dx = ops.layernorm_bwd(x.detach(), dy, gamma.detach(), dg, db)
        assert (dx - x.grad).abs().max().item() < 2e-5 * max(1.0, x.grad.abs().max().item())
        assert (dg - gamma.grad).abs().max().item() < 1e-4 * max(1.0, gamma.grad.abs().max().item())
        assert (db - beta.grad).abs().max().item() < 1e-4 * max(1.0, beta.grad.abs().max().item())
        # accumulate_dx adds onto an existing gradient
        base = torch.randn(rows, D, generator=g).to(dev())
        dx2 = ops.layernorm_bwd(x.detach(), dy, gamma.detach(), dx=base.clone(), accumulate_dx=True)
        assert (dx2 - (base + x.grad)).abs().max().item() < 3e-5 * max(1.0, x.grad.abs().max().item())


def test_gelu_fwd_bwd_exact_erf():
    from speechclip_amd import ops
    z = torch.linspace(-9, 9, 20001, device=dev()).requires_grad_(True)
    y = torch.nn.functional.gelu(z)
    dh = torch.randn(20001, generator=torch.Generator().manual_seed(1)).to(dev())
    y.backward(dh)
    assert (ops.gelu_f32(z.detach()) - y.detach()).abs().max().item() < 1e-6
    assert (ops.gelu_bwd_(z.detach(), dh.clone()) - z.grad).abs().max().item() < 1e-5


def test_colsum_l2norm_bwd_mix_softmax_bwd():
    from speechclip_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(256, 3072, generator=g).to(dev())
    assert (ops.colsum(x) - x.double().sum(0).float()).abs().max().item() < 1e-4
    acc = torch.ones(3072, device=dev())
    ops.colsum(x, acc, accumulate=True)
    assert (acc - 1 - x.double().sum(0).float()).abs().max().item() < 1e-4
    f = torch.randn(256, 512, generator=g).to(dev()).requires_grad_(True)
    dy = torch.randn(256, 512, generator=g).to(dev())
    (f / f.norm(dim=-1, keepdim=True)).backward(dy)
    assert (ops.l2norm_bwd(f.detach(), dy) - f.grad).abs().max().item() < 1e-6
    w = torch.randn(13, generator=g).to(dev()).requires_grad_(True)
    dalpha_b = torch.randn(256, 13, generator=g).to(dev())
    (torch.softmax(w, 0) * dalpha_b.sum(0)).sum().backward()
    dw = torch.full((13,), 0.25, device=dev())
    ops.mix_softmax_bwd(w.detach(), dalpha_b, dw)
    assert (dw - 0.25 - w.grad).abs().max().item() < 1e-5


@pytest.mark.parametrize("Bg,E,dup,margin,dcl,a2b,b2a", [(256, 512, False, 0.0, False, True, True), (300, 768, True, 0.0, False, True, True),
                                                          (64, 512, True, 0.2, False, True, False), (130, 512, True, 0.0, True, False, True)])
def test_infonce_backward_matches_autograd(Bg, E, dup, margin, dcl, a2b, b2a):
    from oracle.speechclip_ref import masked_contrastive_loss
    from speechclip_amd import ops
    g = torch.Generator().manual_seed(Bg + E)
    a = torch.nn.functional.normalize(torch.randn(Bg, E, generator=g), dim=-1)
    b = torch.nn.functional.normalize(torch.randn(Bg, E, generator=g), dim=-1)
    ids = torch.arange(Bg)
    if dup:
        ids[1::7] = ids[0::7][: len(ids[1::7])]
    inv_t = torch.tensor(1 / 0.07, requires_grad=True)
    ar = a.clone().requires_grad_(True)
    loss = masked_contrastive_loss(ar, b, ids, inv_t, margin=margin, dcl=dcl, a2b=a2b, b2a=b2a)
    loss.backward()
    out, da, dinv = ops.infonce_fwd_bwd(a.to(dev()), b.to(dev()), ids.to(dev()), 1 / 0.07, margin, dcl, a2b, b2a)
    assert abs(out[0].item() - loss.item()) < 1e-4
    assert (da.cpu() - ar.grad).abs().max().item() < 2e-5 * max(1.0, ar.grad.abs().max().item() * 10)
    assert abs(dinv.item() - inv_t.grad.item()) < 1e-5 * max(1.0, abs(inv_t.grad.item()) * 10)


def _pool_reference(x16, cls, u, beta, lens, NQ, H, keepmask=None, keep_scale=1.0):
    """fp32 autograd model of the algebraic pooling: scores z.u_r + beta_r over [CLS ; valid frames], softmax, (dropout), weighted sums."""
    B, T, D = x16.shape
    R = NQ * H
    z = torch.cat([cls.unsqueeze(0).expand(B, NQ, D), x16], 1)                     # [B, NQ+T, D]
    s = torch.einsum("bkd,rd->brk", z, u) + beta.view(1, R, 1)
    valid = torch.arange(NQ + T).view(1, 1, -1) < (lens.view(B, 1, 1) + NQ)
    s = s.masked_fill(~valid, float("-inf"))
    p = torch.softmax(s, -1)
    pp = p if keepmask is None else p * keepmask * keep_scale
    return p, torch.einsum("brk,bkd->brd", pp, z)


@pytest.mark.parametrize("B,T,D,NQ,H,n,normalize", [(5, 37, 768, 1, 8, 13, False), (3, 50, 128, 8, 1, 0, False), (4, 21, 1024, 1, 8, 5, True)])
def test_cls_pool_train_fwd_bwd_matches_autograd(B, T, D, NQ, H, n, normalize):
    from speechclip_amd import ops
    R = NQ * H
    g = torch.Generator().manual_seed(B * 100 + T)
    hid = torch.randn(max(n, 1), B, T, D, generator=g).to(torch.bfloat16)
    alpha = torch.softmax(torch.randn(max(n, 1), generator=g), 0).requires_grad_(True)
    hsrc = hid.float()
    if normalize:
        hsrc = torch.nn.functional.layer_norm(hsrc, (D,))
    xmix = torch.einsum("n,nbtd->btd", alpha, hsrc)
    x16 = xmix.to(torch.bfloat16)
    xr = x16.float().detach().requires_grad_(True)          # the kernels see the bf16-rounded frames
    cls = torch.randn(NQ, D, generator=g).requires_grad_(True)
    u = (torch.randn(R, D, generator=g) * D ** -0.5).requires_grad_(True)
    beta = torch.randn(R, generator=g)
    lens = torch.randint(1, T + 1, (B,), generator=g)
    lens[0] = T
    dzbar = torch.randn(B, R, D, generator=g)
    p_ref, zbar_ref = _pool_reference(xr, cls, u, beta, lens, NQ, H)
    zbar_ref.backward(dzbar)
    # d alpha through the (unrounded) mix, as the reference autograd would see it: dalpha_n = sum dx . H_n
    dalpha_ref = torch.einsum("btd,nbtd->bn", xr.grad, hsrc) if n else None

    d = dev()
    x_rows = x16.reshape(B * T, D).to(d)
    scores = (x_rows.float() @ u.detach().to(d).t() + beta.to(d)).contiguous()
    cls_scores = (cls.detach().to(d) @ u.detach().to(d).t() + beta.to(d)).contiguous()
    lens_i = lens.to(d, torch.int32)
    p, xbar = ops.cls_pool_train_fwd(x_rows, cls.detach().to(d).contiguous(), scores, cls_scores, lens_i, B, T, NQ, R, D)
    assert (p.cpu() - p_ref.detach()).abs().max().item() < 2e-5
    assert (xbar.cpu() - zbar_ref.detach()).abs().max().item() < 2e-4
    du, dck, dalpha = ops.cls_pool_bwd(x_rows, cls.detach().to(d).contiguous(), hid.reshape(max(n, 1), B * T, D).to(d) if n else None, p, dzbar.to(d),
                                       u.detach().to(d).contiguous(), lens_i, B, T, NQ, R, D, normalize=normalize)
    scale = max(1.0, u.grad.abs().max().item())
    assert (du.sum(0).cpu() - u.grad).abs().max().item() < 5e-4 * scale
    # CLS tokens as keys: cls.grad of the reference = sum_b dz of the CLS key rows
    assert (dck.sum(0).cpu() - cls.grad).abs().max().item() < 5e-4 * max(1.0, cls.grad.abs().max().item())
    if n:
        assert (dalpha.view(B, -1, n).sum(1).cpu() - dalpha_ref).abs().max().item() < 2e-3 * max(1.0, dalpha_ref.abs().max().item())
        one = ops.cls_pool_bwd(x_rows, cls.detach().to(d).contiguous(), hid.reshape(n, B * T, D).to(d), p, dzbar.to(d), u.detach().to(d).contiguous(),
                               lens_i, B, T, NQ, R, D, normalize=normalize, nsplit=1)
        assert one[2].shape == (B, n) and (one[2].cpu() - dalpha_ref).abs().max().item() < 2e-3 * max(1.0, dalpha_ref.abs().max().item())
        assert (one[0].sum(0) - du.sum(0)).abs().max().item() < 1e-3 * scale


def test_cls_pool_dropout_is_consistent_between_forward_and_backward():
    """With attention dropout the forward's keep-mask must be the one the backward uses: recover the mask from p' / p and check the
    gradients against autograd with that mask; the kept fraction must be close to 1 - p."""
    from speechclip_amd import ops
    B, T, D, NQ, H = 6, 200, 256, 1, 8
    R = NQ * H
    g = torch.Generator().manual_seed(9)
    x16 = torch.randn(B, T, D, generator=g).to(torch.bfloat16)
    cls = torch.randn(NQ, D, generator=g).requires_grad_(True)
    u = (torch.randn(R, D, generator=g) * D ** -0.5).requires_grad_(True)
    beta = torch.zeros(R)
    lens = torch.full((B,), T)
    dzbar = torch.randn(B, R, D, generator=g)
    d = dev()
    x_rows = x16.reshape(B * T, D).to(d)
    scores = (x_rows.float() @ u.detach().to(d).t()).contiguous()
    cls_scores = (cls.detach().to(d) @ u.detach().to(d).t()).contiguous()
    lens_i = lens.to(d, torch.int32)
    pd, seed = 0.1, 1234
    p, xbar = ops.cls_pool_train_fwd(x_rows, cls.detach().to(d).contiguous(), scores, cls_scores, lens_i, B, T, NQ, R, D, pd, seed)
    p0, xbar0 = ops.cls_pool_train_fwd(x_rows, cls.detach().to(d).contiguous(), scores, cls_scores, lens_i, B, T, NQ, R, D, 0.0, seed)
    assert torch.equal(p, p0) and not torch.allclose(xbar, xbar0)
    # recover the mask by probing the backward's p' workspace through a second call with a one-hot dzbar is overkill: rebuild it from
    # linearity -- pooled sum with x = identity-like probes is not available, so compare against autograd using the mask implied by
    # xbar: solve per (b, r) is ill-posed; instead check determinism + statistics + gradient consistency by finite differences on u.
    p2, xbar2 = ops.cls_pool_train_fwd(x_rows, cls.detach().to(d).contiguous(), scores, cls_scores, lens_i, B, T, NQ, R, D, pd, seed)
    assert torch.equal(xbar, xbar2)
    p3, xbar3 = ops.cls_pool_train_fwd(x_rows, cls.detach().to(d).contiguous(), scores, cls_scores, lens_i, B, T, NQ, R, D, pd, seed + 1)
    assert not torch.equal(xbar, xbar3)
    # E[xbar] = xbar0 (inverted dropout): average over seeds
    acc = torch.zeros_like(xbar0)
    K = 64
    for s in range(K):
        acc += ops.cls_pool_train_fwd(x_rows, cls.detach().to(d).contiguous(), scores, cls_scores, lens_i, B, T, NQ, R, D, pd, 100 + s)[1]
    rel = ((acc / K - xbar0).norm() / xbar0.norm()).item()
    assert rel < 0.08, rel
    # backward consistency: directional derivative of L = <xbar, dzbar> along a random du direction, by central differences on the
    # forward kernel with the SAME seed (the mask is fixed, so L is smooth in u)
    du, dck, _ = ops.cls_pool_bwd(x_rows, cls.detach().to(d).contiguous(), None, p, dzbar.to(d), u.detach().to(d).contiguous(), lens_i, B, T, NQ, R, D,
                                  drop_p=pd, seed=seed)
    dirn = torch.randn(R, D, generator=g).to(d) * D ** -0.5
    eps = 1e-2

    def L(uu):
        sc = (x_rows.float() @ uu.t()).contiguous()
        cs = (cls.detach().to(d) @ uu.t()).contiguous()
        return (ops.cls_pool_train_fwd(x_rows, cls.detach().to(d).contiguous(), sc, cs, lens_i, B, T, NQ, R, D, pd, seed)[1].double() * dzbar.to(d).double()).sum().item()
    fd = (L(u.detach().to(d) + eps * dirn) - L(u.detach().to(d) - eps * dirn)) / (2 * eps)
    an = (du.sum(0).double() * dirn.double()).sum().item()
    assert abs(fd - an) < 2e-2 * max(1.0, abs(an)), (fd, an)


def test_dropout_f32_statistics_and_determinism():
    from speechclip_amd import ops
    x = torch.ones(1 << 20, device=dev())
    y = ops.dropout_f32(x, 0.1, 77)
    kept = (y != 0).float().mean().item()
    assert abs(kept - 0.9) < 3e-3
    assert abs(y.mean().item() - 1.0) < 5e-3 and abs(y.max().item() - 1 / 0.9) < 1e-6
    assert torch.equal(y, ops.dropout_f32(x, 0.1, 77)) and not torch.equal(y, ops.dropout_f32(x, 0.1, 78))
    assert torch.equal(ops.dropout_f32(x, 0.0, 5), x)


def test_adam_and_grad_clip_match_torch():
    from speechclip_amd import ops
    g = torch.Generator().manual_seed(2)
    n = 100003
    p0 = torch.randn(n, generator=g)
    pt = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pt], lr=1e-3, weight_decay=1e-6)
    p, m, v = p0.clone().to(dev()), torch.zeros(n, device=dev()), torch.zeros(n, device=dev())
    for step in range(1, 6):
        grad = torch.randn(n, generator=g) * (3.0 if step % 2 else 0.01)
        pt.grad = grad.clone()
        tn = torch.nn.utils.clip_grad_norm_([pt], 4.0)
        opt.step()
        gd = grad.to(dev())
        nc = ops.grad_norm(gd, 4.0)
        assert abs(nc[0].item() - tn.item()) < 1e-3 * tn.item()
        ops.adam_step(p, gd, m, v, step, 1e-3, weight_decay=1e-6, clip_coef=nc)
        assert (p.cpu() - pt.detach()).abs().max().item() < 2e-6


def test_sgemm_batched_heads():
    from speechclip_amd import ops
    g = torch.Generator().manual_seed(4)
    B, H, hd, D = 37, 8, 96, 768
    zbar = torch.randn(B, H, D, generator=g).to(dev())
    Wv = torch.randn(D, D, generator=g).to(dev())
    bv = torch.randn(D, generator=g).to(dev())
    att = torch.empty(B, D, device=dev())
    ops.sgemm_batched(B, hd, D, zbar, H * D, D, Wv, D, hd * D, att, D, hd, H, transb=True, bias=bv, stride_bias=hd)
    ref = torch.einsum("bhd,hjd->bhj", zbar.double(), Wv.double().view(H, hd, D)).reshape(B, D) + bv.double()
    assert (att.double() - ref).abs().max().item() < 2e-3
    datt = torch.randn(B, D, generator=g).to(dev())
    dW = torch.empty(D, D, device=dev())
    ops.sgemm_batched(hd, D, B, datt, D, hd, zbar, H * D, D, dW, D, hd * D, H, transa=True)
    refw = torch.einsum("bhj,bhd->hjd", datt.double().view(B, H, hd), zbar.double()).reshape(D, D)
    assert (dW.double() - refw).abs().max().item() < 2e-3


def test_train_mode_crop_pad_matches_per_utterance_path():
    """The batched train-mode crop (sc_crop_pad) draws the same numpy offsets as the reference's per-utterance loop and yields the same batch."""
    import numpy as np
    from speechclip_amd import ops
    from speechclip_amd.module.speech_encoder_plus import random_crop_max_length
    g = torch.Generator().manual_seed(8)
    lens = [9000, 2500, 6401, 6400, 12000]
    wav = torch.zeros(len(lens), max(lens))
    for i, l in enumerate(lens):
        wav[i, :l] = torch.randn(l, generator=g)
    max_len = 6400
    np.random.seed(5)
    ref = [random_crop_max_length(wav[i, :l], max_len, l) for i, l in enumerate(lens)]
    np.random.seed(5)
    starts, outl = [], []
    for n in lens:
        if n <= max_len:
            starts.append(0); outl.append(n)
        else:
            starts.append(int(np.random.randint(n - max_len))); outl.append(max_len)
    out = ops.crop_pad(wav.to(dev()), starts, outl, max(outl)).cpu()
    for i, r in enumerate(ref):
        assert torch.equal(out[i, :len(r)], r) and bool((out[i, len(r):] == 0).all())


# ---------------------------------------------------------------- cascaded tail (train_cascaded.hip)
@pytest.mark.parametrize("B,L,H,causal", [(5, 10, 8, True), (3, 16, 12, True), (2, 7, 8, False), (1, 1, 8, True)])
def test_attn_small_bwd(B, L, H, causal):
    """dqkv of softmax(q k^T / 8 [+ causal mask]) v over the K+2 live text positions vs torch autograd on the same bf16 qkv."""
    from speechclip_amd import ops
    g = torch.Generator().manual_seed(B * 100 + L)
    W = H * 64
    qkv16 = (0.7 * torch.randn(B * L, 3 * W, generator=g)).to(dev(), torch.bfloat16)
    dout = torch.randn(B * L, W, generator=g).to(dev())
    x = qkv16.float().requires_grad_(True)
    q, k, v = [t.view(B, L, H, 64).transpose(1, 2) for t in x.split(W, dim=1)]
    s = q @ k.transpose(-1, -2) / 8.0
    if causal:
        s = s + torch.full((L, L), float("-inf"), device=dev()).triu(1)
    o = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * L, W)
    o.backward(dout)
    got = ops.attn_small_bwd(qkv16, dout, B, L, H, causal)
    torch.testing.assert_close(got, x.grad, atol=2e-5, rtol=1e-4)


def test_quickgelu_fwd_bwd():
    from speechclip_amd import ops
    z = (3 * torch.randn(1000, 37, generator=torch.Generator().manual_seed(5))).to(dev()).requires_grad_(True)
    ref = z * torch.sigmoid(1.702 * z)
    dh = torch.randn(1000, 37, generator=torch.Generator().manual_seed(6)).to(dev())
    ref.backward(dh)
    torch.testing.assert_close(ops.quickgelu_f32(z.detach()), ref.detach(), atol=1e-6, rtol=1e-5)
    torch.testing.assert_close(ops.quickgelu_f32(z.detach(), out_bf16=True).float(), ref.detach(), atol=2e-2, rtol=1e-2)
    torch.testing.assert_close(ops.quickgelu_bwd_(z.detach(), dh.clone()), z.grad, atol=1e-5, rtol=1e-4)


@pytest.mark.parametrize("R,V,E,temp", [(48, 1000, 64, 0.1), (16, 49408, 512, 0.1), (7, 333, 32, 1.0)])
def test_vq_straight_through_and_cosine_bwd(R, V, E, temp):
    """keywords = (hard + soft - soft.detach()) @ emb with soft = softmax(masked cos / temp), cos = F.cosine_similarity(a, emb):
    d loss / d a from sc_sgemm + sc_vq_st_bwd + sc_sgemm + sc_cosine_bwd_finish vs autograd of exactly that graph
    (my_vector_quantizer.py:75-141, kwClip.py:889-911)."""
    import torch.nn.functional as F
    from speechclip_amd import ops
    g = torch.Generator().manual_seed(R + V)
    emb = torch.randn(V, E, generator=g).to(dev())
    a = (torch.randn(R, E, generator=g) + 0.3).to(dev()).requires_grad_(True)
    dkw = torch.randn(R, E, generator=g).to(dev())
    cos = F.cosine_similarity(a.unsqueeze(2), emb.t().unsqueeze(0), dim=1)          # [R, V]
    x = cos.clone()
    for i in (0, 2, 3):
        x[:, i] += float("-inf")
    hard = torch.zeros_like(x).scatter_(-1, x.argmax(-1, keepdim=True), 1.0)
    soft = torch.softmax(x / temp, -1)
    kw = (hard + soft - soft.detach()) @ emb
    kw.backward(dkw)
    cos_mine = ops.cosine_scores(a.detach(), emb)
    torch.testing.assert_close(cos_mine, cos.detach(), atol=2e-6, rtol=1e-5)
    dprob = ops.sgemm(dkw, emb, transb=True)                                         # d loss / d subword_prob
    rowdot = ops.vq_st_bwd_(cos_mine, dprob, temp)
    G = ops.sgemm(dprob, ops.l2norm(emb))
    da = ops.cosine_bwd_finish(a.detach(), G, rowdot)
    scale = a.grad.abs().max().item()
    assert (da - a.grad).abs().max().item() < 2e-4 * max(scale, 1e-3), ((da - a.grad).abs().max().item(), scale)


@pytest.mark.parametrize("B,K,E", [(6, 8, 16), (256, 8, 512), (3, 1, 5)])
def test_kw_batchnorm_train_fwd_bwd(B, K, E):
    """Kw_BatchNorm eachKw+parallel in train mode = nn.BatchNorm1d(E*K) over the (B, E, K)-flattened keywords (kw_bn.py:122-131)."""
    from speechclip_amd import ops
    g = torch.Generator().manual_seed(B + K + E)
    bn = torch.nn.BatchNorm1d(E * K).to(dev()).train()
    with torch.no_grad():
        bn.weight.copy_(1 + 0.3 * torch.randn(E * K, generator=g))
        bn.bias.copy_(0.2 * torch.randn(E * K, generator=g))
        bn.running_mean.copy_(0.1 * torch.randn(E * K, generator=g))
        bn.running_var.copy_(1 + 0.1 * torch.rand(E * K, generator=g))
    rm, rv = bn.running_mean.clone(), bn.running_var.clone()
    x = (2 * torch.randn(B, K, E, generator=g) + 0.5).to(dev()).requires_grad_(True)
    dy = torch.randn(B, K, E, generator=g).to(dev())
    ref = bn(x.permute(0, 2, 1).reshape(B, -1)).reshape(B, E, K).permute(0, 2, 1)
    ref.backward(dy)
    y, mean, rstd = ops.kw_bn_train_fwd(x.detach().contiguous(), bn.weight.detach(), bn.bias.detach(), rm, rv, bn.momentum, bn.eps)
    torch.testing.assert_close(y, ref.detach(), atol=2e-5, rtol=1e-5)
    torch.testing.assert_close(rm, bn.running_mean, atol=1e-6, rtol=1e-5)
    torch.testing.assert_close(rv, bn.running_var, atol=1e-5, rtol=1e-5)
    dx, dg, db = ops.kw_bn_bwd(x.detach().contiguous(), dy, bn.weight.detach(), mean, rstd)
    torch.testing.assert_close(dx, x.grad, atol=5e-5, rtol=1e-4)
    torch.testing.assert_close(dg, bn.weight.grad, atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(db, bn.bias.grad, atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("R,V,E", [(64, 1024, 64), (16, 49408, 512)])
def test_keyword_st_fn_mfma_path(R, V, E):
    """train_tail.KeywordSTFn end to end (the two [R,V] products on the MFMA GEMM with hi+lo split gradients) vs autograd of
    (hard + soft - soft.detach()) @ emb, soft = softmax(masked cosine / 0.1)."""
    import torch.nn.functional as F
    from speechclip_amd import ops
    from speechclip_amd.train_tail import KeywordSTFn
    g = torch.Generator().manual_seed(R + V + 1)
    emb = torch.nn.Parameter(torch.randn(V, E, generator=g).to(dev()), requires_grad=False)
    a = (torch.randn(R, E, generator=g) + 0.3).to(dev()).requires_grad_(True)
    dkw = torch.randn(R, E, generator=g).to(dev())
    cos = F.cosine_similarity(a.unsqueeze(2), emb.t().unsqueeze(0), dim=1)
    x = cos.clone()
    x[:, [0, 2, 3]] = float("-inf")
    tgt = x.argmax(-1)
    soft = torch.softmax(x / 0.1, -1)
    kw = (torch.zeros_like(x).scatter_(-1, tgt[:, None], 1.0) + soft - soft.detach()) @ emb
    kw.backward(dkw)
    a2 = a.detach().clone().requires_grad_(True)
    cos2 = ops.cosine_scores(a2.detach(), emb)
    kw2 = KeywordSTFn.apply(a2, cos2, tgt, emb, 0.1, (0, 2, 3))
    torch.testing.assert_close(kw2, kw.detach(), atol=1e-5, rtol=1e-5)
    kw2.backward(dkw)
    scale = a.grad.abs().max().item()
    err = (a2.grad - a.grad).abs().max().item()
    assert err < 2e-3 * scale, (err, scale)          # bf16 sub-word table inside the two products (hi+lo split on the gradient side)
    # LEARNABLE temperature (vq.temp: "learnable=0.1", my_vector_quantizer.py:33-38): the parameter goes in, its gradient comes out
    tp = torch.nn.Parameter(torch.tensor([0.1], device=dev()))
    a3 = a.detach().clone().requires_grad_(True)
    cos3 = F.cosine_similarity(a3.unsqueeze(2), emb.t().unsqueeze(0), dim=1)
    x3 = cos3.clone()
    x3[:, [0, 2, 3]] = -1e4      # finite stand-in for the reference's -inf mask: autograd of (-inf / T) w.r.t. T is 0 * inf = NaN (the reference's own learnable
    soft3 = torch.softmax(x3 / tp, -1)      # temperature gets NaN gradients through its masked columns); exp(-1e5) is exactly 0, so the forward is identical
    ((torch.zeros_like(x3).scatter_(-1, tgt[:, None], 1.0) + soft3 - soft3.detach()) @ emb).backward(dkw)
    tq = torch.nn.Parameter(torch.tensor([0.1], device=dev()))
    a4 = a.detach().clone().requires_grad_(True)
    KeywordSTFn.apply(a4, ops.cosine_scores(a4.detach(), emb), tgt, emb, tq, (0, 2, 3)).backward(dkw)
    assert tq.grad is not None and tq.grad.shape == (1,)
    assert abs(tq.grad.item() - tp.grad.item()) < 5e-3 * abs(tp.grad.item()) + 1e-6, (tq.grad.item(), tp.grad.item())
    assert (a4.grad - a3.grad).abs().max().item() < 2e-3 * scale


def test_text_tower_input_gradient_vit_b32_dims():
    """TextTowerTrainFn at the real ViT-B/32 text-tower size (12 layers, width 512, 8 heads; K + 2 = 10 live positions): feature and
    d feature / d token-embeddings vs the fp32 oracle tower (oracle/clip_ref.py) on the same weights."""
    from oracle import clip_ref
    from speechclip_amd.module.clip_model import CLIP, ClipConfig
    torch.manual_seed(11)
    mine = CLIP(ClipConfig.from_name("ViT-B/32")).eval()
    ref = clip_ref.ClipRef(clip_ref.ClipRefConfig.vit_b32()).eval()
    ref.load_state_dict(mine.state_dict())
    mine = mine.to(dev())
    B, L, W = 5, 10, 512
    g = torch.Generator().manual_seed(3)
    emb = (0.05 * torch.randn(B, L, W, generator=g))
    dfeat = torch.randn(B, 512, generator=g)
    e_ref = emb.clone().requires_grad_(True)
    x = torch.zeros(B, 77, W)
    x = torch.cat([e_ref, x[:, L:]], dim=1) + ref.positional_embedding
    x = ref.ln_final(ref.transformer(x.permute(1, 0, 2)).permute(1, 0, 2))
    f_ref = x[:, L - 1] @ ref.text_projection
    f_ref.backward(dfeat)
    e_mine = emb.clone().to(dev()).requires_grad_(True)
    f = mine.encode_text_embeddings(e_mine, torch.full((B,), L - 1, device=dev(), dtype=torch.long))
    assert f.requires_grad
    cos = torch.nn.functional.cosine_similarity(f.detach().cpu().double(), f_ref.detach().double(), dim=-1).min().item()
    assert cos > 0.999, cos
    f.backward(dfeat.to(dev()))
    gm, gr = e_mine.grad.cpu().double(), e_ref.grad.double()
    assert torch.nn.functional.cosine_similarity(gm.reshape(1, -1), gr.reshape(1, -1)).item() > 0.999
    assert abs(gm.norm().item() / gr.norm().item() - 1) < 0.02

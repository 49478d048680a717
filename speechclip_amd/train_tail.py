"""Training of the trainable tail on MI355X.

The reference trains, on top of frozen HuBERT and CLIP, only the parallel branch (`KW_ParallelBranch`, kwClip.py:1004-1108), the
layer-mix weights (`WeightedSumLayer.weights`, weighted_sum.py:19) and -- for the large models -- the loss temperature
(losses.py:161); Lightning calls `training_step` -> `training_step_end` -> `loss.backward()` -> clip(4) -> Adam -> LambdaLR.
Here the same hook protocol works with torch autograd as the *plumbing*: three `torch.autograd.Function`s whose forward and backward
bodies are calls into libspeechclip_hip.so (fp32 master weights; frames stay bf16), so `loss.backward()` fills `.grad` of exactly the
parameters the reference optimises and any torch optimizer -- or `FusedAdam` below (sc_grad_norm + sc_adam_step on one flat buffer)
-- can step them.

  ParallelBranchTrainFn : hidden states -> layer mix -> [CLS; frames] encoder layer (CLS row only) -> final LN -> Linear  [B, E]
  CascadedPoolTrainFn   : hidden states -> layer mix -> K keyword queries over [CLS_1..K; frames] -> LN(MHA + CLS) -> Linear          [B*K, E_txt]
  KwBatchNormTrainFn    : train-mode keyword BatchNorm (batch statistics, running-stat update)     (kw_bn.py:122-131)
  KeywordSTFn           : cosine scores -> straight-through VQ -> sub-word embeddings              (kwClip.py:889-911)
  TextTowerTrainFn      : frozen CLIP text tower with the gradient w.r.t. its input embeddings     (clip_official.py:220-264)
  L2NormFn              : x / |x|                                                      (kwClip.py:1436)
  MaskedContrastiveFn   : masked InfoNCE on the gathered global batch                  (losses.py:185-245)
  PackedGatherFn        : ONE RCCL all-gather of all features + ids whose backward keeps the local rows (replaces DP's gather, kwClip.py:147-191)

Dropout (0.1 at four sites of nn.TransformerEncoderLayer) uses a counter-based hash RNG inside the kernels, seeded per call from
torch's generator: masks are reproducible from (seed, site) and are regenerated -- not stored -- in the backward.
"""
from typing import Optional

import torch
import torch.distributed as dist

from . import ops, parallel

BF = torch.bfloat16


def _c(t):
    return t.detach().float().contiguous()


def _dup_k(w16: torch.Tensor) -> torch.Tensor:
    """[W | W] bf16 [N, 2K]: the weight operand that pairs with ops.split_hilo."""
    return torch.cat([w16, w16], dim=1).contiguous()


def _mfma_f32(a: torch.Tensor, w2: torch.Tensor) -> torch.Tensor:
    """f32 [M,N] = a f32 [M,K] @ W^T on the MFMA GEMM, W given as w2 = [W | W] bf16 [N, 2K] (frozen): a is split into two bf16 terms
    (hi | lo) by sc_split_hilo_bf16 and ONE GEMM of depth 2K adds both products (the gradient keeps ~16 mantissa bits; a single bf16
    term would put 0.4 % noise on every backward product).  Shapes the GEMM kernel does not take (2K % 64, N % 4: reduced test
    vocabularies) go to the fp32 SIMT sgemm."""
    N, K2 = w2.shape
    if K2 % 64 or N % 4:
        return ops.sgemm(a, w2[:, :K2 // 2].float().contiguous(), transb=True)
    return ops.gemm(ops.split_hilo(a), w2, out_f32=True)


def _hidden_rows(hid, B, Tp, D):
    """hidden [n, B, Tp', D] -> contiguous [n, B*Tp, D] rows matching the frames view's row count.  With B == 1 the frames view has no batch stride
    to read the padded length Tp' from and uses Tp = T; the padding rows of the states are cut off then (a small copy, B == 1 only)."""
    if hid.shape[2] != Tp:
        assert B == 1 and hid.shape[2] > Tp, (tuple(hid.shape), B, Tp)
        hid = hid[:, :, :Tp].contiguous()
    return hid.reshape(hid.shape[0], B * Tp, D)


class ParallelBranchTrainFn(torch.autograd.Function):
    """out f32 [B, E (or D)] = linear_proj(norm(layer([CLS; mix(hidden)]))[:, 0]).

    args: meta (dict: heads, eps, drop_p, seed, normalize), hidden bf16 [n, B, Tp, D] (frozen encoder states) or None,
          x16 bf16 [B, T<=Tp, D] view of the mixed frames (what WeightedSumLayer produced from `hidden`), lens int [B],
          then tensors: mixw [n] | None, cls [1,1,D], in_w [3D,D], in_b [3D], out_w, out_b, n1w, n1b, l1w, l1b, l2w, l2b, n2w, n2b,
          nfw, nfb, pw [E,D] | None, pb [E] | None."""

    @staticmethod
    def forward(ctx, meta, hidden, x16, lens, mixw, cls, in_w, in_b, out_w, out_b, n1w, n1b, l1w, l1b, l2w, l2b, n2w, n2b, nfw, nfb, pw, pb):
        from .module.kw_modules.TransformerModels import _frames_view
        H, eps, pd, seed = meta["heads"], meta["eps"], float(meta["drop_p"]), int(meta["seed"])
        B, T, D = x16.shape
        NQ = cls.shape[-2]
        assert NQ == 1, "the parallel branch has one CLS token (kwClip.py:1040-1047)"
        hd, R = D // H, NQ * H
        scale = hd ** -0.5
        rows, Tp = _frames_view(x16)
        dev = rows.device
        lens_i = lens.to(device=dev, dtype=torch.int32).contiguous()
        c = _c(cls).view(NQ, D)
        Win, bin_ = _c(in_w), _c(in_b)
        # ---- parameter-only part: queries of the CLS token, u_r = scale Wk_h^T q_h, beta_r = scale q_h . bk_h
        qt = ops.sgemm(c, Win[:D], transb=True, bias=bin_[:D])                              # [NQ, D]
        U = torch.empty(R, D, device=dev, dtype=torch.float32)                               # rows r = q*H + h
        beta = torch.empty(R, device=dev, dtype=torch.float32)
        Wk, bk, Wv, bv = Win[D:2 * D], bin_[D:2 * D], Win[2 * D:], bin_[2 * D:]
        ops.sgemm_batched(NQ, D, hd, qt, D, hd, Wk, D, hd * D, U, H * D, D, H, alpha=scale)                      # U[q,h,:] = scale q_h^T Wk_h
        ops.sgemm_batched(NQ, 1, hd, qt, D, hd, bk, 1, hd, beta, H, 1, H, alpha=scale)                             # beta[q,h] = scale q_h . bk_h
        # ---- frame scores on the MFMA GEMM (bf16 frames x bf16 u, fp32 accumulate/out), pooling in fp32
        scores = ops.gemm(rows, U.to(BF).contiguous(), beta, out_f32=True)                    # [B*Tp, R]
        cls_scores = ops.sgemm(c, U, transb=True, bias=beta)                                  # [NQ, R]
        p, zbar = ops.cls_pool_train_fwd(rows, c, scores, cls_scores, lens_i, B, Tp, NQ, R, D, pd, seed)
        att = torch.empty(B * NQ, D, device=dev, dtype=torch.float32)
        ops.sgemm_batched(B, hd, D, zbar, R * D, D, Wv, D, hd * D, att, D, hd, H, transb=True, bias=bv, stride_bias=hd)   # o_h = Wv_h zbar_h + bv_h
        # ---- rest of the encoder layer on the CLS rows (post-LN), final norm, projection
        sa = ops.sgemm(att, _c(out_w), transb=True, bias=_c(out_b))
        if pd > 0:
            ops.dropout_f32(sa, pd, seed + 1, out=sa)
        y = ops.add_rows(sa, c)                                                               # x + dropout1(SA(x)), x = CLS token
        x1 = ops.layernorm(y, _c(n1w), _c(n1b), eps, out_f32=True)
        z1 = ops.sgemm(x1, _c(l1w), transb=True, bias=_c(l1b))
        hm = ops.gelu_f32(z1)
        if pd > 0:
            ops.dropout_f32(hm, pd, seed + 2, out=hm)
        ff = ops.sgemm(hm, _c(l2w), transb=True, bias=_c(l2b))
        if pd > 0:
            ops.dropout_f32(ff, pd, seed + 3, out=ff)
        y2 = ops.add_rows(ff, x1)
        x2 = ops.layernorm(y2, _c(n2w), _c(n2b), eps, out_f32=True)
        x3 = ops.layernorm(x2, _c(nfw), _c(nfb), 1e-5, out_f32=True)
        out = ops.sgemm(x3, _c(pw), transb=True, bias=_c(pb)) if pw is not None else x3.clone()
        ctx.meta = dict(meta, B=B, T=T, Tp=Tp, D=D, NQ=NQ, R=R, hd=hd, scale=scale)
        ctx.hidden = hidden
        ctx.has = (mixw is not None, pw is not None)
        ctx.save_for_backward(rows, lens_i, c, Win, qt, U, p, zbar, att, y, x1, z1, hm, y2, x2, x3,
                              _c(out_w), _c(n1w), _c(l1w), _c(l2w), _c(n2w), _c(nfw), _c(pw) if pw is not None else c,
                              _c(mixw) if mixw is not None else c)
        return out

    @staticmethod
    def backward(ctx, dout):
        (rows, lens_i, c, Win, qt, U, p, zbar, att, y, x1, z1, hm, y2, x2, x3, Wo, g1, W1, W2, g2, gf, Wp, mixw) = ctx.saved_tensors
        m = ctx.meta
        B, Tp, D, NQ, R, hd, H, scale, eps, pd, seed = m["B"], m["Tp"], m["D"], m["NQ"], m["R"], m["hd"], m["heads"], m["scale"], m["eps"], float(m["drop_p"]), int(m["seed"])
        has_mix, has_proj = ctx.has
        dev = rows.device
        z = lambda *s: torch.zeros(*s, device=dev, dtype=torch.float32)   # noqa: E731
        dout = dout.float().contiguous()
        # projection, final norm, norm2
        if has_proj:
            dpw = ops.sgemm(dout, x3, transa=True)
            dpb = ops.colsum(dout)
            dx3 = ops.sgemm(dout, Wp)
        else:
            dpw = dpb = None
            dx3 = dout
        dnfw, dnfb, dn2w, dn2b, dn1w, dn1b = z(D), z(D), z(D), z(D), z(D), z(D)
        dx2 = ops.layernorm_bwd(x2, dx3, gf, dnfw, dnfb, 1e-5)
        dy2 = ops.layernorm_bwd(y2, dx2, g2, dn2w, dn2b, eps)
        # FFN
        dff = ops.dropout_f32(dy2, pd, seed + 3) if pd > 0 else dy2
        dl2w = ops.sgemm(dff, hm, transa=True)
        dl2b = ops.colsum(dff)
        dhm = ops.sgemm(dff, W2)
        if pd > 0:
            ops.dropout_f32(dhm, pd, seed + 2, out=dhm)
        ops.gelu_bwd_(z1, dhm)                                                              # dhm is now dz1
        dl1w = ops.sgemm(dhm, x1, transa=True)
        dl1b = ops.colsum(dhm)
        dx1 = ops.sgemm(dhm, W1, beta=1.0, out=dy2.clone())                                   # residual + through linear1
        dy = ops.layernorm_bwd(y, dx1, g1, dn1w, dn1b, eps)
        # attention block: y = c + dropout1(att Wo^T + bo)
        dcls = ops.colsum(dy).view(NQ, D)
        dsa = ops.dropout_f32(dy, pd, seed + 1) if pd > 0 else dy
        dWo = ops.sgemm(dsa, att, transa=True)
        dbo = ops.colsum(dsa)
        datt = ops.sgemm(dsa, Wo)
        dWin, dbin = z(3 * D, D), z(3 * D)
        dzbar = torch.empty(B, R, D, device=dev, dtype=torch.float32)
        Wk, Wv = Win[D:2 * D], Win[2 * D:]
        ops.sgemm_batched(hd, D, B, datt, D, hd, zbar, R * D, D, dWin[2 * D:], D, hd * D, H, transa=True)        # dWv_h = datt_h^T zbar_h
        ops.colsum(datt, out=dbin[2 * D:])
        ops.sgemm_batched(B, D, hd, datt, D, hd, Wv, D, hd * D, dzbar, R * D, D, H)                               # dzbar_h = datt_h Wv_h
        hid = ctx.hidden
        hid2 = _hidden_rows(hid, B, Tp, D) if (has_mix and hid is not None) else None
        dx16 = None
        if ctx.needs_input_grad[2]:      # the frames themselves carry a gradient (fine-tuned encoder layers: train_hubert.py)
            du, dck, dalpha, ds_ws, pp_ws = ops.cls_pool_bwd(rows, c, hid2, p, dzbar, U, lens_i, B, Tp, NQ, R, D, normalize=bool(m.get("normalize", False)),
                                                             drop_p=pd, seed=seed, return_ws=True)
            dx16 = ops.cls_pool_dz(pp_ws, ds_ws, dzbar, U, lens_i, B, Tp, NQ, R, D).view(B, Tp, D)[:, :m["T"]]
        else:
            du, dck, dalpha = ops.cls_pool_bwd(rows, c, hid2, p, dzbar, U, lens_i, B, Tp, NQ, R, D, normalize=bool(m.get("normalize", False)), drop_p=pd, seed=seed)
        dU = ops.colsum(du.view(-1, R * D)).view(R, D)                                        # rows: B x key-splits
        ops.colsum(dck.view(-1, NQ * D), out=dcls.view(NQ * D), accumulate=True)               # CLS token as a key / value
        # parameter-only chain: u_r = scale Wk_h^T q_h (beta carries no gradient: softmax is shift invariant)
        dqt = torch.empty(NQ, D, device=dev, dtype=torch.float32)
        ops.sgemm_batched(hd, D, NQ, qt, D, hd, dU, H * D, D, dWin[D:2 * D], D, hd * D, H, transa=True, alpha=scale)   # dWk_h = scale q_h (x) dU_h
        ops.sgemm_batched(NQ, hd, D, dU, H * D, D, Wk, D, hd * D, dqt, D, hd, H, transb=True, alpha=scale)             # dq_h = scale Wk_h dU_h
        ops.sgemm(dqt, c, transa=True, out=dWin[:D])
        ops.colsum(dqt, out=dbin[:D])
        ops.sgemm(dqt, Win[:D], beta=1.0, out=dcls)
        dmix = None
        if has_mix and dalpha is not None:
            dmix = z(mixw.shape[0])
            ops.mix_softmax_bwd(mixw, dalpha, dmix)
        return (None, None, dx16, None, dmix, dcls.view(1, NQ, D), dWin, dbin, dWo, dbo, dn1w, dn1b, dl1w, dl1b, dl2w, dl2b, dn2w, dn2b, dnfw, dnfb,
                dpw, dpb)


# ================================================================= cascaded tail (kwClip.py:697-916)
def _pool_params(c, Win, bin_, NQ, D, H, hd, scale, dev):
    """Parameter-only operands of the algebraic CLS pooling: queries qt [NQ,D], U [R,D] (u_r = scale Wk_h^T q_h), beta [R]."""
    R = NQ * H
    qt = ops.sgemm(c, Win[:D], transb=True, bias=bin_[:D])
    U = torch.empty(R, D, device=dev, dtype=torch.float32)
    beta = torch.empty(R, device=dev, dtype=torch.float32)
    Wk, bk = Win[D:2 * D], bin_[D:2 * D]
    ops.sgemm_batched(NQ, D, hd, qt, D, hd, Wk, D, hd * D, U, H * D, D, H, alpha=scale)
    ops.sgemm_batched(NQ, 1, hd, qt, D, hd, bk, 1, hd, beta, H, 1, H, alpha=scale)
    return qt, U, beta


class CascadedPoolTrainFn(torch.autograd.Function):
    """kp f32 [B*K, E] = linear_proj(LN(MHA([CLS_1..K ; frames])[:, :K] + CLS))   (kwClip.py:870-884, TransformerModels.py:99-124).
    Same algebraic pooling as the parallel branch with K query tokens; attention-probability dropout from the hash RNG.

    args: meta (heads, eps, drop_p, seed, normalize), hidden bf16 [n,B,Tp,D] | None, x16 bf16 [B,T,D], lens, mixw [n] | None,
          cls [1,K,D], in_w [3D,D], in_b [3D], out_w, out_b, nw, nb (attentionBlock_Norm), pw [E,D], pb [E]."""

    @staticmethod
    def forward(ctx, meta, hidden, x16, lens, mixw, cls, in_w, in_b, out_w, out_b, nw, nb, pw, pb):
        from .module.kw_modules.TransformerModels import _frames_view
        H, eps, pd, seed = meta["heads"], meta["eps"], float(meta["drop_p"]), int(meta["seed"])
        B, T, D = x16.shape
        NQ = cls.shape[-2]
        hd, R = D // H, NQ * H
        scale = hd ** -0.5
        rows, Tp = _frames_view(x16)
        dev = rows.device
        lens_i = lens.to(device=dev, dtype=torch.int32).contiguous()
        c = _c(cls).view(NQ, D)
        Win, bin_ = _c(in_w), _c(in_b)
        qt, U, beta = _pool_params(c, Win, bin_, NQ, D, H, hd, scale, dev)
        scores = ops.gemm(rows, U.to(BF).contiguous(), beta, out_f32=True)                    # [B*Tp, R]
        cls_scores = ops.sgemm(c, U, transb=True, bias=beta)                                  # [NQ, R]
        p, zbar = ops.cls_pool_train_fwd(rows, c, scores, cls_scores, lens_i, B, Tp, NQ, R, D, pd, seed)
        att = torch.empty(B * NQ, D, device=dev, dtype=torch.float32)
        Wv, bv = Win[2 * D:], bin_[2 * D:]
        ops.sgemm_batched(B * NQ, hd, D, zbar, H * D, D, Wv, D, hd * D, att, D, hd, H, transb=True, bias=bv, stride_bias=hd)
        sa = ops.sgemm(att, _c(out_w), transb=True, bias=_c(out_b))
        y = ops.add_rows(sa, c)                                                               # + src rows (the CLS tokens)
        kn = ops.layernorm(y, _c(nw), _c(nb), eps, out_f32=True)
        kp = ops.sgemm(kn, _c(pw), transb=True, bias=_c(pb))
        ctx.meta = dict(meta, B=B, T=T, Tp=Tp, D=D, NQ=NQ, R=R, hd=hd, scale=scale)
        ctx.hidden = hidden
        ctx.has_mix = mixw is not None
        ctx.save_for_backward(rows, lens_i, c, Win, qt, U, p, zbar, att, y, kn, _c(out_w), _c(nw), _c(pw), _c(mixw) if mixw is not None else c)
        return kp

    @staticmethod
    def backward(ctx, dkp):
        rows, lens_i, c, Win, qt, U, p, zbar, att, y, kn, Wo, gn, Wp, mixw = ctx.saved_tensors
        m = ctx.meta
        B, Tp, D, NQ, R, hd, H, scale, eps, pd, seed = m["B"], m["Tp"], m["D"], m["NQ"], m["R"], m["hd"], m["heads"], m["scale"], m["eps"], float(m["drop_p"]), int(m["seed"])
        dev = rows.device
        z = lambda *s: torch.zeros(*s, device=dev, dtype=torch.float32)   # noqa: E731
        dkp = dkp.float().contiguous()
        dpw = ops.sgemm(dkp, kn, transa=True)
        dpb = ops.colsum(dkp)
        dkn = ops.sgemm(dkp, Wp)
        dnw, dnb = z(D), z(D)
        dy = ops.layernorm_bwd(y, dkn, gn, dnw, dnb, eps)
        dcls = ops.colsum(dy.view(B, NQ * D)).view(NQ, D).contiguous()                        # residual path: src rows are the CLS tokens
        dWo = ops.sgemm(dy, att, transa=True)
        dbo = ops.colsum(dy)
        datt = ops.sgemm(dy, Wo)
        dWin, dbin = z(3 * D, D), z(3 * D)
        dzbar = torch.empty(B, R, D, device=dev, dtype=torch.float32)
        Wk, Wv = Win[D:2 * D], Win[2 * D:]
        ops.sgemm_batched(hd, D, B * NQ, datt, D, hd, zbar, H * D, D, dWin[2 * D:], D, hd * D, H, transa=True)   # dWv_h = datt_h^T zbar_h
        ops.colsum(datt, out=dbin[2 * D:])
        ops.sgemm_batched(B * NQ, D, hd, datt, D, hd, Wv, D, hd * D, dzbar, H * D, D, H)                          # dzbar_h = datt_h Wv_h
        hid = ctx.hidden
        hid2 = _hidden_rows(hid, B, Tp, D) if (ctx.has_mix and hid is not None) else None
        dx16 = None
        if ctx.needs_input_grad[2]:      # fine-tuned encoder layers below: the frames carry a gradient (train_hubert.py)
            du, dck, dalpha, ds_ws, pp_ws = ops.cls_pool_bwd(rows, c, hid2, p, dzbar, U, lens_i, B, Tp, NQ, R, D, normalize=bool(m.get("normalize", False)),
                                                             drop_p=pd, seed=seed, return_ws=True)
            dx16 = ops.cls_pool_dz(pp_ws, ds_ws, dzbar, U, lens_i, B, Tp, NQ, R, D).view(B, Tp, D)[:, :m["T"]]
        else:
            du, dck, dalpha = ops.cls_pool_bwd(rows, c, hid2, p, dzbar, U, lens_i, B, Tp, NQ, R, D, normalize=bool(m.get("normalize", False)), drop_p=pd, seed=seed)
        dU = ops.colsum(du.view(-1, R * D)).view(R, D)
        ops.colsum(dck.view(-1, NQ * D), out=dcls.view(NQ * D), accumulate=True)
        dqt = torch.empty(NQ, D, device=dev, dtype=torch.float32)
        ops.sgemm_batched(hd, D, NQ, qt, D, hd, dU, H * D, D, dWin[D:2 * D], D, hd * D, H, transa=True, alpha=scale)
        ops.sgemm_batched(NQ, hd, D, dU, H * D, D, Wk, D, hd * D, dqt, D, hd, H, transb=True, alpha=scale)
        ops.sgemm(dqt, c, transa=True, out=dWin[:D])
        ops.colsum(dqt, out=dbin[:D])
        ops.sgemm(dqt, Win[:D], beta=1.0, out=dcls)
        dmix = None
        if ctx.has_mix and dalpha is not None:
            dmix = z(mixw.shape[0])
            ops.mix_softmax_bwd(mixw, dalpha, dmix)
        return None, None, dx16, None, dmix, dcls.view(1, NQ, D), dWin, dbin, dWo, dbo, dnw, dnb, dpw, dpb


class KwBatchNormTrainFn(torch.autograd.Function):
    """Train-mode Kw_BatchNorm (eachKw + parallel): y = BN_{batch stats}(x), x f32 [B,K,E]; the running statistics (buffers, (e,k) order) are
    updated in place by the forward kernel, as nn.BatchNorm1d does (kw_bn.py:122-131)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps):
        x = x.float().contiguous()
        w = _c(weight)
        y, mean, rstd = ops.kw_bn_train_fwd(x, w, _c(bias), running_mean, running_var, momentum, eps)
        ctx.save_for_backward(x, w, mean, rstd)
        ctx.want = weight.requires_grad or bias.requires_grad
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, mean, rstd = ctx.saved_tensors
        dx, dg, db = ops.kw_bn_bwd(x, dy.float().contiguous(), w, mean, rstd, want_param_grads=ctx.want)
        return dx, dg, db, None, None, None, None


_VQ_TABLES = {}


def _vq_tables(emb):
    """GEMM operands of the VQ backward for the frozen sub-word table (rebuilt if the table object or its version changes):
    [emb | emb] bf16 [V, 2E] and [(emb/|emb|)^T | same] bf16 [E, 2V]."""
    import weakref
    key = emb.data_ptr()
    hit = _VQ_TABLES.get(key)
    if hit is not None and hit[0] == (emb._version, ops.param_epoch(emb), tuple(emb.shape)) and hit[2]() is emb:
        return hit[1]
    e = emb.detach().float().contiguous()
    out = (_dup_k(e.to(BF)), _dup_k(ops.l2norm(e, clamp=True).t().contiguous().to(BF)))
    _VQ_TABLES.clear()
    _VQ_TABLES[key] = ((emb._version, ops.param_epoch(emb), tuple(emb.shape)), out, weakref.ref(emb))
    return out


class KeywordSTFn(torch.autograd.Function):
    """keywords [R,E] = subword_prob @ emb with the straight-through estimator of SimpleVectorQuantizer in train mode
    (hard one-hot forward, softmax(cos/temp) backward; my_vector_quantizer.py:133-141, kwClip.py:909-911) and the cosine similarity behind it
    (kwClip.py:889-897).  Forward = a row gather; backward = sc_sgemm -> sc_vq_st_bwd -> sc_sgemm -> sc_cosine_bwd_finish."""

    @staticmethod
    def forward(ctx, kb, cos, targets, emb, temp, mask_ids):
        """temp: a float, or the quantizer's LEARNABLE temperature parameter (`vq.temp: "learnable=..."`, my_vector_quantizer.py:33-38): then its gradient is
        returned too.  With z = cos / T and soft = softmax(z): d loss / d T = sum_v (d loss / d z_v)(-cos_v / T^2) = -(1 / T) sum_v dcos_v cos_v, and
        `rowdot` = sum_v dcos_v cos_v is what sc_vq_st_bwd already returns per row for the cosine backward."""
        ctx.save_for_backward(kb.detach().float().contiguous(), cos, emb)
        ctx.temp_is_param = torch.is_tensor(temp) and temp.requires_grad
        ctx.temp, ctx.mask_ids = float(temp), tuple(int(i) for i in mask_ids)
        return ops.gather_rows(emb, targets.reshape(-1))

    @staticmethod
    def backward(ctx, dkw):
        kb, cos, emb = ctx.saved_tensors
        E2, U2 = _vq_tables(emb)
        dprob = _mfma_f32(dkw.float().contiguous(), E2)                                       # d loss / d subword_prob = dkw @ emb^T  [R, V]
        rowdot = ops.vq_st_bwd_(cos, dprob, ctx.temp, ctx.mask_ids)                           # dprob is now d loss / d cos
        G = _mfma_f32(dprob, U2)                                                              # dcos @ (emb / |emb|)  [R, E]
        dtemp = (rowdot.sum() * (-1.0 / ctx.temp)).reshape(1) if ctx.temp_is_param else None
        return ops.cosine_bwd_finish(kb, G, rowdot), None, None, None, dtemp, None


class TextTowerTrainFn(torch.autograd.Function):
    """feat f32 [B, E] = (ln_final(text_transformer(emb + pos))[:, pos_index]) @ text_projection with the gradient w.r.t. the token
    embeddings `emb` f32 [B, L, W] (the tower itself is frozen: clip_official.py:220-264 under loss.backward()).  Forward on the bf16 MFMA
    kernels (the eval path's rounding points), saving x / qkv / x_mid / pre-activation per layer; backward: fp32 row ops, the dX products on
    the MFMA GEMM against transposed bf16 copies of the frozen weights with hi+lo split gradients (_mfma_f32)."""

    @staticmethod
    def forward(ctx, clip, emb, pos_index):
        dev = emb.device
        P = clip.packed(dev)
        B, L, W = emb.shape
        H = clip.transformer.heads
        x = (emb.detach().float() + P["txt_pos"][:L]).reshape(B * L, W).contiguous()
        saved = []
        for Ly in P["txt"]:
            n = ops.layernorm(x, *Ly["ln1"])
            qkv = ops.gemm(n, Ly["wqkv"], Ly["bqkv"])
            att = ops.attention(qkv, B, L, H, None, causal=True)
            xm = ops.gemm(att, Ly["wo"], Ly["bo"], residual=x, out_f32=True)
            n2 = ops.layernorm(xm, *Ly["ln2"])
            u = ops.gemm(n2, Ly["w1"], Ly["b1"], out_f32=True)
            h = ops.quickgelu_f32(u, out_bf16=True)
            xo = ops.gemm(h, Ly["w2"], Ly["b2"], residual=xm, out_f32=True)
            saved += [x, qkv, xm, u]
            x = xo
        rows = x.view(B, L, W)[:, pos_index].contiguous()
        n = ops.layernorm(rows, *P["ln_final"])
        feat = ops.gemm(n, P["txt_proj_t"], out_f32=True)
        ctx.clip, ctx.dims = clip, (B, L, W, H, int(pos_index))
        ctx.save_for_backward(rows, *saved)
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        rows, *saved = ctx.saved_tensors
        B, L, W, H, pos = ctx.dims
        dev = rows.device
        F = ctx.clip.packed_bwd(dev)
        dn = _mfma_f32(dfeat.float().contiguous(), F["txt_proj"])                             # [B, W] = dfeat @ text_projection^T
        drows = ops.layernorm_bwd(rows, dn, F["ln_final_g"], eps=1e-5)
        dx = torch.zeros(B * L, W, device=dev, dtype=torch.float32)
        dx.view(B, L, W)[:, pos] = drows
        for li in range(len(F["txt"]) - 1, -1, -1):
            Ly = F["txt"][li]
            x, qkv, xm, u = saved[4 * li:4 * li + 4]
            dh = _mfma_f32(dx, Ly["w2t"])                                                    # [M, 4W] = dx @ W2
            ops.quickgelu_bwd_(u, dh)
            dn2 = _mfma_f32(dh, Ly["w1t"])
            ops.layernorm_bwd(xm, dn2, Ly["g2"], eps=1e-5, dx=dx, accumulate_dx=True)         # dx is now d loss / d x_mid
            datt = _mfma_f32(dx, Ly["wot"])
            dqkv = ops.attn_small_bwd(qkv, datt, B, L, H, True)
            dn1 = _mfma_f32(dqkv, Ly["wqkvt"])
            ops.layernorm_bwd(x, dn1, Ly["g1"], eps=1e-5, dx=dx, accumulate_dx=True)
        return None, dx.view(B, L, W), None


class L2NormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.float().contiguous()
        ctx.save_for_backward(x)
        return ops.l2norm(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.l2norm_bwd(x, dy.float().contiguous())


class MaskedContrastiveFn(torch.autograd.Function):
    """loss = MaskedContrastiveLoss(feat_a, feat_b, ids); gradients for feat_a and (if given) the log-temperature parameter."""

    @staticmethod
    def forward(ctx, feat_a, feat_b, ids, log_temp, inv_t, margin, dcl, a2b, b2a):
        if log_temp is not None:
            inv_t = float(log_temp.detach().float().exp().item())
        out, da, dinv = ops.infonce_fwd_bwd(feat_a.detach().float().contiguous(), feat_b.detach().float().contiguous(), ids, inv_t, margin, dcl, a2b, b2a)
        ctx.save_for_backward(da, dinv)
        ctx.inv_t, ctx.has_temp = inv_t, log_temp is not None
        if feat_b.requires_grad:
            raise NotImplementedError("image-side gradients: the CLIP tower and its projection are frozen in every shipped config")
        return out[0].clone()

    @staticmethod
    def backward(ctx, dloss):
        da, dinv = ctx.saved_tensors
        g = dloss.float()
        # inv_t = exp(param).  Every rank evaluates the FULL global loss, so every rank holds the full d loss / d log-temperature, while
        # FusedAdam SUMS the flat gradient over ranks (right for the branch weights, whose per-rank gradients are partial sums through the
        # local rows): the temperature's share is therefore 1/ws per rank.
        ws = parallel.world()[1]
        dtemp = (dinv * (ctx.inv_t / ws) * g).reshape(()) if ctx.has_temp else None
        return da * g, None, None, dtemp, None, None, None, None, None


class PackedGatherFn(torch.autograd.Function):
    """The training-time exchange step as ONE collective: all float features of the rank are packed side by side with the bit-cast ids
    (parallel.pack_feats), all-gathered rank-major, and unpacked.  Every rank evaluates the SAME global loss, so the gradient of its local
    rows is simply its slice of d loss / d gathered (no reduce-scatter); parameter gradients are summed over ranks afterwards (FusedAdam)."""

    @staticmethod
    def forward(ctx, ids, keys, *tensors):
        rank, ws = parallel.world()
        feats = dict(zip(keys, tensors), id=ids)
        packed, pkeys, widths = parallel.pack_feats(feats)
        out = parallel.unpack_feats(parallel.all_gather_packed(packed), pkeys, widths)
        ctx.rank, ctx.B = rank, ids.shape[0]
        # only the features that carry a gradient on this rank (the audio embeddings) stay differentiable: the gathered image features of a
        # frozen tower must not look trainable to the loss
        ctx.mark_non_differentiable(out["id"], *[out[k] for k, t in zip(keys, tensors) if not t.requires_grad])
        return (out["id"],) + tuple(out[k] for k in keys)

    @staticmethod
    def backward(ctx, _did, *dys):
        lo, hi = ctx.rank * ctx.B, (ctx.rank + 1) * ctx.B
        return (None, None) + tuple(None if dy is None else dy[lo:hi] for dy in dys)


def gather_loss_feats_train(feats: dict) -> dict:
    """Training-time variant of parallel.gather_loss_feats: differentiable w.r.t. the audio features, still a single collective."""
    rank, ws = parallel.world()
    if ws == 1:
        return feats
    keys = tuple(k for k in sorted(feats) if k != "id" and torch.is_tensor(feats[k]))
    outs = PackedGatherFn.apply(feats["id"], keys, *[feats[k] for k in keys])
    res = {k: v for k, v in feats.items() if k not in keys and k != "id"}
    res["id"] = outs[0]
    res.update(dict(zip(keys, outs[1:])))
    return res


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics on ONE flat fp32 buffer: parameters, gradients and both moments live contiguously, `p.data` / `p.grad`
    are views, so a step is  [all-reduce of the flat gradient over ranks]  ->  sc_grad_norm (clip coefficient)  ->  sc_adam_step."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_grad_norm: float = 0.0):
        params = [p for p in params if p.requires_grad]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        dev = params[0].device
        n = sum(p.numel() for p in params)
        self.flat_p = torch.empty(n, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(n, device=dev, dtype=torch.float32)
        self.m = torch.zeros(n, device=dev, dtype=torch.float32)
        self.v = torch.zeros(n, device=dev, dtype=torch.float32)
        off = 0
        for p in params:
            k = p.numel()
            self.flat_p[off:off + k] = p.data.detach().float().reshape(-1)
            p.data = self.flat_p[off:off + k].view_as(p)
            p.grad = self.flat_g[off:off + k].view_as(p)
            off += k
        self._params = params
        self.max_grad_norm = max_grad_norm
        self.steps = 0
        self.last_grad_norm: Optional[torch.Tensor] = None

    def zero_grad(self, set_to_none: bool = False):
        self.flat_g.zero_()
        off = 0
        for p in self._params:          # re-attach the views if someone replaced .grad
            k = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.flat_g[off:off + k].data_ptr():
                p.grad = self.flat_g[off:off + k].view_as(p)
            off += k

    @torch.no_grad()
    def step(self, closure=None):
        assert closure is None
        rank, ws = parallel.world()
        if ws > 1:
            dist.all_reduce(self.flat_g)            # sum over ranks of the per-rank contributions to the single global loss
        g = self.param_groups[0]
        nc = ops.grad_norm(self.flat_g, self.max_grad_norm)
        self.last_grad_norm = nc
        self.steps += 1
        ops.adam_step(self.flat_p, self.flat_g, self.m, self.v, self.steps, g["lr"], g["betas"], g["eps"], g["weight_decay"], clip_coef=nc)
        # sc_adam_step wrote the parameters through raw pointers: torch's per-tensor version counters did not move, so every
        # parameter-derived cache of the eval path (bf16 casts, pooling operands) is invalidated through the global epoch instead
        ops.bump_param_epoch()

    # ---- checkpoint / resume (Lightning saves optimizer.state_dict() and restores it under --resume, base_task.py:60-61,:212)
    def state_dict(self):
        sd = super().state_dict()
        sd["fused"] = {"m": self.m.detach().clone(), "v": self.v.detach().clone(), "steps": int(self.steps),
                       "numel": [int(p.numel()) for p in self._params]}
        return sd

    def load_state_dict(self, state_dict):
        state_dict = dict(state_dict)
        fused = state_dict.pop("fused", None)
        super().load_state_dict(state_dict)
        if fused is not None:
            if list(fused["numel"]) != [int(p.numel()) for p in self._params]:
                raise ValueError("FusedAdam.load_state_dict: parameter layout differs from the checkpoint's")
            self.m.copy_(fused["m"].to(self.m.device))
            self.v.copy_(fused["v"].to(self.v.device))
            self.steps = int(fused["steps"])
        self._reattach()

    def _reattach(self):
        """p.data / p.grad back onto the flat buffers (a module.load_state_dict copies in place and keeps the views; anything that REPLACED
        p.data -- module.to(), a manual assignment -- is folded back here)."""
        off = 0
        for p in self._params:
            k = p.numel()
            view = self.flat_p[off:off + k].view_as(p)
            if p.data.data_ptr() != view.data_ptr():
                view.copy_(p.data.detach().float())
                p.data = view
            if p.grad is None or p.grad.data_ptr() != self.flat_g[off:off + k].data_ptr():
                p.grad = self.flat_g[off:off + k].view_as(p)
            off += k
        ops.bump_param_epoch()

"""Fine-tuning HuBERT transformer layers on MI355X.

Reference: `FairseqSpeechEncoder_Hubert(trainable=True, reinit_layers=[...] | unfreeze_layers=[...])` (avssl/module/speech_encoder_plus.py:416-446)
trains the listed fairseq `TransformerSentenceEncoderLayer`s and freezes everything below them in the data path's sense -- the conv feature
extractor, `post_extract_proj`, `layer_norm`, `pos_conv` (`feature_grad_mult = 0`) -- plus the unlisted layers.  So the gradient enters at the
hidden states (through `WeightedSumLayer`, weighted_sum.py:26-45) and has to travel down to the LOWEST listed layer; nothing below it needs one.

`HubertLayersTrainFn` runs layers L0..n-1 (L0 = lowest trainable layer) as one autograd node: forward on the eval path's kernels (MFMA GEMMs
with fused epilogues, flash attention, LayerNorm) keeping what the backward needs; backward in bf16 with fp32 accumulation:
  dX = dY W            sc_gemm_bf16 on transposed bf16 weight copies (residual branches added in the epilogue)
  dW = dY^T X          sc_transpose_bf16 of both operands + ONE split-K sc_gemm_bf16_batched + sc_colsum over the splits (fp32 result)
  attention            S = Q K^T and dP = dO V^T recomputed per head (batched MFMA GEMMs), sc_attn_softmax_bwd -> P, dS, then dQ = dS K,
                       dK = dS^T Q, dV = P^T dO as batched GEMMs over transposed operands (key-padding mask = the forward's klens)
  LayerNorm / GELU     sc_layernorm_bwd_bf16 (+ partial column sums -> dgamma, dbeta), sc_gelu_bwd_bf16 (fc1's pre-activation is recomputed)
Post-LN layers (HuBERT-base) and, with meta["pre_ln"], pre-LN layers on an fp32 residual stream (HuBERT-large, `unfreeze_layers` / `reinit_layers`
only: its LayerNorm-extractor front end has no backward here).  meta["drop"] = dict(hidden, attention, activation, seed) applies the
checkpoint's dropouts inside the trained layers as fairseq does in train mode (dropout1 / dropout2 / dropout3, attention probabilities): the
masks are counter-based, so the backward regenerates them from the per-site seeds instead of storing them.
"""
from typing import List, Sequence

import torch

from . import ops
from .ops import ACT_GELU

BF = torch.bfloat16
PER_LAYER = 16     # q_w q_b k_w k_b v_w v_b o_w o_b ln1_w ln1_b fc1_w fc1_b fc2_w fc2_b ln2_w ln2_b


def layer_params(lyr) -> List[torch.nn.Parameter]:
    a = lyr.self_attn
    return [a.q_proj.weight, a.q_proj.bias, a.k_proj.weight, a.k_proj.bias, a.v_proj.weight, a.v_proj.bias, a.out_proj.weight, a.out_proj.bias,
            lyr.self_attn_layer_norm.weight, lyr.self_attn_layer_norm.bias, lyr.fc1.weight, lyr.fc1.bias, lyr.fc2.weight, lyr.fc2.bias,
            lyr.final_layer_norm.weight, lyr.final_layer_norm.bias]


def _w16(t):
    return t.detach().to(BF).contiguous()


def _f32(t):
    return t.detach().float().contiguous()


def wgrad(dy: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """dW f32 [N, K] = dy^T x for bf16 dy [M, N], x [M, K]: both operands transposed (rows padded to the split size), one split-K batched MFMA
    GEMM into per-split fp32 partials, column-summed."""
    M, N = dy.shape
    K = x.shape[1]
    if N >= 256 and K >= 256:      # the split products run as ONE persistent tile list on the 256-tile kernel (gemm256 BATCH): aim at ~2 tiles per CU
        tiles = ((N + 255) // 256) * ((K + 255) // 256)
    else:
        tiles = ((N + 127) // 128) * ((K + 127) // 128)
    S = max(1, min(32, 512 // tiles, M // 2048 if M >= 4096 else 1))
    chunk = -(-M // (64 * S)) * 64
    Mp = chunk * S
    dyT = ops.transpose_bf16(dy, dy.stride(0), 0, M, N, 1, rows_padded=Mp)[0]          # [N, Mp]
    xT = ops.transpose_bf16(x, x.stride(0), 0, M, K, 1, rows_padded=Mp)[0]             # [K, Mp]
    part = torch.empty(S, N, K, device=dy.device, dtype=torch.float32)
    ops.gemm_batched(dyT, Mp, chunk, xT, chunk, S, part, K, N * K, None, N, K, chunk, S, ldw=Mp)
    if S == 1:
        return part[0]
    return ops.colsum(part.view(S, N * K)).view(N, K)


def attention_bwd(qkv: torch.Tensor, att: torch.Tensor, datt: torch.Tensor, B: int, Tp: int, H: int, klens_i32: torch.Tensor, drop=None) -> torch.Tensor:
    """qkv bf16 [>= B*Tp (+ Lp - Tp slack rows), 3*H*64] packed (q | k | v) as the forward produced it; att / datt bf16 [B*Tp, H*64] (attention output and its
    gradient) -> dqkv bf16 [B*Tp, 3*H*64].  Every product covers all (utterance, head) pairs in ONE two-level batched launch; drop = (p, seed) of a
    forward that ran attention_dropout."""
    d = H * 64
    M = B * Tp
    Lp = -(-Tp // 64) * 64
    assert qkv.shape[0] >= M + (Lp - Tp) and qkv.shape[1] == 3 * d and qkv.is_contiguous() and att.is_contiguous() and datt.is_contiguous()
    dev = qkv.device
    Z = B * H
    dqkv = torch.empty(M, 3 * d, device=dev, dtype=BF)
    q, k, v = qkv, qkv[:, d:], qkv[:, 2 * d:]
    img, rq, ro = Lp * Lp, Tp * 3 * d, Tp * d
    # S = Q K^T, dP = dO V^T, P = softmax(S / 8 + key mask) [dropped], dS = P (m dP - dO.O) / 8 in ONE kernel; the fp32 S / dP images are never written
    P, dS = ops.attn_bwd_probs(qkv, datt, att, klens_i32, B, Tp, H, drop)
    PT = ops.transpose_bf16(P, Lp, img, Lp, Lp, Z)
    dST = ops.transpose_bf16(dS, Lp, img, Lp, Lp, Z)
    del P
    kT = torch.empty(Z, 64, Lp, device=dev, dtype=BF)
    qT, doT = torch.empty_like(kT), torch.empty_like(kT)
    for h in range(H):                                                  # [Tp, 64] blocks of the packed rows -> [64, Lp] per (b, h)
        o = h * 64 * Lp
        ops.transpose_bf16(k[:, h * 64:], 3 * d, rq, Tp, 64, B, rows_padded=Lp, out=kT.view(-1)[o:], ld_out=Lp, stride_out=H * 64 * Lp)
        ops.transpose_bf16(q[:, h * 64:], 3 * d, rq, Tp, 64, B, rows_padded=Lp, out=qT.view(-1)[o:], ld_out=Lp, stride_out=H * 64 * Lp)
        ops.transpose_bf16(datt[:, h * 64:], d, ro, Tp, 64, B, rows_padded=Lp, out=doT.view(-1)[o:], ld_out=Lp, stride_out=H * 64 * Lp)
    t = 64 * Lp
    ops.gemm_batched2(dS, Lp, H * img, img, kT, Lp, H * t, t, dqkv, 3 * d, rq, 64, Tp, 64, Lp, B, H)                      # dQ = dS K
    ops.gemm_batched2(dST, Lp, H * img, img, qT, Lp, H * t, t, dqkv[:, d:], 3 * d, rq, 64, Tp, 64, Lp, B, H)              # dK = dS^T Q
    ops.gemm_batched2(PT, Lp, H * img, img, doT, Lp, H * t, t, dqkv[:, 2 * d:], 3 * d, rq, 64, Tp, 64, Lp, B, H)          # dV = P^T dO
    return dqkv


class HubertLayersTrainFn(torch.autograd.Function):
    """hidden bf16 [n, M, d] = outputs of post-LN layers L0 .. L0+n-1 applied to h_in.
    args: meta (B, Tp, H, eps, train (list of bool per layer: compute parameter gradients)), h_in bf16 [M, d], valid_i32 [B], then 16 tensors per layer."""

    @staticmethod
    def forward(ctx, meta, h_in, valid_i32, *params):
        B, Tp, H, eps = meta["B"], meta["Tp"], meta["H"], meta["eps"]
        n = len(params) // PER_LAYER
        M, d = h_in.shape
        assert M == B * Tp and d == H * 64
        dev = h_in.device
        Lp = -(-Tp // 64) * 64
        pre_ln = bool(meta.get("pre_ln", False))
        if pre_ln:
            return HubertLayersTrainFn._forward_pre_ln(ctx, meta, h_in, valid_i32, params)
        hidden = torch.empty(n, M, d, device=dev, dtype=BF)
        saved = []
        drop = meta.get("drop")
        seeds = []
        if drop is not None:
            s0 = int(drop["seed"]) & 0x7fffffff
            for _ in range(4 * n):
                s0 = (s0 * 1103515245 + 12345) & 0x7fffffff
                seeds.append(s0)
        h = h_in.detach()
        for li in range(n):
            qw, qb, kw, kb, vw, vb, ow, ob, g1, b1n, w1, b1, w2, b2, g2, b2n = params[li * PER_LAYER:(li + 1) * PER_LAYER]
            wqkv, bqkv = _w16(torch.cat([qw, kw, vw], 0)), _f32(torch.cat([qb, kb, vb], 0))
            qkv = torch.empty(M + (Lp - Tp), 3 * d, device=dev, dtype=BF)
            qkv[M:].zero_()          # slack rows: the backward's S / dP products read Lp keys per utterance
            ops.gemm(h, wqkv, bqkv, out=qkv[:M])
            if drop is None:
                att = ops.attention(qkv[:M], B, Tp, H, valid_i32)
                y1 = ops.gemm(att, _w16(ow), _f32(ob), residual=h)
                x1 = ops.layernorm(y1, _f32(g1), _f32(b1n), eps)
                hm = ops.gemm(x1, _w16(w1), _f32(b1), ACT_GELU)
                y2 = ops.gemm(hm, _w16(w2), _f32(b2), residual=x1)
            else:       # x = LN(x + dropout1(attn(x)));  x = LN(x + dropout3(fc2(dropout2(gelu(fc1 x)))))
                sa, s1, s2, s3 = seeds[4 * li:4 * li + 4]
                att = ops.attention_dropout(qkv[:M], B, Tp, H, valid_i32, drop["attention"], sa)
                y1 = ops.gemm(att, _w16(ow), _f32(ob))
                ops.dropout_bf16(y1, drop["hidden"], s1, residual=h, out=y1)
                x1 = ops.layernorm(y1, _f32(g1), _f32(b1n), eps)
                hm = ops.gemm(x1, _w16(w1), _f32(b1), ACT_GELU)
                if drop["activation"] > 0:
                    ops.dropout_bf16(hm, drop["activation"], s2, out=hm)
                y2 = ops.gemm(hm, _w16(w2), _f32(b2))
                ops.dropout_bf16(y2, drop["hidden"], s3, residual=x1, out=y2)
            ops.layernorm(y2, _f32(g2), _f32(b2n), eps, out=hidden[li])
            saved += [h, qkv, att, y1, x1, hm, y2]
            h = hidden[li]
        ctx.meta = dict(meta, n=n, seeds=seeds)
        ctx.valid = valid_i32
        ctx.save_for_backward(*saved, *[p.detach() for p in params])
        return hidden

    @staticmethod
    def _forward_pre_ln(ctx, meta, h_in, valid_i32, params):
        """Pre-LN layers ([3P fairseq] layer_norm_first, HuBERT-large): x += attn(LN1 x); x += fc2(gelu(fc1(LN2 x))) on an fp32 residual stream.
        h_in / hidden are f32; the saved copies of the stream (LayerNorm inputs of the backward) are bf16.  The large checkpoint's dropouts are 0."""
        B, Tp, H, eps = meta["B"], meta["Tp"], meta["H"], meta["eps"]
        assert meta.get("drop") is None or not any(v > 0 for k, v in meta["drop"].items() if k != "seed"), "dropout inside pre-LN layers is not built"
        n = len(params) // PER_LAYER
        M, d = h_in.shape
        dev = h_in.device
        Lp = -(-Tp // 64) * 64
        hidden = torch.empty(n, M, d, device=dev, dtype=torch.float32)
        saved = []
        h = h_in.detach().float().contiguous()
        for li in range(n):
            qw, qb, kw, kb, vw, vb, ow, ob, g1, b1n, w1, b1, w2, b2, g2, b2n = params[li * PER_LAYER:(li + 1) * PER_LAYER]
            wqkv, bqkv = _w16(torch.cat([qw, kw, vw], 0)), _f32(torch.cat([qb, kb, vb], 0))
            t1 = ops.layernorm(h, _f32(g1), _f32(b1n), eps)                                   # bf16
            qkv = torch.empty(M + (Lp - Tp), 3 * d, device=dev, dtype=BF)
            qkv[M:].zero_()
            ops.gemm(t1, wqkv, bqkv, out=qkv[:M])
            att = ops.attention(qkv[:M], B, Tp, H, valid_i32)
            xmid = ops.gemm(att, _w16(ow), _f32(ob), residual=h, out_f32=True)
            t2 = ops.layernorm(xmid, _f32(g2), _f32(b2n), eps)
            hm = ops.gemm(t2, _w16(w1), _f32(b1), ACT_GELU)
            ops.gemm(hm, _w16(w2), _f32(b2), residual=xmid, out=hidden[li], out_f32=True)
            saved += [h.to(BF), qkv, att, xmid.to(BF), t1, hm, t2]
            h = hidden[li]
        ctx.meta = dict(meta, n=n, seeds=[])
        ctx.valid = valid_i32
        ctx.save_for_backward(*saved, *[p.detach() for p in params])
        return hidden

    @staticmethod
    def _backward_pre_ln(ctx, dhidden):
        m = ctx.meta
        B, Tp, H, eps, n, train = m["B"], m["Tp"], m["H"], m["eps"], m["n"], m["train"]
        tensors = ctx.saved_tensors
        acts, params = tensors[:7 * n], tensors[7 * n:]
        dhidden = dhidden.to(BF).contiguous()
        grads = [None] * len(params)
        g = dhidden[n - 1].clone()                                   # gradient of the residual stream after the top layer (bf16)
        for li in range(n - 1, -1, -1):
            h16, qkv, att, xmid16, t1, hm, t2 = acts[7 * li:7 * li + 7]
            qw, qb, kw, kb, vw, vb, ow, ob, g1, b1n, w1, b1, w2, b2, g2, b2n = params[li * PER_LAYER:(li + 1) * PER_LAYER]
            want = bool(train[li])
            M, d = h16.shape
            # out = xmid + fc2(gelu(fc1(t2))),  t2 = LN2(xmid)
            dhm = ops.gemm(g, _w16(w2.t()))
            u = ops.gemm(t2, _w16(w1), _f32(b1))
            du = ops.gelu_bwd_bf16(u, dhm)
            del u, dhm
            dt2 = ops.gemm(du, _w16(w1.t()))
            dxm, dg2, db2n = ops.layernorm_bwd_bf16(xmid16, dt2, _f32(g2), eps, want)
            ops.axpy_bf16(dxm, g, 1.0)                                 # + the residual path
            # xmid = h + out_proj(attn(qkv(t1))),  t1 = LN1(h)
            datt = ops.gemm(dxm, _w16(ow.t()))
            dqkv = attention_bwd(qkv, att, datt, B, Tp, H, ctx.valid)
            wqkv = torch.cat([qw, kw, vw], 0)
            dt1 = ops.gemm(dqkv, _w16(wqkv.t()))
            dh, dg1, db1n = ops.layernorm_bwd_bf16(h16, dt1, _f32(g1), eps, want)
            ops.axpy_bf16(dh, dxm, 1.0)
            if want:
                dwqkv = wgrad(dqkv, t1)
                dbqkv = ops.colsum_bf16(dqkv)
                base = li * PER_LAYER
                grads[base + 0], grads[base + 2], grads[base + 4] = dwqkv[:d], dwqkv[d:2 * d], dwqkv[2 * d:]
                grads[base + 1], grads[base + 3], grads[base + 5] = dbqkv[:d], dbqkv[d:2 * d], dbqkv[2 * d:]
                grads[base + 6], grads[base + 7] = wgrad(dxm, att), ops.colsum_bf16(dxm)
                grads[base + 8], grads[base + 9] = dg1, db1n
                grads[base + 10], grads[base + 11] = wgrad(du, t2), ops.colsum_bf16(du)
                grads[base + 12], grads[base + 13] = wgrad(g, hm), ops.colsum_bf16(g)
                grads[base + 14], grads[base + 15] = dg2, db2n
            g = dh
            if li > 0:
                ops.axpy_bf16(g, dhidden[li - 1], 1.0)
        return (None, g.float() if ctx.needs_input_grad[1] else None, None, *grads)

    @staticmethod
    def backward(ctx, dhidden):
        if ctx.meta.get("pre_ln"):
            return HubertLayersTrainFn._backward_pre_ln(ctx, dhidden)
        m = ctx.meta
        B, Tp, H, eps, n, train = m["B"], m["Tp"], m["H"], m["eps"], m["n"], m["train"]
        tensors = ctx.saved_tensors
        acts, params = tensors[:7 * n], tensors[7 * n:]
        dhidden = dhidden.to(BF).contiguous()
        grads = [None] * len(params)
        g = dhidden[n - 1].clone()                                   # d loss / d (output of the top layer)
        for li in range(n - 1, -1, -1):
            h, qkv, att, y1, x1, hm, y2 = acts[7 * li:7 * li + 7]
            qw, qb, kw, kb, vw, vb, ow, ob, g1, b1n, w1, b1, w2, b2, g2, b2n = params[li * PER_LAYER:(li + 1) * PER_LAYER]
            want = bool(train[li])
            M, d = h.shape
            drop = m.get("drop")
            sa, s1, s2, s3 = m["seeds"][4 * li:4 * li + 4] if drop is not None else (0, 0, 0, 0)
            # x2 = LN2(y2)
            dy2r, dg2, db2n = ops.layernorm_bwd_bf16(y2, g, _f32(g2), eps, want)
            # y2 = dropout3(hm W2^T + b2) + x1: the residual branch takes dy2r as it is, the fc2 branch the masked gradient
            dy2 = dy2r if drop is None else ops.dropout_bf16(dy2r, drop["hidden"], s3)
            dhm = ops.gemm(dy2, _w16(w2.t()))                                  # [M, ffn] = dy2 W2
            if drop is not None and drop["activation"] > 0:
                ops.dropout_bf16(dhm, drop["activation"], s2, out=dhm)
            u = ops.gemm(x1, _w16(w1), _f32(b1))                               # fc1's pre-activation, recomputed (not kept by the forward)
            du = ops.gelu_bwd_bf16(u, dhm)
            del u, dhm
            # u = x1 W1^T + b1 ; x1 also feeds the residual of fc2
            dx1 = ops.gemm(du, _w16(w1.t()), residual=dy2r)                    # [M, d] = du W1 + dy2 (unmasked: the residual path)
            # x1 = LN1(y1)
            dy1r, dg1, db1n = ops.layernorm_bwd_bf16(y1, dx1, _f32(g1), eps, want)
            # y1 = dropout1(att Wo^T + bo) + h
            dy1 = dy1r if drop is None else ops.dropout_bf16(dy1r, drop["hidden"], s1)
            datt = ops.gemm(dy1, _w16(ow.t()))
            dqkv = attention_bwd(qkv, att, datt, B, Tp, H, ctx.valid, None if drop is None or drop["attention"] <= 0 else (drop["attention"], sa))
            wqkv = torch.cat([qw, kw, vw], 0)
            dh = ops.gemm(dqkv, _w16(wqkv.t()), residual=dy1r)                 # [M, d] = dqkv Wqkv + dy1 (unmasked: the residual path)
            if want:
                dwqkv = wgrad(dqkv, h)
                dbqkv = ops.colsum_bf16(dqkv)
                base = li * PER_LAYER
                grads[base + 0], grads[base + 2], grads[base + 4] = dwqkv[:d], dwqkv[d:2 * d], dwqkv[2 * d:]
                grads[base + 1], grads[base + 3], grads[base + 5] = dbqkv[:d], dbqkv[d:2 * d], dbqkv[2 * d:]
                grads[base + 6], grads[base + 7] = wgrad(dy1, att), ops.colsum_bf16(dy1)
                grads[base + 8], grads[base + 9] = dg1, db1n
                grads[base + 10], grads[base + 11] = wgrad(du, x1), ops.colsum_bf16(du)
                grads[base + 12], grads[base + 13] = wgrad(dy2, hm), ops.colsum_bf16(dy2)
                grads[base + 14], grads[base + 15] = dg2, db2n
            g = dh
            if li > 0:
                ops.axpy_bf16(g, dhidden[li - 1], 1.0)                         # + the direct gradient of hidden[li - 1] (its share of the layer mix)
        return (None, g if ctx.needs_input_grad[1] else None, None, *grads)


class WeightedSumTrainFn(torch.autograd.Function):
    """mixed bf16 [M, D] = sum_l softmax(w)_l hidden_l with the gradient w.r.t. the HIDDEN STATES (dhidden_l = softmax(w)_l dmixed); the gradient of
    the mix weights keeps coming out of the pooling head's backward (sc_cls_pool_bwd's dalpha), so it is not produced here a second time."""

    @staticmethod
    def forward(ctx, hidden, weights, normalize):
        ctx.normalize = bool(normalize)
        ctx.save_for_backward(weights.detach(), hidden.detach().to(BF) if normalize else weights.detach())
        n, M, D = hidden.shape
        return ops.weighted_sum(hidden.detach().contiguous(), weights.detach().float(), bool(normalize))

    @staticmethod
    def backward(ctx, dmixed):
        w, h16 = ctx.saved_tensors
        sm = torch.softmax(w.float(), 0).tolist()
        dm = dmixed.to(BF).contiguous()
        out = torch.zeros(len(sm), *dm.shape, device=dm.device, dtype=BF)
        for l, a in enumerate(sm):
            ops.axpy_bf16(out[l], dm, a)
        if ctx.normalize:      # F.layer_norm(hidden_l) without affine in front of the mix (weighted_sum.py:41-42): its backward per state
            ones = torch.ones(dm.shape[-1], device=dm.device, dtype=torch.float32)
            for l in range(len(sm)):
                out[l] = ops.layernorm_bwd_bf16(h16[l].contiguous(), out[l].contiguous(), ones, 1e-5, False)[0]
        return out, None, None

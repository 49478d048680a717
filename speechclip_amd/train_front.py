"""Training the HuBERT front end on MI355X: `audio_encoder.trainable: true` WITHOUT `reinit_layers` / `unfreeze_layers`
(avssl/module/speech_encoder_plus.py:399-401 -- `freeze_model` is simply not called, so the conv feature extractor, `post_extract_proj`,
`layer_norm`, the positional conv and every transformer layer train; [3P fairseq] `HubertModel.forward_features` multiplies the gradient that
enters the feature extractor by `feature_grad_mult`, 0.1 in the released base checkpoint).

`HubertFrontTrainFn` = wave -> hidden state 0 (the input of transformer layer 0) as ONE autograd node, post-LN / GroupNorm models (HuBERT-base):
  forward   the eval path's kernels (sc_conv0_fwd, conv-as-GEMM with fused GELU, LayerNorm, projection, sc_posconv_conv), keeping the layer outputs,
            plus sc_posconv_finish_train (pre-activation and LayerNorm input of the positional-conv tail)
  backward  encoder LayerNorm      sc_layernorm_bwd_bf16
            positional conv        dX: the SAME grouped conv on the time-reversed gradient with in/out channels swapped (sc_posconv_conv +
                                   sc_posconv_dgrad_finish); dW: per group, split-K sc_gemm_bf16_batched over the transposed sliding-window
                                   view (sc_posconv_pack + sc_transpose_bf16); weight-norm (g, v) from dW in fp32 on the [d, d/G, Kw] tensors
            projection, feature LN sc_gemm_bf16 on the transposed weight, train_hubert.wgrad, sc_layernorm_bwd_bf16
            conv layers 6..1       pre-activation recomputed (sc_gemm_bf16 without the GELU), sc_gelu_bwd_bf16, dW = split-K TN GEMM over the
                                   overlapping-row view, dX = one GEMM per group of taps written straight into the channels-last input gradient
                                   (taps 0..s-1 tile it exactly; tap s.. accumulate through the residual epilogue, in place)
            conv layer 0           sc_conv0_bwd (GroupNorm + GELU + conv from the wave)
meta["drop"] = dict(features, hidden, seed) applies the two dropouts fairseq has on this stretch in train mode (dropout_input on the projected
features, F.dropout on hidden state 0) with counter-based masks the backward regenerates.
"""
import torch

from . import ops
from .ops import ACT_GELU, ACT_NONE
from .train_hubert import wgrad

BF = torch.bfloat16
N_FRONT = 18   # conv0 w, gn w, gn b, conv1..6 w, feat-LN w b, proj w b, pos g v bias, enc-LN w b


def front_params(enc) -> list:
    """The trainable tensors of the front end of module.hubert.HubertModel, in HubertFrontTrainFn's order (base / GroupNorm extractor)."""
    convs = enc.feature_extractor.conv_layers
    assert enc.cfg.extractor_mode == "default" and not enc.cfg.conv_bias and not enc.cfg.layer_norm_first and len(convs) == 7
    gn = getattr(convs[0], "2")
    pc = getattr(enc.encoder.pos_conv, "0")
    return ([getattr(convs[0], "0").weight, gn.weight, gn.bias] + [getattr(convs[i], "0").weight for i in range(1, 7)] +
            [enc.layer_norm.weight, enc.layer_norm.bias, enc.post_extract_proj.weight, enc.post_extract_proj.bias,
             pc.weight_g, pc.weight_v, pc.bias, enc.encoder.layer_norm.weight, enc.encoder.layer_norm.bias])


def _f32(t):
    return t.detach().float().contiguous()


def _conv_w16(w):
    """[out, in, k] -> bf16 [out, k*in] (K index = tap*C + c_in), the conv-as-GEMM operand."""
    return w.detach().permute(0, 2, 1).reshape(w.shape[0], -1).to(BF).contiguous()


def _fold_weight_norm(g, v):
    v = v.detach().float()
    n = v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()
    return g.detach().float() * v / n, n


def _pos_operands(wfold, G, Kw):
    """folded weight f32 [d, d/G, Kw] -> (forward operand, adjoint operand) bf16 [G, cg, Kw*cg], K index = tap*cg + c_in."""
    d, cg, _ = wfold.shape
    w4 = wfold.view(G, cg, cg, Kw)                                   # [g, out, in, tap]
    fwd = w4.permute(0, 1, 3, 2).reshape(G, cg, Kw * cg).to(BF).contiguous()
    adj = w4.permute(0, 2, 3, 1).reshape(G, cg, Kw * cg).to(BF).contiguous()      # roles of in / out swapped
    return fwd, adj


def posconv_wgrad(du, xp, valid_i32, B, Tp, D, G, Kw):
    """dW of the grouped positional conv in the folded weight's layout: f32 [D, D/G, Kw] = sum_{b,t} du[b,t,out] * mask(xp)[b, t + tap - Kw/2, in]."""
    cg = D // G
    dev = du.device
    Tq = -(-Tp // 64) * 64
    Ktot = B * Tq
    S = max(s for s in (16, 8, 4, 2, 1) if B % s == 0)
    chunk = Ktot // S
    xg = ops.posconv_pack(xp, valid_i32, B, Tp, D, G, Kw)              # [B, G, Tp + Kw, cg]
    xvT = torch.empty(Kw * cg, Ktot, device=dev, dtype=BF)
    duT = torch.empty(cg, Ktot, device=dev, dtype=BF)
    part = torch.empty(S, Kw * cg, cg, device=dev, dtype=torch.float32)
    out = torch.empty(D, cg, Kw, device=dev, dtype=torch.float32)
    rows = Tp + Kw
    for g in range(G):
        ops.transpose_bf16(xg[g * rows * cg:], cg, G * rows * cg, Tp, Kw * cg, B, rows_padded=Tq, out=xvT, ld_out=Ktot, stride_out=Tq)
        ops.transpose_bf16(du[:, g * cg:], D, Tp * D, Tp, cg, B, rows_padded=Tq, out=duT, ld_out=Ktot, stride_out=Tq)
        ops.gemm_batched(xvT, Ktot, chunk, duT, chunk, S, part, cg, Kw * cg * cg, None, Kw * cg, cg, chunk, S, ldw=Ktot)
        tot = part[0] if S == 1 else ops.colsum(part.view(S, Kw * cg * cg)).view(Kw * cg, cg)      # [(tap, in), out]
        out[g * cg:(g + 1) * cg] = tot.view(Kw, cg, cg).permute(2, 1, 0)
    return out


def _rows_with_slack(n_rows, cols, dev):
    """[n_rows + 8, cols] bf16 whose 8 slack rows are zero (the conv-as-GEMM views over-read / the overlapping tap accumulates into them); the body is
    written in full by the producing GEMM, so it is not filled first."""
    t = torch.empty(n_rows + 8, cols, device=dev, dtype=BF)
    t[n_rows:].zero_()
    return t


def conv_layer_backward(xin, w, du_i, B, rows_out, dim, k, s, C):
    """Conv-as-GEMM layer, gradient du_i bf16 [B*rows_out, dim] of its (pre-activation) output -> (dW f32 [dim, C, k], dX bf16 [B*rows_in + 8, C])."""
    Mi = B * rows_out
    rows_in = rows_out * s
    w16 = _conv_w16(w)
    xview = torch.as_strided(xin, (Mi, k * C), (s * C, 1))
    dW = wgrad(du_i, xview).view(dim, k, C).permute(0, 2, 1).contiguous()
    wT = w16.t().contiguous()                                                    # [k*C, dim]
    dx = _rows_with_slack(B * rows_in, C, xin.device)
    ops.gemm(du_i, wT[:s * C], out=dx[:s * Mi].view(Mi, s * C))                  # taps 0 .. s-1 tile the input rows exactly
    for j in range(s, k):                                                        # overlapping taps: accumulate in place
        tgt = torch.as_strided(dx, (Mi, C), (s * C, 1), storage_offset=j * C)
        ops.gemm(du_i, wT[j * C:(j + 1) * C], residual=tgt, out=tgt)
    return dW, dx


def posconv_tail_backward(ds, u, xp, valid, pg, pv, B, Tp, d, G, Kw):
    """s = mask(xp) + gelu(u), u = grouped_conv(mask(xp)) + bias, weight-normalised weight (g, v): gradient ds of s ->
    (dxp bf16 [B*Tp, d], dg, dv, dbias)."""
    dev = ds.device
    du = ops.gelu_bwd_bf16(u, ds)
    dbias = ops.colsum_bf16(du)
    wfold, norm = _fold_weight_norm(pg, pv)
    _, wg_adj = _pos_operands(wfold, G, Kw)
    full = ops.dev_ints([Tp] * B, torch.int32, dev)
    convT = ops.posconv_conv(ops.reverse_rows_bf16(du, B, Tp, d), full, wg_adj, B, Tp, d, G, Kw)
    dxp = ops.posconv_dgrad_finish(convT, ds, valid, B, Tp, d, G)
    dwf = posconv_wgrad(du, xp, valid, B, Tp, d, G, Kw)              # gradient of the FOLDED weight
    v = pv.detach().float()
    dot = (dwf * v).sum(dim=(0, 1), keepdim=True)                     # weight-norm: w = g v / |v|  (norm over dims 0, 1 per tap)
    gf = pg.detach().float()
    return dxp, (dot / norm).to(pg.dtype), (gf / norm * dwf - gf * dot / norm.pow(3) * v).to(pv.dtype), dbias


class HubertFrontTrainFn(torch.autograd.Function):
    """h0 bf16 [B*Tp, d] = LN(mask(x) + gelu(pos_conv(mask(x)))) with x = proj(LN(conv stack(wav))).
    args: meta (conv_layers, T0, P0, Tp, d, G, Kw, grad_mult, train: compute parameter gradients), wav f32 [B, L], valid_i32 [B], N_FRONT tensors."""

    @staticmethod
    def forward(ctx, meta, wav, valid_i32, *params):
        assert len(params) == N_FRONT
        c0w, gnw, gnb = params[:3]
        cws = params[3:9]
        flw, flb, pw, pb, pg, pv, pbias, elw, elb = params[9:]
        cl, T0, P0, Tp, d, G, Kw = meta["conv_layers"], meta["T0"], meta["P0"], meta["Tp"], meta["d"], meta["G"], meta["Kw"]
        B = wav.shape[0]
        dev = wav.device
        C = cl[0][0]
        x = ops.conv0(wav, _f32(c0w).reshape(C, -1), T0, P0, gn_gamma=_f32(gnw), gn_beta=_f32(gnb))
        acts = [x]
        rows = P0
        for (dim, k, s), w in zip(cl[1:], cws):
            rows //= s
            y = _rows_with_slack(B * rows, dim, dev)
            ops.gemm(x, _conv_w16(w), None, ACT_GELU, out=y[:B * rows], M=B * rows, K=k * C, lda=s * C)
            acts.append(y)
            x, C = y, dim
        assert rows == Tp
        M = B * Tp
        feats = ops.layernorm(x[:M], _f32(flw), _f32(flb))
        xp = ops.gemm(feats, pw.detach().to(BF).contiguous(), _f32(pb))
        drop = meta.get("drop")          # dict(features, hidden, seed): dropout_input on the projected features, F.dropout on hidden state 0
        if drop is not None and drop["features"] > 0:
            ops.dropout_bf16(xp, drop["features"], drop["seed"] ^ 0x2545F491, out=xp)
        wfold, _ = _fold_weight_norm(pg, pv)
        wg, _ = _pos_operands(wfold, G, Kw)
        conv = ops.posconv_conv(xp, valid_i32, wg, B, Tp, d, G, Kw)
        u, s_ = ops.posconv_finish_train(xp, valid_i32, conv, _f32(pbias), B, Tp, d, G)
        h0 = ops.layernorm(s_, _f32(elw), _f32(elb), 1e-5)
        if drop is not None and drop["hidden"] > 0:
            h0 = ops.dropout_bf16(h0, drop["hidden"], drop["seed"] ^ 0x61C88647)
        ctx.meta = meta
        ctx.valid = valid_i32
        ctx.save_for_backward(wav, *acts, feats, xp, u, s_, *[p.detach() for p in params])
        return h0

    @staticmethod
    def backward(ctx, dh0):
        meta = ctx.meta
        cl, T0, P0, Tp, d, G, Kw = meta["conv_layers"], meta["T0"], meta["P0"], meta["Tp"], meta["d"], meta["G"], meta["Kw"]
        t = ctx.saved_tensors
        wav, acts, (feats, xp, u, s_), params = t[0], t[1:8], t[8:12], t[12:]
        c0w, gnw, gnb = params[:3]
        cws = params[3:9]
        flw, flb, pw, pb, pg, pv, pbias, elw, elb = params[9:]
        B = wav.shape[0]
        M = B * Tp
        dev = wav.device
        valid = ctx.valid
        grads = [None] * N_FRONT
        drop = meta.get("drop")
        dh0 = dh0.to(BF).contiguous()
        if drop is not None and drop["hidden"] > 0:
            dh0 = ops.dropout_bf16(dh0, drop["hidden"], drop["seed"] ^ 0x61C88647)
        # ---- h0 = [dropout] LN(s)
        ds, grads[16], grads[17] = ops.layernorm_bwd_bf16(s_, dh0, _f32(elw), 1e-5)
        # ---- s = mask(xp) + gelu(u),  u = conv(mask(xp)) + bias
        dxp, grads[13], grads[14], grads[15] = posconv_tail_backward(ds, u, xp, valid, pg, pv, B, Tp, d, G, Kw)
        del ds
        if drop is not None and drop["features"] > 0:
            ops.dropout_bf16(dxp, drop["features"], drop["seed"] ^ 0x2545F491, out=dxp)      # saved xp is the dropped tensor; its gradient takes the same mask
        # ---- xp = [dropout] (feats W^T + b) ; feats = LN(x6)
        dfeats = ops.gemm(dxp, pw.detach().t().to(BF).contiguous())
        grads[11], grads[12] = wgrad(dxp, feats), ops.colsum_bf16(dxp)
        x6 = acts[6][:M]
        g, grads[9], grads[10] = ops.layernorm_bwd_bf16(x6, dfeats, _f32(flw), 1e-5)
        mult = float(meta["grad_mult"])
        if mult != 1.0:      # [3P fairseq] GradMultiply on the feature extractor's output
            g = ops.axpy_bf16(torch.zeros_like(g), g, mult)
        # ---- conv layers 6 .. 1
        rows_out = Tp
        for i in range(6, 0, -1):
            dim, k, s = cl[i]
            C = cl[i - 1][0]
            xin = acts[i - 1]
            Mi = B * rows_out
            rows_in = rows_out * s
            w16 = _conv_w16(cws[i - 1])
            upre = ops.gemm(xin, w16, None, ACT_NONE, M=Mi, K=k * C, lda=s * C)        # pre-activation, recomputed
            du_i = ops.gelu_bwd_bf16(upre, g)
            del upre
            grads[3 + i - 1], dx = conv_layer_backward(xin, cws[i - 1], du_i, B, rows_out, dim, k, s, C)
            g = dx[:B * rows_in]
            rows_out = rows_in
            del du_i
        assert rows_out == P0
        # ---- conv layer 0 from the wave
        C0 = cl[0][0]
        dw0, dgn, dbn = ops.conv0_bwd(wav, _f32(c0w).reshape(C0, -1), _f32(gnw), _f32(gnb), g.contiguous(), T0, P0)
        grads[0], grads[1], grads[2] = dw0.view_as(c0w), dgn, dbn
        return (None, None, None, *grads)


N_FRONT_LN = 35   # 7 x (conv w, conv b, ln w, ln b), feat-LN w b, proj w b, pos g v bias


def front_params_ln(enc) -> list:
    """Front-end tensors of a LayerNorm-extractor / pre-LN model (HuBERT-large) in HubertFrontLNTrainFn's order.  `encoder.layer_norm` is not among
    them: with layer_norm_first the reference applies it to the encoder's final `x` only, which the hidden states never see (speech_encoder_plus.py:101)."""
    convs = enc.feature_extractor.conv_layers
    assert enc.cfg.extractor_mode == "layer_norm" and enc.cfg.conv_bias and enc.cfg.layer_norm_first and len(convs) == 7
    out = []
    for blk in convs:
        c, ln = getattr(blk, "0"), getattr(getattr(blk, "2"), "1")
        out += [c.weight, c.bias, ln.weight, ln.bias]
    pc = getattr(enc.encoder.pos_conv, "0")
    return out + [enc.layer_norm.weight, enc.layer_norm.bias, enc.post_extract_proj.weight, enc.post_extract_proj.bias, pc.weight_g, pc.weight_v, pc.bias]


class HubertFrontLNTrainFn(torch.autograd.Function):
    """The same node for the LayerNorm extractor + pre-LN encoder (HuBERT-large): every conv layer is conv + bias -> LayerNorm(C) -> GELU, the wave is
    layer-normalised per utterance first (no parameters), and hidden state 0 = mask(x) + gelu(pos_conv(mask(x))) WITHOUT a LayerNorm, in fp32.
    The large checkpoint has no dropouts and feature_grad_mult = 1."""

    @staticmethod
    def forward(ctx, meta, wav, lens_i32, valid_i32, *params):
        assert len(params) == N_FRONT_LN
        cl, T0, P0, Tp, d, G, Kw = meta["conv_layers"], meta["T0"], meta["P0"], meta["Tp"], meta["d"], meta["G"], meta["Kw"]
        assert meta.get("drop") is None, "the LayerNorm-extractor model has no dropouts"
        B = wav.shape[0]
        dev = wav.device
        if meta["normalize"]:
            wav = ops.wave_layernorm(wav.contiguous(), lens_i32)
        C = cl[0][0]
        w0, b0, g0, be0 = params[:4]
        u = ops.conv0(wav, _f32(w0).reshape(C, -1), T0, P0, bias=_f32(b0))                  # conv + bias; rows >= T0 are zeros
        pre, acts = [u], []
        x = torch.zeros_like(u)
        ops.layernorm(u[:B * P0], _f32(g0), _f32(be0), gelu=True, out=x[:B * P0])
        acts.append(x)
        rows = P0
        for li, (dim, k, s) in enumerate(cl[1:], start=1):
            w, b, g, be = params[4 * li:4 * li + 4]
            rows //= s
            u = _rows_with_slack(B * rows, dim, dev)
            ops.gemm(x, _conv_w16(w), _f32(b), ACT_NONE, out=u[:B * rows], M=B * rows, K=k * C, lda=s * C)
            y = _rows_with_slack(B * rows, dim, dev)
            ops.layernorm(u[:B * rows], _f32(g), _f32(be), gelu=True, out=y[:B * rows])
            pre.append(u)
            acts.append(y)
            x, C = y, dim
        assert rows == Tp
        M = B * Tp
        flw, flb, pw, pb, pg, pv, pbias = params[28:]
        feats = ops.layernorm(x[:M], _f32(flw), _f32(flb))
        xp = ops.gemm(feats, pw.detach().to(BF).contiguous(), _f32(pb))
        wfold, _ = _fold_weight_norm(pg, pv)
        wg, _ = _pos_operands(wfold, G, Kw)
        conv = ops.posconv_conv(xp, valid_i32, wg, B, Tp, d, G, Kw)
        upos, s_ = ops.posconv_finish_train(xp, valid_i32, conv, _f32(pbias), B, Tp, d, G)
        ctx.meta = meta
        ctx.valid = valid_i32
        ctx.save_for_backward(wav, *pre, *acts, feats, xp, upos, *[p.detach() for p in params])
        return s_.float()                                                                     # hidden state 0 of a pre-LN model lives on the fp32 stream

    @staticmethod
    def backward(ctx, dh0):
        meta = ctx.meta
        cl, T0, P0, Tp, d, G, Kw = meta["conv_layers"], meta["T0"], meta["P0"], meta["Tp"], meta["d"], meta["G"], meta["Kw"]
        t = ctx.saved_tensors
        wav, pre, acts, (feats, xp, upos), params = t[0], t[1:8], t[8:15], t[15:18], t[18:]
        flw, flb, pw, pb, pg, pv, pbias = params[28:]
        B = wav.shape[0]
        M = B * Tp
        grads = [None] * N_FRONT_LN
        ds = dh0.to(BF).contiguous()
        dxp, grads[32], grads[33], grads[34] = posconv_tail_backward(ds, upos, xp, ctx.valid, pg, pv, B, Tp, d, G, Kw)
        dfeats = ops.gemm(dxp, pw.detach().t().to(BF).contiguous())
        grads[30], grads[31] = wgrad(dxp, feats), ops.colsum_bf16(dxp)
        g, grads[28], grads[29] = ops.layernorm_bwd_bf16(acts[6][:M], dfeats, _f32(flw), 1e-5)
        mult = float(meta["grad_mult"])
        if mult != 1.0:
            g = ops.axpy_bf16(torch.zeros_like(g), g, mult)
        rows_out = Tp
        for li in range(6, -1, -1):
            dim, k, s = cl[li]
            w, b, gam, bet = params[4 * li:4 * li + 4]
            rows = rows_out
            u = pre[li][:B * rows]
            z = ops.layernorm(u, _f32(gam), _f32(bet))                                        # the GELU's argument, recomputed
            dz = ops.gelu_bwd_bf16(z, g.contiguous())
            del z
            du, grads[4 * li + 2], grads[4 * li + 3] = ops.layernorm_bwd_bf16(u, dz, _f32(gam), 1e-5)
            del dz
            if li == 0:
                C0 = cl[0][0]
                dw0, db0 = ops.conv0_wgrad(wav, du.contiguous(), C0, T0, P0)
                grads[0], grads[1] = dw0.view_as(w), db0
                break
            C = cl[li - 1][0]
            grads[4 * li + 1] = ops.colsum_bf16(du)
            grads[4 * li], dx = conv_layer_backward(acts[li - 1], w, du, B, rows_out, dim, k, s, C)
            rows_out = rows_out * s
            g = dx[:B * rows_out]
        return (None, None, None, None, *grads)

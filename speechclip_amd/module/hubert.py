"""HuBERT speech encoder on MI355X: parameter container with fairseq checkpoint key names + the
HIP execution engine (conv stack as overlapping-row GEMMs, grouped positional conv, transformer
layers with varlen flash attention).

Replaces what the reference obtains from fairseq (`HubertModel`, not vendored) and drives through
`customFunc_hubert_forward` / `custom_FairseqTransformerEncoder_extract_features`
(avssl/module/speech_encoder_plus.py:29-107).  Parameter names follow fairseq so the reference's
checkpoints (`audio_encoder.encoder.*`) load by key name.

Layout notes (DESIGN.md has the full picture):
  * activations are channels-last bf16 [B, P_l, C]; P_0 = ceil(T_0/64)*64 and P_{l+1} = P_l / 2, so all
    utterances of the batch flatten into ONE GEMM per conv layer (row m = b*P_l + t, lda = stride*C, K = k*C);
    rows t >= T_l are finite don't-care rows that never reach a valid output.
  * the transformer runs on T_p = P_0/64 >= T rows per utterance; rows >= T are treated exactly like
    padded frames (zeroed before the positional conv, masked as keys).
"""
import math
from dataclasses import dataclass, field
from typing import List, Sequence, Tuple

import os

import torch
import torch.nn as nn

from .. import ops
from ..ops import ACT_GELU, ACT_NONE

# Launch-order hook (kwClip.forward): called with "extractor" behind the conv feature extractor and "layer<i>" in front of transformer layer i, so that the
# caller can enqueue independent work (the image tower on its side stream) at that point of the speech tower's launch sequence.  None: no calls.
STAGE_HOOK = None

CONV_LAYERS = [(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512, 2, 2)] * 2


# SC_CONV0_LN_FUSED=0 (A/B): conv layer 0 of a "layer_norm" extractor as conv + bias, then a separate LayerNorm + GELU pass (rounds 1-5)
_CONV0_LN_FUSED = os.environ.get("SC_CONV0_LN_FUSED", "1") != "0"
# Operand format of the frozen PRE-LN transformer layers (HuBERT-large): IEEE half (default) -- LayerNorm outputs, q|k|v, attention probabilities and outputs, GELU
# outputs and the layer weights carry 11 significand bits instead of bf16's 8; the residual stream is fp32 either way.  This is the precision the reference runs
# these models at on a GPU (fp16 autocast: config/speechCLIP/model_large/coco/spchclp_p.yaml:122); 24 layers of bf16 operand rounding were what held the
# P-large parity floor at 0.985.  SC_PRELN_F16=0 (A/B): bf16 operands, rounds 1-5.  Post-LN models (HuBERT-base) keep bf16: their residual
# stream itself is 16-bit, and bf16's range is what that needs.
_PRELN_F16 = os.environ.get("SC_PRELN_F16", "1") != "0"


@dataclass
class HubertConfig:
    extractor_mode: str = "default"      # "default": GroupNorm on conv layer 0 (base); "layer_norm": LN on every layer (large)
    conv_bias: bool = False
    conv_layers: List[Tuple[int, int, int]] = field(default_factory=lambda: list(CONV_LAYERS))
    encoder_layers: int = 12
    encoder_embed_dim: int = 768
    encoder_ffn_embed_dim: int = 3072
    encoder_attention_heads: int = 12
    layer_norm_first: bool = False
    conv_pos: int = 128
    conv_pos_groups: int = 16
    normalize: bool = False              # fairseq task.cfg.normalize (per-utterance wave layer-norm)
    feature_grad_mult: float = 0.1       # [3P fairseq hubert_base_librispeech: 0.1; hubert_large_librivox: 1.0]: scale of the gradient entering the conv extractor
    # the checkpoint's dropouts [3P fairseq hubert_base_librispeech; hubert_large_librivox has 0 everywhere]: applied by the FROZEN encoder too while
    # the module is in train mode (speech_encoder_plus.py:42, :87 and the layers' dropout modules)
    dropout: float = 0.1
    attention_dropout: float = 0.1
    activation_dropout: float = 0.0
    dropout_input: float = 0.1
    encoder_layerdrop: float = 0.05      # the checkpoint's own rate [3P fairseq hubert_base_librispeech: 0.05, hubert_large_librivox: 0.0]; used
                                         # only with audio_encoder.layer_drop: "original" (speech_encoder_plus.py:411-412)

    @staticmethod
    def from_name(name: str) -> "HubertConfig":
        if name in ("hubert", "hubert_base"):
            return HubertConfig()
        if name == "hubert_large_ll60k":
            return HubertConfig(extractor_mode="layer_norm", conv_bias=True, encoder_layers=24, encoder_embed_dim=1024,
                                encoder_ffn_embed_dim=4096, encoder_attention_heads=16, layer_norm_first=True, normalize=True,
                                encoder_layerdrop=0.0, feature_grad_mult=1.0, dropout=0.0, attention_dropout=0.0, activation_dropout=0.0,
                                dropout_input=0.0)
        raise KeyError(name)


class _Holder(nn.Module):
    """Parameter holder whose children are registered under explicit (numeric) names."""


def _conv_block(cfg: HubertConfig, i: int, in_d: int, dim: int, k: int) -> nn.Module:
    blk = _Holder()
    conv = _Holder()
    conv.weight = nn.Parameter(torch.empty(dim, in_d, k))
    nn.init.kaiming_normal_(conv.weight)
    if cfg.conv_bias:
        conv.bias = nn.Parameter(torch.zeros(dim))
    blk.add_module("0", conv)
    if cfg.extractor_mode == "layer_norm":
        wrap = _Holder()
        wrap.add_module("1", nn.LayerNorm(dim))
        blk.add_module("2", wrap)
    elif i == 0:
        blk.add_module("2", nn.GroupNorm(dim, dim))
    return blk


class _Attn(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.k_proj, self.v_proj, self.q_proj, self.out_proj = (nn.Linear(d, d) for _ in range(4))


class _EncLayer(nn.Module):
    def __init__(self, cfg: HubertConfig):
        super().__init__()
        d = cfg.encoder_embed_dim
        self.self_attn = _Attn(d)
        self.self_attn_layer_norm = nn.LayerNorm(d)
        self.fc1 = nn.Linear(d, cfg.encoder_ffn_embed_dim)
        self.fc2 = nn.Linear(cfg.encoder_ffn_embed_dim, d)
        self.final_layer_norm = nn.LayerNorm(d)


class _Encoder(nn.Module):
    def __init__(self, cfg: HubertConfig):
        super().__init__()
        d, k, g = cfg.encoder_embed_dim, cfg.conv_pos, cfg.conv_pos_groups
        pc = _Holder()
        v = torch.empty(d, d // g, k)
        nn.init.normal_(v, 0.0, math.sqrt(4.0 / (k * d)))
        pc.weight_v = nn.Parameter(v)
        pc.weight_g = nn.Parameter(v.detach().pow(2).sum(dim=(0, 1), keepdim=True).sqrt())
        pc.bias = nn.Parameter(torch.zeros(d))
        self.pos_conv = _Holder()
        self.pos_conv.add_module("0", pc)
        self.layers = nn.ModuleList([_EncLayer(cfg) for _ in range(cfg.encoder_layers)])
        self.layer_norm = nn.LayerNorm(d)
        self.layerdrop = cfg.encoder_layerdrop       # fairseq TransformerEncoder.layerdrop; FairseqSpeechEncoder_Hubert overrides it (layer_drop)
        self.layer_norm_first = cfg.layer_norm_first


def conv_lengths(L: int, conv_layers) -> List[int]:
    out = []
    for _, k, s in conv_layers:
        L = (L - k) // s + 1
        out.append(L)
    return out


class HubertModel(nn.Module):
    """fairseq-compatible parameter tree; `extract_all_layers` is the HIP forward."""

    def __init__(self, cfg: HubertConfig):
        super().__init__()
        self.cfg = cfg
        fe = _Holder()
        fe.conv_layers = nn.ModuleList()
        in_d = 1
        for i, (dim, k, s) in enumerate(cfg.conv_layers):
            fe.conv_layers.append(_conv_block(cfg, i, in_d, dim, k))
            in_d = dim
        self.feature_extractor = fe
        d = cfg.encoder_embed_dim
        self.post_extract_proj = nn.Linear(in_d, d)
        self.mask_emb = nn.Parameter(torch.empty(d).uniform_())
        self.encoder = _Encoder(cfg)
        self.layer_norm = nn.LayerNorm(in_d)
        self.feature_grad_mult = 1.0
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0.0, 0.02)
                nn.init.zeros_(m.bias)
        self._packed = None
        self._ws = {}
        self.register_load_state_dict_post_hook(lambda module, keys: module.invalidate_packed())

    # ------------------------------------------------------------------ weight packing (load time, not hot path)
    def invalidate_packed(self):
        self._packed = None

    def _apply(self, fn, *a, **k):
        self._packed = None
        self._ws = {}
        return super()._apply(fn, *a, **k)

    def _pack(self, dev):
        cfg = self.cfg
        bf, f32 = torch.bfloat16, torch.float32
        P = {}

        def w16(t):
            return t.detach().to(dev, bf).contiguous()

        def w32(t):
            return None if t is None else t.detach().to(dev, f32).contiguous()

        convs = self.feature_extractor.conv_layers
        c0 = getattr(convs[0], "0")
        P["conv0_w"] = w32(c0.weight.reshape(c0.weight.shape[0], -1))
        P["conv0_b"] = w32(getattr(c0, "bias", None))
        P["conv_ln"] = []
        if cfg.extractor_mode == "layer_norm":
            for blk in convs:
                ln = getattr(getattr(blk, "2"), "1")
                P["conv_ln"].append((w32(ln.weight), w32(ln.bias)))
        else:
            gn = getattr(convs[0], "2")
            P["gn"] = (w32(gn.weight), w32(gn.bias))
        P["conv_w"], P["conv_b"] = [], []
        for i in range(1, len(convs)):
            c = getattr(convs[i], "0")
            P["conv_w"].append(w16(c.weight.permute(0, 2, 1).reshape(c.weight.shape[0], -1)))   # [out, k*in], K idx = tap*C + c_in
            P["conv_b"].append(w32(getattr(c, "bias", None)))
        P["feat_ln"] = (w32(self.layer_norm.weight), w32(self.layer_norm.bias))
        P["proj_w"], P["proj_b"] = w16(self.post_extract_proj.weight), w32(self.post_extract_proj.bias)
        pc = getattr(self.encoder.pos_conv, "0")
        v = pc.weight_v.detach().float()
        wfold = pc.weight_g.detach().float() * v / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()   # weight-norm, dim=2
        d, G, Kw = cfg.encoder_embed_dim, cfg.conv_pos_groups, cfg.conv_pos
        cg = d // G
        P["pos_w"] = w16(wfold.view(G, cg, cg, Kw).permute(0, 1, 3, 2).reshape(G, cg, Kw * cg))
        P["pos_b"] = w32(pc.bias)
        P["enc_ln"] = (w32(self.encoder.layer_norm.weight), w32(self.encoder.layer_norm.bias))
        P["layers"] = []
        P["layer_dtype"] = lw = torch.float16 if (cfg.layer_norm_first and _PRELN_F16) else bf      # operand format of the transformer layers (see _PRELN_F16)

        def wl(t):
            return t.detach().to(dev, lw).contiguous()
        for lyr in self.encoder.layers:
            a = lyr.self_attn
            P["layers"].append(dict(
                wqkv=wl(torch.cat([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight], 0)),
                bqkv=w32(torch.cat([a.q_proj.bias, a.k_proj.bias, a.v_proj.bias], 0)),
                wo=wl(a.out_proj.weight), bo=w32(a.out_proj.bias),
                ln1=(w32(lyr.self_attn_layer_norm.weight), w32(lyr.self_attn_layer_norm.bias)),
                w1=wl(lyr.fc1.weight), b1=w32(lyr.fc1.bias), w2=wl(lyr.fc2.weight), b2=w32(lyr.fc2.bias),
                ln2=(w32(lyr.final_layer_norm.weight), w32(lyr.final_layer_norm.bias))))
        return P

    @staticmethod
    def _conv0_ln_coef(P):
        """gamma | beta | eps of the extractor's first LayerNorm as ONE f32 tensor (the `coef` operand of sc_conv0_fwd mode 2), built once per packed weight set."""
        if "conv0_ln_coef" not in P:
            g, b = P["conv_ln"][0]
            P["conv0_ln_coef"] = torch.cat([g.float().flatten(), b.float().flatten(), g.new_tensor([1e-5], dtype=torch.float32)]).contiguous()
        return P["conv0_ln_coef"]

    def _buf(self, name, shape, dtype, dev, zero=False, cap=False):
        """Workspace tensors, reused across steps.  cap=False: keyed by the exact shape (fixed-shape batches).  cap=True (packed batches, whose
        row count changes every step): ONE flat allocation per (name, dtype) that only grows (in 1/8 steps, zero-filled so rows nobody wrote
        stay finite), viewed at the requested shape."""
        if cap:
            key = (name, "cap", dtype)
            need = 1
            for v in shape:
                need *= int(v)
            t = self._ws.get(key)
            if t is None or t.device != dev or t.numel() < need:
                self._ws.pop(key, None)
                t = None                                          # free the old block before growing
                t = torch.zeros(need + need // 8, device=dev, dtype=dtype)
                self._ws[key] = t
            return t[:need].view(*shape)
        key = (name, tuple(shape), dtype)
        t = self._ws.get(key)
        if t is None or t.device != dev:
            t = (torch.zeros if zero else torch.empty)(shape, device=dev, dtype=dtype)
            self._ws[key] = t
        return t

    # ------------------------------------------------------------------ geometry helpers (host ints)
    def frame_geometry(self, lmax: int):
        cl = self.cfg.conv_layers
        lens = conv_lengths(lmax, cl)
        T0, T = lens[0], lens[-1]
        ds = 1
        for _, _, s in cl[1:]:
            ds *= s
        P0 = (T0 + ds - 1) // ds * ds
        return T0, T, P0, P0 // ds

    @staticmethod
    def valid_frames(lens: Sequence[int], lmax: int, T: int) -> List[int]:
        """fairseq forward_padding_mask rule: trim lmax % T samples, chunk = remaining / T, a frame is padding iff
        ALL its samples are padding  =>  valid = ceil(len / chunk), clamped to T."""
        chunk = (lmax - lmax % T) // T
        return [min(T, -(-int(l) // chunk)) for l in lens]

    def packed_geometry(self, lens: Sequence[int], lmax: int, need_rows: Sequence[int] = None):
        """Row allotment of the padding-free engine.  The reference pads every utterance to the batch maximum and runs the conv stack and the
        transformer GEMMs on B x T rows (speech_encoder_plus.py:506-518, :540-556); only frames below each utterance's own length reach an
        output.  Here utterance b gets r_b = max(valid_b, need_b) + 1 transformer rows (valid_b: fairseq's padding-mask rule; need_b: rows the
        caller will read, e.g. round(len / 320)) and 2^(6-l) r_b rows at conv layer l, laid back to back: row offsets double per level
        (o_l = 2 o_{l+1}), so every stride-2 conv layer is still ONE overlapping-row GEMM over the whole batch.  The extra row covers the
        receptive-field halo: output frame F - 1 needs 64 F + 15 layer-0 frames <= 64 (F + 1); they are computed from the zero-padded wave
        exactly as the padded layout computes them (GroupNorm statistics stay those of the padded length: sc_conv0_gn_coef over all T0
        frames, zero samples adding nothing)."""
        T0, T, P0, Tp = self.frame_geometry(lmax)
        valid = self.valid_frames(lens, lmax, T)
        need = [0] * len(lens) if need_rows is None else [min(int(n), T) for n in need_rows]
        rows = [max(v, n) + 1 for v, n in zip(valid, need)]
        off = [0]
        for r in rows:
            off.append(off[-1] + r)
        ds = P0 // Tp
        return dict(rows=rows, row_off=off, total=off[-1], rows_max=max(rows), valid=valid, scale0=ds, T0=T0, T=T, padded_rows=len(lens) * Tp)

    # ------------------------------------------------------------------ forward
    def dropout_rates(self):
        c = self.cfg
        return dict(features=float(c.dropout_input), hidden=float(c.dropout), attention=float(c.attention_dropout), activation=float(c.activation_dropout))

    @torch.no_grad()
    def extract_all_layers(self, wav: torch.Tensor, lens: Sequence[int], stop_layer: int = None, drop_layers=(),
                           dropout_seed: int = None, pack: dict = None):
        """wav: f32 [B, Lmax] device tensor (right zero-padded); lens: host ints.
        Returns (hidden [n_layers+1, B, Tp, d] (bf16 for post-LN, f32 for pre-LN), T, Tp, valid_frames); with `stop_layer` = L only
        hidden[0..L] are computed.  `drop_layers` (layerdrop, speech_encoder_plus.py:49-53): the listed layers are skipped and leave NO entry
        in the result -- hidden has 1 + (layers run) states, as the reference's `layer_results`.
        `dropout_seed` (train mode of the FROZEN encoder, post-LN models): the checkpoint's dropouts are applied -- dropout_input on the projected
        features, `dropout` after the positional conv + LayerNorm and after out_proj / fc2 (before the residual add), attention_dropout on the
        probabilities, activation_dropout after the GELU; masks are counter-based, one stream per site derived from the seed.
        pack (packed_geometry): the padding-free layout -- every tensor has pack["total"] rows, utterance b at rows pack["row_off"][b] ...;
        returns (hidden [n, total, d], T, None, valid); ops.unpack_rows restores the padded [B, T, d] view where a caller needs it."""
        cfg = self.cfg
        dev = wav.device

        def buf(name, shape, dtype, dev_, zero=False):
            # every large workspace is ONE flat allocation per name that only grows: a job whose batch maximum (hence T) differs from step to step
            # would otherwise keep a full set of buffers per distinct shape (tools/soak_varlen.py: +107 GB over ten distinct T before this).
            return self._buf(name, shape, dtype, dev_, zero=zero, cap=True)
        # packed bf16 operands are rebuilt when a load replaced the weights (post-hook) or an optimizer step moved TRAINABLE encoder weights:
        # ops.param_epoch moves on FusedAdam steps (raw-pointer writes), the tensors' own `_version` on any torch optimizer (the fallback of
        # configure_optimizers for optim.name != "Adam" / CPU params) or in-place edit; frozen encoders never repack
        trainable = [p for p in self.parameters() if p.requires_grad]
        epoch = (ops.param_epoch(*trainable), sum(p._version for p in trainable)) if trainable else -1
        if self._packed is None or getattr(self, "_packed_epoch", -1) != epoch:
            self._packed = self._pack(dev)
            self._packed_epoch = epoch
        P = self._packed
        B, lmax = wav.shape
        T0, T, P0, Tp = self.frame_geometry(lmax)
        assert T >= 1, "waveform too short for the conv stack"
        C = cfg.conv_layers[0][0]
        bf = torch.bfloat16
        lens_i32 = ops.dev_ints(lens, torch.int32, dev)
        if cfg.normalize:
            wav = ops.wave_layernorm(wav.contiguous(), lens_i32)
        ln_mode = cfg.extractor_mode == "layer_norm"
        if pack is not None:
            assert pack["scale0"] * Tp == P0
            off_i32 = ops.dev_ints(pack["row_off"], torch.int32, dev)
            Mt = pack["total"]
            rows_all = pack["scale0"] * Mt            # rows of the whole batch at conv layer 0
        else:
            rows_all = B * P0
        # ---- conv layer 0
        x = buf("conv0", (rows_all + 8, C), bf, dev, zero=True)
        if pack is not None:
            if ln_mode and C % 64 == 0 and _CONV0_LN_FUSED:      # conv + bias + LayerNorm + GELU in one kernel (sc_conv0_fwd mode 2)
                ops.conv0_packed(wav, P["conv0_w"], T0, off_i32, pack["scale0"], pack["rows_max"], Mt, bias=P["conv0_b"], out=x, ln_coef=self._conv0_ln_coef(P))
            elif ln_mode:
                ops.conv0_packed(wav, P["conv0_w"], T0, off_i32, pack["scale0"], pack["rows_max"], Mt, bias=P["conv0_b"], out=x)
                ops.layernorm(x[:rows_all], *P["conv_ln"][0], gelu=True, out=x[:rows_all])
            else:
                ops.conv0_packed(wav, P["conv0_w"], T0, off_i32, pack["scale0"], pack["rows_max"], Mt, gn_gamma=P["gn"][0], gn_beta=P["gn"][1], out=x)
        elif ln_mode and C % 64 == 0 and P0 % 64 == 0 and _CONV0_LN_FUSED:
            ops.conv0(wav, P["conv0_w"], T0, P0, bias=P["conv0_b"], out=x, ln_coef=self._conv0_ln_coef(P))
        elif ln_mode:
            ops.conv0(wav, P["conv0_w"], T0, P0, bias=P["conv0_b"], out=x)
            ops.layernorm(x[: B * P0], *P["conv_ln"][0], gelu=True, out=x[: B * P0])
        else:
            ops.conv0(wav, P["conv0_w"], T0, P0, gn_gamma=P["gn"][0], gn_beta=P["gn"][1], out=x)
        # ---- conv layers 1.. as overlapping-row GEMMs (packed batches: the same ONE GEMM per layer, over sum_b rows_b rows)
        for i, (dim, k, s) in enumerate(cfg.conv_layers[1:]):
            rows_all //= s
            y = buf(f"conv{i + 1}", (rows_all + 8, dim), bf, dev, zero=True)
            ops.gemm(x, P["conv_w"][i], P["conv_b"][i], ACT_NONE if ln_mode else ACT_GELU, out=y[:rows_all],
                     M=rows_all, K=k * C, lda=s * C)
            if ln_mode:
                ops.layernorm(y[:rows_all], *P["conv_ln"][i + 1], gelu=True, out=y[:rows_all])
            x, C = y, dim
        assert rows_all == (pack["total"] if pack is not None else B * Tp)
        M = rows_all
        d = cfg.encoder_embed_dim
        # ---- feature LayerNorm + projection
        feats = ops.layernorm(x[:M], *P["feat_ln"], out=buf("feat_ln", (M, C), bf, dev))
        xp = ops.gemm(feats, P["proj_w"], P["proj_b"], out=buf("proj", (M, d), bf, dev))
        rates = self.dropout_rates() if dropout_seed is not None else None
        if rates is not None and any(v > 0 for v in rates.values()):
            if cfg.layer_norm_first:
                raise NotImplementedError("train-mode dropout inside a pre-LN encoder is not built (the released large checkpoint has all rates 0)")
        else:
            rates = None
        site = [int(dropout_seed) & 0x7fffffff if dropout_seed is not None else 0]

        def next_seed():       # one mask stream per dropout site, in forward order
            site[0] = (site[0] * 1103515245 + 12345) & 0x7fffffff
            return site[0]
        if rates and rates["features"] > 0:
            ops.dropout_bf16(xp, rates["features"], next_seed(), out=xp)                      # dropout_input (speech_encoder_plus.py:87)
        if STAGE_HOOK is not None:
            STAGE_HOOK("extractor")
        # ---- frame mask, positional conv (+ LN for post-LN models)
        valid = self.valid_frames(lens, lmax, T)
        valid_i32 = ops.dev_ints(valid, torch.int32, dev)
        nl = cfg.encoder_layers
        pre_ln = cfg.layer_norm_first
        hid_dtype = torch.float32 if pre_ln else bf
        hidden = buf("hidden", (nl + 1, M, d), hid_dtype, dev)
        g, bta = (None, None) if pre_ln else P["enc_ln"]
        if pack is not None:
            ops.posconv_packed(xp, valid_i32, off_i32, P["pos_w"], P["pos_b"], g, bta, B, pack["rows_max"], M, d, cfg.conv_pos_groups, cfg.conv_pos,
                               out=hidden[0])
        else:
            ops.posconv(xp, valid_i32, P["pos_w"], P["pos_b"], g, bta, B, Tp, d, cfg.conv_pos_groups, cfg.conv_pos, out=hidden[0])
        if rates and rates["hidden"] > 0:
            ops.dropout_bf16(hidden[0], rates["hidden"], next_seed(), out=hidden[0])            # F.dropout before the layers (:42); layer_results[0] is the dropped state
        # ---- transformer layers
        H = cfg.encoder_attention_heads
        lw = P["layer_dtype"]                      # bf16, or IEEE half for pre-LN models (_PRELN_F16): every 16-bit tensor inside a layer
        qkv = buf("qkv", (M, 3 * d), lw, dev)
        att = buf("att", (M, d), lw, dev)
        ffn = buf("ffn", (M, cfg.encoder_ffn_embed_dim), lw, dev)
        tmp = buf("tmp", (M, d), lw, dev)
        tmp2 = buf("tmp2", (M, d), lw, dev)

        def attn(qkv_, att_):
            if pack is not None:
                ops.attention_packed(qkv_, B, pack["rows_max"], H, valid_i32, off_i32, out=att_)
            else:
                ops.attention(qkv_, B, Tp, H, valid_i32, out=att_)
        kept = 0
        # (measured, round 3: moving the residual add out of the out-proj / fc2 epilogues into one LayerNorm(residual + x) pass -- the residual
        #  variant of the GEMM is 20 % slower than the plain one in isolation -- is worth 0.05 ms per step: the bytes only change kernels)
        for i, L in enumerate(P["layers"]):
            if stop_layer is not None and i >= stop_layer:      # fine-tuning: the layers from here on run as one autograd node (train_hubert.py)
                break
            if i in drop_layers:
                continue
            h, h_out = hidden[kept], hidden[kept + 1]
            kept += 1
            if STAGE_HOOK is not None:
                STAGE_HOOK("layer%d" % i)
            if not pre_ln and rates:      # [3P fairseq] TransformerSentenceEncoderLayer in train mode: x = LN(x + dropout1(attn(x))); x = LN(x + dropout3(fc2(dropout2(act(fc1 x)))))
                ops.gemm(h, L["wqkv"], L["bqkv"], out=qkv)
                if pack is not None:
                    ops.attention_packed(qkv, B, pack["rows_max"], H, valid_i32, off_i32, out=att, drop_p=rates["attention"], seed=next_seed())
                else:
                    ops.attention_dropout(qkv, B, Tp, H, valid_i32, rates["attention"], next_seed(), out=att)
                ops.gemm(att, L["wo"], L["bo"], out=tmp)
                ops.dropout_add_layernorm(tmp, h, *L["ln1"], rates["hidden"], next_seed(), out=tmp2)
                ops.gemm(tmp2, L["w1"], L["b1"], ACT_GELU, out=ffn)
                if rates["activation"] > 0:
                    ops.dropout_bf16(ffn, rates["activation"], next_seed(), out=ffn)
                ops.gemm(ffn, L["w2"], L["b2"], out=tmp)
                ops.dropout_add_layernorm(tmp, tmp2, *L["ln2"], rates["hidden"], next_seed(), out=h_out)
            elif not pre_ln:
                ops.gemm(h, L["wqkv"], L["bqkv"], out=qkv)
                attn(qkv, att)
                ops.gemm(att, L["wo"], L["bo"], residual=h, out=tmp)
                ops.layernorm(tmp, *L["ln1"], out=tmp2)
                ops.gemm(tmp2, L["w1"], L["b1"], ACT_GELU, out=ffn)
                ops.gemm(ffn, L["w2"], L["b2"], residual=tmp2, out=tmp)
                ops.layernorm(tmp, *L["ln2"], out=h_out)
            else:
                xmid = buf("xmid", (M, d), torch.float32, dev)
                ops.layernorm(h, *L["ln1"], out=tmp)
                ops.gemm(tmp, L["wqkv"], L["bqkv"], out=qkv)
                attn(qkv, att)
                ops.gemm(att, L["wo"], L["bo"], residual=h, out=xmid, out_f32=True)
                ops.layernorm(xmid, *L["ln2"], out=tmp)
                ops.gemm(tmp, L["w1"], L["b1"], ACT_GELU, out=ffn)
                ops.gemm(ffn, L["w2"], L["b2"], residual=xmid, out=h_out, out_f32=True)
        if pack is not None:
            return (hidden[: kept + 1] if (drop_layers or stop_layer is not None) else hidden), T, None, valid
        if drop_layers:
            return hidden[: kept + 1].view(kept + 1, B, Tp, d), T, Tp, valid
        return hidden.view(nl + 1, B, Tp, d), T, Tp, valid

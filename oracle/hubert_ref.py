"""fp32 CPU restatement of fairseq HuBERT (test oracle; see oracle/__init__.py).

The arithmetic is the published algorithm of fairseq @ b5a039c (requirements.txt:6 of the
reference; sources are NOT under /root/reference): `fairseq/models/hubert/hubert.py`
(HubertModel.forward_features / forward_padding_mask), `fairseq/models/wav2vec/wav2vec2.py`
(ConvFeatureExtractionModel, TransformerEncoder, TransformerSentenceEncoderLayer,
make_conv_pos) and `fairseq/modules/multihead_attention.py`.  Parameter names follow
fairseq so that a checkpoint `state_dict` loads unchanged.

The call protocol (which attributes are touched, in what order) follows the reference's
call sites:
  * avssl/module/speech_encoder_plus.py:67-107  (customFunc_hubert_forward)
  * avssl/module/speech_encoder_plus.py:29-64   (custom_FairseqTransformerEncoder_extract_features)
`hubert_forward` below restates those two functions for eval mode (dropout off,
layerdrop 0, mask=None); the classes also expose the attribute protocol so that the
reference's own patched functions can run over them (tests/golden/make_golden.py).
"""
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import math
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

CONV_LAYERS = [(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512, 2, 2)] * 2


@dataclass
class HubertRefConfig:
    extractor_mode: str = "default"  # "default" (base: GroupNorm on layer 0) | "layer_norm" (large)
    conv_bias: bool = False
    conv_layers: List[Tuple[int, int, int]] = field(default_factory=lambda: list(CONV_LAYERS))
    encoder_layers: int = 12
    encoder_embed_dim: int = 768
    encoder_ffn_embed_dim: int = 3072
    encoder_attention_heads: int = 12
    layer_norm_first: bool = False
    conv_pos: int = 128
    conv_pos_groups: int = 16
    normalize: bool = False  # task.cfg.normalize: per-utterance wave layer-norm

    @staticmethod
    def base():
        return HubertRefConfig()

    @staticmethod
    def large():
        return HubertRefConfig(extractor_mode="layer_norm", conv_bias=True, encoder_layers=24,
                               encoder_embed_dim=1024, encoder_ffn_embed_dim=4096,
                               encoder_attention_heads=16, layer_norm_first=True, normalize=True)

    @staticmethod
    def tiny(layer_norm_first=False, extractor_mode="default", conv_bias=False):
        # smallest shape the MI355X kernels accept: head_dim 64, conv K = k*C a multiple of 64
        return HubertRefConfig(extractor_mode=extractor_mode, conv_bias=conv_bias,
                               conv_layers=[(64, 10, 5)] + [(64, 3, 2)] * 4 + [(64, 2, 2)] * 2,
                               encoder_layers=2, encoder_embed_dim=128, encoder_ffn_embed_dim=256,
                               encoder_attention_heads=2, layer_norm_first=layer_norm_first,
                               conv_pos=16, conv_pos_groups=4, normalize=layer_norm_first)


class _TransposeLast(nn.Module):
    def forward(self, x):
        return x.transpose(-2, -1)


class ConvFeatureExtractionModel(nn.Module):
    """7 Conv1d blocks: conv -> dropout(0) -> [norm] -> GELU(erf).  [3P wav2vec2.py]"""

    def __init__(self, cfg: HubertRefConfig):
        super().__init__()
        self.conv_layers = nn.ModuleList()
        in_d = 1
        for i, (dim, k, s) in enumerate(cfg.conv_layers):
            conv = nn.Conv1d(in_d, dim, k, stride=s, bias=cfg.conv_bias)
            nn.init.kaiming_normal_(conv.weight)
            if cfg.extractor_mode == "layer_norm":
                norm = nn.Sequential(_TransposeLast(), nn.LayerNorm(dim, elementwise_affine=True), _TransposeLast())
                block = nn.Sequential(conv, nn.Dropout(0.0), norm, nn.GELU())
            elif cfg.extractor_mode == "default" and i == 0:
                block = nn.Sequential(conv, nn.Dropout(0.0), nn.GroupNorm(dim, dim, affine=True), nn.GELU())
            else:
                block = nn.Sequential(conv, nn.Dropout(0.0), nn.GELU())
            self.conv_layers.append(block)
            in_d = dim

    def forward(self, x):  # [B, L] -> [B, C, T]
        x = x.unsqueeze(1)
        for block in self.conv_layers:
            x = block(x)
        return x


class MultiheadAttentionRef(nn.Module):
    """Separate q/k/v/out projections, q scaled by head_dim^-0.5, -inf on padded keys."""

    def __init__(self, d, heads):
        super().__init__()
        self.embed_dim, self.num_heads, self.head_dim = d, heads, d // heads
        self.scaling = self.head_dim ** -0.5
        self.k_proj = nn.Linear(d, d)
        self.v_proj = nn.Linear(d, d)
        self.q_proj = nn.Linear(d, d)
        self.out_proj = nn.Linear(d, d)

    def forward(self, x, key_padding_mask=None):  # x: [T, B, C]
        T, B, C = x.shape
        H, hd = self.num_heads, self.head_dim
        q = self.q_proj(x) * self.scaling
        k = self.k_proj(x)
        v = self.v_proj(x)
        q = q.contiguous().view(T, B * H, hd).transpose(0, 1)
        k = k.contiguous().view(T, B * H, hd).transpose(0, 1)
        v = v.contiguous().view(T, B * H, hd).transpose(0, 1)
        w = torch.bmm(q, k.transpose(1, 2))  # [B*H, T, T]
        if key_padding_mask is not None:
            w = w.view(B, H, T, T).masked_fill(key_padding_mask[:, None, None, :].bool(), float("-inf")).view(B * H, T, T)
        w = torch.softmax(w.float(), dim=-1).type_as(w)
        o = torch.bmm(w, v).transpose(0, 1).contiguous().view(T, B, C)
        return self.out_proj(o)


class TransformerSentenceEncoderLayerRef(nn.Module):
    def __init__(self, cfg: HubertRefConfig):
        super().__init__()
        d = cfg.encoder_embed_dim
        self.layer_norm_first = cfg.layer_norm_first
        self.self_attn = MultiheadAttentionRef(d, cfg.encoder_attention_heads)
        self.self_attn_layer_norm = nn.LayerNorm(d)
        self.fc1 = nn.Linear(d, cfg.encoder_ffn_embed_dim)
        self.fc2 = nn.Linear(cfg.encoder_ffn_embed_dim, d)
        self.final_layer_norm = nn.LayerNorm(d)

    def forward(self, x, self_attn_padding_mask=None, need_weights=False):  # [T, B, C]
        if self.layer_norm_first:
            x = x + self.self_attn(self.self_attn_layer_norm(x), self_attn_padding_mask)
            x = x + self.fc2(F.gelu(self.fc1(self.final_layer_norm(x))))
        else:
            x = self.self_attn_layer_norm(x + self.self_attn(x, self_attn_padding_mask))
            x = self.final_layer_norm(x + self.fc2(F.gelu(self.fc1(x))))
        return x, None


class _SamePad(nn.Module):
    def __init__(self, k):
        super().__init__()
        self.remove = 1 if k % 2 == 0 else 0

    def forward(self, x):
        return x[:, :, : -self.remove] if self.remove else x


class _WeightNormConv1d(nn.Module):
    """Conv1d with weight-norm over dim=2: w[:,:,k] = g[0,0,k] * v[:,:,k] / ||v[:,:,k]||.
    Holds fairseq's checkpoint keys `weight_g` [1,1,K], `weight_v` [d, d/groups, K], `bias`."""

    def __init__(self, d, k, groups):
        super().__init__()
        self.groups, self.k = groups, k
        v = torch.empty(d, d // groups, k)
        nn.init.normal_(v, mean=0.0, std=math.sqrt(4.0 / (k * d)))
        self.weight_v = nn.Parameter(v)
        self.weight_g = nn.Parameter(v.detach().pow(2).sum(dim=(0, 1), keepdim=True).sqrt())
        self.bias = nn.Parameter(torch.zeros(d))

    def folded_weight(self):
        v = self.weight_v
        return self.weight_g * v / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()

    def forward(self, x):  # [B, C, T]
        return F.conv1d(x, self.folded_weight(), self.bias, padding=self.k // 2, groups=self.groups)


class TransformerEncoderRef(nn.Module):
    def __init__(self, cfg: HubertRefConfig):
        super().__init__()
        d = cfg.encoder_embed_dim
        self.dropout = 0.0
        self.layerdrop = 0.0
        self.layer_norm_first = cfg.layer_norm_first
        self.pos_conv = nn.Sequential(_WeightNormConv1d(d, cfg.conv_pos, cfg.conv_pos_groups),
                                      _SamePad(cfg.conv_pos), nn.GELU())
        self.layers = nn.ModuleList([TransformerSentenceEncoderLayerRef(cfg) for _ in range(cfg.encoder_layers)])
        self.layer_norm = nn.LayerNorm(d)

    # protocol used by the reference's patched code (speech_encoder_plus.py:101)
    def forward(self, x, padding_mask=None, layer=None):
        x, layer_results = self.extract_features(x, padding_mask, layer)
        if self.layer_norm_first and layer is None:
            x = self.layer_norm(x)
        return x, layer_results

    def extract_features(self, x, padding_mask=None, tgt_layer=None):
        """Eval-mode restatement of speech_encoder_plus.py:29-64."""
        if padding_mask is not None:
            x = x.masked_fill(padding_mask[:, :, None], 0.0)          # :32-33 index_put(x, mask, 0)
        x = x + self.pos_conv(x.transpose(1, 2)).transpose(1, 2)      # :35-37
        if not self.layer_norm_first:
            x = self.layer_norm(x)                                     # :39-40
        x = x.transpose(0, 1)                                          # :45 B,T,C -> T,B,C
        layer_results = [x.transpose(0, 1)]                            # :47
        for layer in self.layers:                                      # :49-53: one draw per layer in EVERY mode; skipped only when training
            dropout_probability = np.random.random()
            if not self.training or (dropout_probability > self.layerdrop):
                x, _ = layer(x, self_attn_padding_mask=padding_mask, need_weights=False)
                layer_results.append(x.transpose(0, 1))
        return x.transpose(0, 1), layer_results


class _GradMultiply(torch.autograd.Function):
    """[3P fairseq modules/grad_multiply.py]: identity forward, gradient scaled."""

    @staticmethod
    def forward(ctx, x, scale):
        ctx.scale = scale
        return x.new(x)

    @staticmethod
    def backward(ctx, grad):
        return grad * ctx.scale, None


class HubertModelRef(nn.Module):
    def __init__(self, cfg: HubertRefConfig):
        super().__init__()
        self.cfg = cfg
        embed = cfg.conv_layers[-1][0]
        d = cfg.encoder_embed_dim
        self.feature_extractor = ConvFeatureExtractionModel(cfg)
        self.post_extract_proj = nn.Linear(embed, d) if embed != d else None
        self.dropout_input = nn.Dropout(0.0)
        self.dropout_features = nn.Dropout(0.0)
        self.feature_grad_mult = 1.0
        self.mask_emb = nn.Parameter(torch.empty(d).uniform_())
        self.encoder = TransformerEncoderRef(cfg)
        self.layer_norm = nn.LayerNorm(embed)
        self.apply(_init_bert_params)

    def forward_features(self, source):
        """[3P fairseq hubert.py HubertModel.forward_features]: the extractor's output carries feature_grad_mult times the gradient
        (GradMultiply); feature_grad_mult == 0 runs it without autograd."""
        if self.feature_grad_mult > 0:
            features = self.feature_extractor(source)
            if self.feature_grad_mult != 1.0:
                features = _GradMultiply.apply(features, self.feature_grad_mult)
            return features
        with torch.no_grad():
            return self.feature_extractor(source)

    def forward_padding_mask(self, features, padding_mask):
        """[3P hubert.py] trim Lmax % T samples, view [B, T, -1], frame is pad iff all samples pad."""
        extra = padding_mask.size(1) % features.size(1)
        if extra > 0:
            padding_mask = padding_mask[:, :-extra]
        padding_mask = padding_mask.view(padding_mask.size(0), features.size(1), -1)
        return padding_mask.all(-1)

    def apply_mask(self, x, padding_mask, target_list):  # unused: the reference passes mask=None
        return x, None


def _init_bert_params(m):
    if isinstance(m, nn.Linear):
        nn.init.normal_(m.weight, mean=0.0, std=0.02)
        if m.bias is not None:
            nn.init.zeros_(m.bias)


def randomize_norm_affine(model: nn.Module, gen: torch.Generator, scale: float = 0.2):
    """Make LayerNorm/GroupNorm affines and biases non-trivial so parity tests exercise them."""
    for m in model.modules():
        if isinstance(m, (nn.LayerNorm, nn.GroupNorm)) and m.weight is not None:
            with torch.no_grad():
                m.weight.add_(scale * torch.randn(m.weight.shape, generator=gen))
                m.bias.add_(scale * torch.randn(m.bias.shape, generator=gen))
        if isinstance(m, (nn.Linear, nn.Conv1d, nn.Conv2d)) and m.bias is not None:
            with torch.no_grad():
                m.bias.add_(0.02 * torch.randn(m.bias.shape, generator=gen))
        if isinstance(m, _WeightNormConv1d):
            with torch.no_grad():
                m.bias.add_(0.02 * torch.randn(m.bias.shape, generator=gen))
                m.weight_g.mul_(1.0 + 0.1 * torch.randn(m.weight_g.shape, generator=gen))


def preprocess_input(wavs, normalize: bool):
    """speech_encoder_plus.py:506-518: optional per-utterance layer-norm, right-pad, sample mask."""
    if normalize:
        wavs = [F.layer_norm(w, w.shape) for w in wavs]
    lens = torch.tensor([len(w) for w in wavs], dtype=torch.long)
    lmax = int(lens.max())
    mask = ~(torch.arange(lmax)[None, :] < lens[:, None])
    padded = torch.zeros(len(wavs), lmax, dtype=wavs[0].dtype)
    for i, w in enumerate(wavs):
        padded[i, : len(w)] = w
    return padded, mask


@torch.no_grad()
def hubert_forward(model: HubertModelRef, source, padding_mask):
    """Eval-mode restatement of customFunc_hubert_forward (speech_encoder_plus.py:67-107) with mask=None.

    Returns {"x": [B,T,d], "layer_results": list of (n_layers+1) x [B,T,d]}.
    """
    features = model.forward_features(source)                 # :75   [B,512,T]
    features = features.transpose(1, 2)                       # :77
    features = model.layer_norm(features)                     # :78
    if padding_mask is not None:
        padding_mask = model.forward_padding_mask(features, padding_mask)  # :81-82
    if model.post_extract_proj is not None:
        features = model.post_extract_proj(features)          # :84-85
    x, layer_results = model.encoder(features, padding_mask=padding_mask, layer=None)  # :101
    return {"x": x, "layer_results": layer_results, "padding_mask": padding_mask}


def feat_lengths(wav_lens, downsample_rate: int, t_max: int):
    """speech_encoder_plus.py:604-611: Python round() (banker's) of len/320, clamped to T."""
    return torch.clamp_max(torch.tensor([round(int(l) / downsample_rate) for l in wav_lens], dtype=torch.long), t_max)


def conv_out_length(L: int, conv_layers=CONV_LAYERS) -> int:
    for _, k, s in conv_layers:
        L = (L - k) // s + 1
    return L

"""fp32 CPU restatement of the reference's own glue on the hot path (test oracle).

Each function cites the reference file:line it follows (paths relative to /root/reference).
Eval-mode semantics (dropout off, hard VQ) unless stated.  Pinned against the reference's
glue itself by tests/golden/make_golden.py -> tests/golden/*.npz.
"""
import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .clip_ref import ClipRef, ClipRefConfig
from .hubert_ref import (HubertModelRef, HubertRefConfig, feat_lengths, hubert_forward, preprocess_input)


# ----------------------------------------------------------------------------- small pieces
def keypadding_mask(max_length: int, data_lens: torch.Tensor) -> torch.Tensor:
    """avssl/util/data_utils.py:4-20 -- bool [B, max_length], True = padding."""
    return torch.arange(max_length)[None, :] >= data_lens.reshape(-1, 1).long()


def weighted_sum(hidden: Sequence[torch.Tensor], weights: torch.Tensor, normalize: bool) -> torch.Tensor:
    """avssl/module/weighted_sum.py:26-45 -- softmax(w) . stack(h) (optional per-feature layer_norm)."""
    w = torch.softmax(weights.float(), dim=0)
    x = torch.stack(list(hidden), dim=0)
    if normalize:
        x = F.layer_norm(x, (x.shape[-1],))
    return (w.view(-1, 1, 1, 1) * x).sum(0)


def normalize_hidden_states(hidden: Sequence[torch.Tensor], method: str) -> list:
    """avssl/module/speech_encoder_plus.py:572-592 (`normalize_hiddenstates` with `normalize_type` method1 / method2; "s3prl" is the per-feature
    layer_norm inside WeightedSumLayer instead): method1 = every frame to unit L2 norm, x / (||x|| + 1e-8); method2 = every hidden state divided by its
    utterance's mean frame norm over ALL T frames of the padded batch (padded frames included, as the reference does)."""
    out = []
    for h in hidden:
        if method == "method1":
            out.append(h / (torch.norm(h, dim=-1, keepdim=True) + 1e-8))
        elif method == "method2":
            out.append(h / torch.mean(torch.norm(h, dim=-1), dim=-1).view(-1, 1, 1))
        else:
            raise ValueError(method)
    return out


def l2_normalize(x: torch.Tensor) -> torch.Tensor:
    """avssl/model/kwClip.py:1436,1444-1454 -- x / ||x||, no eps."""
    return x / x.norm(dim=-1, keepdim=True)


def masked_contrastive_loss(feat_a, feat_b, index=None, inv_temperature: float = 1.0 / 0.07,
                            margin: float = 0.0, dcl: bool = False, a2b: bool = True, b2a: bool = True):
    """avssl/module/losses.py:185-245 -- symmetric InfoNCE with id-based false-negative masking.

    `inv_temperature` is the multiplier the reference stores (1/tau, or exp(log-param) when
    trainable, losses.py:160-163,218-221).  No MAX_EYE=256 limit (losses.py:126): the mask is built
    for any batch size; identical for B <= 256.
    """
    assert feat_a.shape == feat_b.shape
    B = feat_a.shape[0]
    eye = torch.eye(B, dtype=torch.bool)
    if index is not None:
        idx = index.reshape(-1, 1)
        neg = idx != idx.t()
    else:
        neg = ~eye
    if not dcl:
        neg = neg | eye
    logits = feat_a.float() @ feat_b.float().t() * inv_temperature
    if margin > 0.0:
        logits = logits - margin * eye.float()
    pos = logits.diagonal()
    e = logits.exp() * neg.float()
    loss = 0.0
    if a2b:
        loss = loss + (-pos + torch.log(e.sum(1))).mean()
    if b2a:
        loss = loss + (-pos + torch.log(e.sum(0))).mean()
    if a2b and b2a:
        loss = loss / 2
    return loss


def mutual_retrieval(score_a: torch.Tensor, score_b: torch.Tensor, ab_answers: torch.Tensor,
                     ba_answers: torch.Tensor, recall_at: Sequence[int]):
    """avssl/module/retrieval.py:6-121 -- recall@K both directions and their mean, in percent."""
    def one_way(score, own_ids, other_ids):
        order = torch.argsort(score, dim=1, descending=True)
        ranked_ids = other_ids[order]                      # [N_own, N_other] ids in rank order
        hit = ranked_ids == own_ids[:, None]
        out = {}
        for k in recall_at:
            kk = min(k, hit.shape[1])
            out[f"recall@{k}"] = (hit[:, :kk].any(dim=1).sum() / hit.shape[0]).item() * 100
        return out
    ab = one_way(score_a, ab_answers, ba_answers)
    ba = one_way(score_b, ba_answers, ab_answers)
    mean = {k: (ab[k] + ba[k]) / 2.0 for k in ab}
    return ab, ba, mean


# ----------------------------------------------------------------------------- attention helpers [3P torch.nn]
def _mha_packed(x, in_w, in_b, out_w, out_b, heads, key_padding_mask=None, attn_mask=None, return_probs=False):
    """torch nn.MultiheadAttention arithmetic (packed in_proj, scale head_dim^-0.5).  x: [B, L, D]."""
    B, L, D = x.shape
    hd = D // heads
    qkv = F.linear(x, in_w, in_b).view(B, L, 3, heads, hd)
    q, k, v = qkv[:, :, 0].transpose(1, 2), qkv[:, :, 1].transpose(1, 2), qkv[:, :, 2].transpose(1, 2)
    s = (q * hd ** -0.5) @ k.transpose(-1, -2)                 # [B, H, L, L]
    if attn_mask is not None:
        s = s + attn_mask
    if key_padding_mask is not None:
        s = s.masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
    p = torch.softmax(s, dim=-1)
    o = p @ v
    y = F.linear(o.transpose(1, 2).reshape(B, L, D), out_w, out_b)
    return (y, p) if return_probs else y


def post_ln_encoder_layer(x, layer: nn.TransformerEncoderLayer, key_padding_mask):
    """[3P torch] nn.TransformerEncoderLayer(norm_first=False, gelu), eval mode."""
    sa = layer.self_attn
    x = layer.norm1(x + _mha_packed(x, sa.in_proj_weight, sa.in_proj_bias, sa.out_proj.weight, sa.out_proj.bias,
                                    sa.num_heads, key_padding_mask))
    return layer.norm2(x + layer.linear2(F.gelu(layer.linear1(x))))


# ----------------------------------------------------------------------------- parallel branch
class _NNTransformerEncoder(nn.Module):  # key names: model.layers.N.*, model.norm.*
    def __init__(self, d, heads, ffn, n_layers):
        super().__init__()
        self.layers = nn.ModuleList([nn.TransformerEncoderLayer(d, heads, ffn, 0.1, "gelu", 1e-5, True, False)
                                     for _ in range(n_layers)])
        self.norm = nn.LayerNorm(d, eps=1e-5)


class _TransformerEncoder(nn.Module):
    def __init__(self, d, heads, ffn, n_layers):
        super().__init__()
        self.model = _NNTransformerEncoder(d, heads, ffn, n_layers)

    def forward(self, src, key_padding_mask):
        for layer in self.model.layers:
            src = post_ln_encoder_layer(src, layer, key_padding_mask)
        return self.model.norm(src)


class ParallelBranchRef(nn.Module):
    """avssl/model/kwClip.py:1004-1108 + avssl/module/kw_modules/TransformerModels.py:48-96."""

    def __init__(self, d_model=768, nhead=8, ffn=3072, n_layers=1, out_dim=512):
        super().__init__()
        self.self_att = _TransformerEncoder(d_model, nhead, ffn, n_layers)
        self.cls = nn.Parameter(torch.randn(1, 1, d_model))
        self.linear_proj = nn.Linear(d_model, out_dim)

    def forward(self, audio_feat, audio_len):
        B, T, D = audio_feat.shape
        src = torch.cat([self.cls.expand(B, -1, -1), audio_feat], dim=1)      # :1089-1090
        mask = keypadding_mask(T + 1, audio_len + 1)                          # :1092-1095
        out = self.self_att(src, mask)                                       # :1097
        return self.linear_proj(out[:, 0])                                   # :1099-1104


# ----------------------------------------------------------------------------- cascaded branch
class _MHAAndNorm(nn.Module):
    """avssl/module/kw_modules/TransformerModels.py:99-125 -- LN(MHA(x) + x)."""

    def __init__(self, d, heads):
        super().__init__()
        self.multihead_attn_layer = nn.MultiheadAttention(d, heads, dropout=0.1, batch_first=True)
        self.attentionBlock_Norm = nn.LayerNorm(d, eps=1e-5)

    def forward(self, src, key_padding_mask):
        m = self.multihead_attn_layer
        y = _mha_packed(src, m.in_proj_weight, m.in_proj_bias, m.out_proj.weight, m.out_proj.bias, m.num_heads,
                        key_padding_mask)
        return self.attentionBlock_Norm(y + src)

    def extract_attention_map(self, src, key_padding_mask):
        """TransformerModels.py:130-135: need_weights=True, average_attn_weights=False -> per-head probabilities [B, H, L, L]."""
        m = self.multihead_attn_layer
        y, p = _mha_packed(src, m.in_proj_weight, m.in_proj_bias, m.out_proj.weight, m.out_proj.bias, m.num_heads,
                           key_padding_mask, return_probs=True)
        return self.attentionBlock_Norm(y + src), p


class _KwBatchNorm(nn.Module):
    """avssl/module/speechclip_c_modules/kw_bn.py:8-164, mode eachKw + parallel (shipped config)."""

    def __init__(self, kw_num, kw_dim, init_bias, init_scale, std_scale=1.0):
        super().__init__()
        self.kw_num, self.kw_dim = kw_num, kw_dim
        self.bn_layer = nn.BatchNorm1d(kw_dim * kw_num)
        with torch.no_grad():
            self.bn_layer.weight.copy_((init_scale * std_scale).repeat(kw_num))
            self.bn_layer.bias.copy_(init_bias.repeat(kw_num))

    def forward(self, kw):  # [B, K, D] -> BN over flattened (D, K) feature order (kw_bn.py:122-131)
        B = kw.shape[0]
        x = kw.permute(0, 2, 1).reshape(B, -1)
        x = self.bn_layer(x)
        return x.reshape(B, self.kw_dim, self.kw_num).permute(0, 2, 1)


def simple_vq(cos_score: torch.Tensor, temp: float, training: bool, prob_msk=(0, 2, 3)) -> Dict:
    """avssl/module/speechclip_c_modules/my_vector_quantizer.py:64-165 (use_gumbel False, hard True)."""
    B, K, V = cos_score.shape
    x = cos_score.reshape(B * K, V).clone()
    x[:, list(prob_msk)] = float("-inf")
    k = x.argmax(-1)
    hard = torch.zeros_like(x).scatter_(-1, k.view(-1, 1), 1.0)
    hard_probs = hard.float().mean(0)
    res = {"num_vars": V}
    res["code_perplexity"] = torch.exp(-(hard_probs * torch.log(hard_probs + 1e-7)).sum(-1)).sum()
    sm = torch.softmax(x.float(), dim=-1)
    avg_probs = sm.mean(0)
    probs_per_t = sm.view(B, K, V).permute(1, 0, 2)
    res["ent_per_t"] = (-(probs_per_t * torch.log(probs_per_t + 1e-9)).sum(-1)).mean(-1)
    res["prob_perplexity"] = torch.exp(-(avg_probs * torch.log(avg_probs + 1e-7)).sum(-1)).sum()
    res["temp"] = float(temp)
    if training:
        soft = torch.softmax(x / temp, dim=-1)
        out = hard + soft - soft.detach()
    else:
        out = hard
    res["subword_prob"] = out.view(B, K, V)
    res["diversity_loss"] = (V - res["prob_perplexity"]) / V
    res["targets"] = out.argmax(-1).view(B, K, 1)
    return res


def encode_keywords(clip: ClipRef, keywords: torch.Tensor, keyword_num: int, sot: int, eot: int) -> torch.Tensor:
    """avssl/module/clip_official.py:220-264 -- [SOT, kw x K, EOT, 0...] through the CLIP text tower."""
    B = keywords.shape[0]
    text = torch.zeros(B, clip.context_length, dtype=torch.long)
    text[:, 0] = sot
    text[:, keyword_num + 1] = eot
    x = clip.token_embedding(text).clone()
    x[:, 1:1 + keyword_num] = keywords
    x = x + clip.positional_embedding
    x = clip.transformer(x.permute(1, 0, 2)).permute(1, 0, 2)
    x = clip.ln_final(x)
    return x[:, 1 + keyword_num] @ clip.text_projection


class CascadedBranchRef(nn.Module):
    """avssl/model/kwClip.py:697-916 (shipped config: MultiheadAttentionAndNorm, 1 head, eachKw parallel BN,
    cosine retrieval, SimpleVectorQuantizer fixed temp 0.1)."""

    def __init__(self, clip: ClipRef, d_model=768, nhead=1, keyword_num=8, vq_temp=0.1, sot=2, eot=3):
        super().__init__()
        object.__setattr__(self, "_clip", clip)   # shared, not re-registered (reference registers a duplicate)
        text_dim = clip.token_embedding.weight.shape[1]
        self.keyword_num, self.text_dim, self.vq_temp, self.sot, self.eot = keyword_num, text_dim, vq_temp, sot, eot
        self.cls = nn.Parameter(torch.randn(1, keyword_num, d_model))
        self.self_att = _MHAAndNorm(d_model, nhead)
        self.linear_proj = nn.Linear(d_model, text_dim)
        emb = clip.token_embedding.weight.detach()
        self.bn_layer = _KwBatchNorm(keyword_num, text_dim, emb.mean(0), emb.std(0))

    def forward(self, audio_feat, audio_len):
        B, T, D = audio_feat.shape
        K = self.keyword_num
        src = torch.cat([self.cls.expand(B, -1, -1), audio_feat], dim=1)          # :869-871
        mask = keypadding_mask(T + K, audio_len + K)                              # :873-875
        kw = self.self_att(src, mask)[:, :K]                                      # :877-881
        kw = self.bn_layer(self.linear_proj(kw))                                  # :883-886
        emb = self._clip.token_embedding.weight
        cos = F.cosine_similarity(kw[:, :, None, :], emb[None, None, :, :], dim=-1)   # :889-898
        vq = simple_vq(cos, self.vq_temp, self.training)                          # :907
        keywords = vq["subword_prob"] @ emb                                       # :909
        feat = encode_keywords(self._clip, keywords, K, self.sot, self.eot)       # :912
        return feat, vq, keywords


def get_attention_map(branch: "CascadedBranchRef", audio_feat, audio_len, decoder=None, reduced_to_original=None, topk: int = 10):
    """KW_CascadedBranch.getAttentionMap (avssl/model/kwClip.py:918-1001): per-utterance keyword-row attention maps
    [H, K, len_i + K] and the `topk` nearest sub-words of every keyword (special ids 0 / 2 / 3 pushed down by 100, :977-979).
    Returns (cls_weights, topk_kw strings or None when no decoder is given, topk ids [B, K, topk], cos scores)."""
    B, T, D = audio_feat.shape
    K = branch.keyword_num
    src = torch.cat([branch.cls.expand(B, -1, -1), audio_feat], dim=1)
    mask = keypadding_mask(T + K, audio_len + K)
    _, w = branch.self_att.extract_attention_map(src, mask)
    cls_weights = [w[i, :, :K, : int(audio_len[i]) + K] for i in range(B)]
    kw = branch.bn_layer(branch.linear_proj(branch.self_att(src, mask)[:, :K]))
    emb = branch._clip.token_embedding.weight
    cos = F.cosine_similarity(kw[:, :, None, :], emb[None, None, :, :], dim=-1).clone()
    cos[..., 0] -= 100
    cos[..., 2] -= 100
    cos[..., 3] -= 100
    ids = torch.topk(cos, dim=-1, k=topk)[1]
    names = None
    if decoder is not None:
        o = (lambda i: reduced_to_original[i]) if reduced_to_original is not None else (lambda i: i)
        names = [[[decoder[o(x.item())].replace("</w>", "") for x in ids[b, k]] for k in range(K)] for b in range(B)]
    return cls_weights, names, ids, cos


def detokenize_keywords(keywords, gold_token_sets, token_embedding, K: int = 10, method: str = "cosine", reduced_to_original=None,
                        chunk: int = 8):
    """The numerical half of KWClipBase.validation_epoch_end's keyword de-tokenisation (avssl/model/kwClip.py:332-420): keywords
    [N, Kw, D], gold_token_sets = one set of ORIGINAL token ids per utterance.  Returns (hit_rate % [Kw], values [N, Kw, K],
    indices [N, Kw, K] into `token_embedding`, first hit token per keyword lists)."""
    N, Kw, D = keywords.shape
    emb = token_embedding.detach().float()
    pinv = torch.linalg.pinv(emb.T).float() if method == "pseudo_inverse" else None
    o = (lambda i: reduced_to_original[i]) if reduced_to_original is not None else (lambda i: i)
    hit = [0] * Kw
    first_hits = [[] for _ in range(Kw)]
    vals, idxs = [], []
    for i in range(0, N, chunk):
        flat = keywords[i:i + chunk].reshape(-1, D).float()
        if pinv is not None:
            score = (pinv @ flat.permute(1, 0)).permute(1, 0)                                              # :362-371
        else:
            score = F.cosine_similarity(flat.view(-1, D, 1), emb.transpose(0, 1).unsqueeze(0), dim=1)     # :372-379
        v, ix = torch.topk(score, K)
        v, ix = v.view(-1, Kw, K), ix.view(-1, Kw, K)
        vals.append(v)
        idxs.append(ix)
        for x in range(v.shape[0]):
            for k in range(Kw):
                common = set(o(t.item()) for t in ix[x, k]) & gold_token_sets[i + x]
                if common:
                    hit[k] += 1
                    first_hits[k].append(int(list(common)[0]))
    return torch.tensor(hit, dtype=torch.float32) / N * 100, torch.cat(vals), torch.cat(idxs), first_hits


# ----------------------------------------------------------------------------- full model
class SpeechClipRef(nn.Module):
    """Eval-mode restatement of KWClip_GeneralTransformer.forward / compute_loss
    (avssl/model/kwClip.py:1385-1478, :1248-1297) over the restated backbones."""

    def __init__(self, hubert_cfg: HubertRefConfig, clip_cfg: ClipRefConfig, parallel: bool = True,
                 cascaded: bool = False, branch_heads: int = 8, normalize_hiddenstates: bool = False,
                 inv_temperature: float = 1.0 / 0.07, keyword_num: int = 8, reduced_vocab: Optional[torch.Tensor] = None):
        super().__init__()
        self.hubert_cfg, self.clip_cfg = hubert_cfg, clip_cfg
        self.encoder = HubertModelRef(hubert_cfg)
        self.clip = ClipRef(clip_cfg)
        sot, eot = clip_cfg.vocab_size - 2, clip_cfg.vocab_size - 1
        if reduced_vocab is not None:      # clip_official.py:62-106: slice token_embedding by id list
            self.clip.token_embedding = nn.Embedding.from_pretrained(self.clip.token_embedding.weight[reduced_vocab])
            ids = reduced_vocab.tolist()
            sot, eot = ids.index(sot), ids.index(eot)
        n_hidden = hubert_cfg.encoder_layers + 1
        self.ws_weights = nn.Parameter(torch.zeros(n_hidden))
        self.normalize_hiddenstates = normalize_hiddenstates
        d = hubert_cfg.encoder_embed_dim
        tw = self.clip.token_embedding.weight.shape[1]
        self.parallel_branch = ParallelBranchRef(d, branch_heads, 4 * d, 1, tw) if parallel else None
        self.cascaded_branch = CascadedBranchRef(self.clip, d, 1, keyword_num, 0.1, sot, eot) if cascaded else None
        self.inv_temperature = inv_temperature
        self.downsample_rate = 320

    @torch.no_grad()
    def forward_audio(self, wav: torch.Tensor, wav_len: torch.Tensor):
        """speech_encoder_plus.py:520-634 (eval): returns (audio_feat [B,T,d], feat_len [B], hidden list)."""
        wavs = [wav[b, : int(wav_len[b])] for b in range(wav.shape[0])]
        padded, mask = preprocess_input(wavs, self.hubert_cfg.normalize)
        out = hubert_forward(self.encoder, padded, mask)
        hidden = out["layer_results"]
        flen = feat_lengths([len(w) for w in wavs], self.downsample_rate, hidden[-1].shape[1])
        feat = weighted_sum(hidden, self.ws_weights, self.normalize_hiddenstates)
        return feat, flen, hidden

    @torch.no_grad()
    def forward(self, batch: Dict[str, torch.Tensor]):
        audio_feat, audio_len, _ = self.forward_audio(batch["wav"], batch["wav_len"])
        image_feat = l2_normalize(self.clip.encode_image(batch["image"]))
        out = {"id": batch["id"], "image_feat": image_feat, "audio_feat": audio_feat, "audio_len": audio_len}
        if self.cascaded_branch is not None:
            c, vq, kw = self.cascaded_branch(audio_feat, audio_len)
            out.update(cascaded_audio_feat=l2_normalize(c), vq_results=vq, keywords=kw)
        if self.parallel_branch is not None:
            out["parallel_audio_feat"] = l2_normalize(self.parallel_branch(audio_feat, audio_len))
        return out

    def compute_loss(self, feats: Dict[str, torch.Tensor], w_par: float = 1.0, w_casc: float = 0.0):
        losses = {"loss": 0.0}
        if w_casc > 0:
            losses["c_cl_loss"] = masked_contrastive_loss(feats["cascaded_audio_feat"], feats["image_feat"], feats["id"],
                                                          self.inv_temperature)
            losses["loss"] = losses["loss"] + w_casc * losses["c_cl_loss"]
        if w_par > 0:
            losses["p_cl_loss"] = masked_contrastive_loss(feats["parallel_audio_feat"], feats["image_feat"], feats["id"],
                                                          self.inv_temperature)
            losses["loss"] = losses["loss"] + w_par * losses["p_cl_loss"]
        return losses

"""FairseqSpeechEncoder_Hubert -- the reference's HuBERT wrapper (avssl/module/speech_encoder_plus.py:319-634)
re-implemented over the MI355X engine in `hubert.py`.

Same constructor arguments, attributes (`out_dim`, `downsample_rate`, `upstream_model_hiddenstates_len`,
`encoder`, `weightedsum_layer`, `trainable_params()`) and `forward(wav, wav_len, feat_select_idx,
return_hidden_states)` contract.  fairseq is not required: `self.encoder` is a parameter tree with fairseq's
checkpoint key names (hubert.HubertModel) whose forward runs on HIP kernels.  No network: a fairseq checkpoint is
loaded only if it is already on disk ($SPEECHCLIP_HUBERT_CKPT or ~/.cache/speechclip_amd/<file name of the URL>).
"""
import logging
import math
import os
from typing import List, Tuple, Union

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from .hubert import HubertConfig, HubertModel
from .weighted_sum import WeightedSumLayer

logger = logging.getLogger(__name__)
FEAT_SELECT_IDX_WEIGHTED_SUM_MODE = "weighted_sum"


def random_crop_max_length(audio: torch.Tensor, max_len: int, orig_len: int = 1000000000) -> torch.Tensor:
    """avssl/data/audio_transforms.py:5-23 (train-mode crop applied inside the encoder forward)."""
    n = min(len(audio), orig_len)
    if max_len < 0 or n <= max_len:
        return audio[:n]
    start = np.random.randint(n - max_len)
    return audio[start:start + max_len]


def _find_local_ckpt(url: str):
    cands = [os.environ.get("SPEECHCLIP_HUBERT_CKPT", ""),
             os.path.join(os.path.expanduser("~/.cache/speechclip_amd"), os.path.basename(url))]
    for c in cands:
        if c and os.path.isfile(c):
            return c
    return None


class FairseqSpeechEncoder_Hubert(nn.Module):
    MODEL2URL = {
        "hubert": "https://dl.fbaipublicfiles.com/hubert/hubert_base_ls960.pt",
        "hubert_base": "https://dl.fbaipublicfiles.com/hubert/hubert_base_ls960.pt",
        "hubert_large_ll60k": "https://dl.fbaipublicfiles.com/hubert/hubert_large_ll60k.pt",
    }
    MODEL_DOWNSAMPLE_RATE = {"hubert": 320, "hubert_base": 320, "hubert_large_ll60k": 320}

    def __init__(self, name: str, pretrained: bool = False, trainable: bool = False, device: str = "cpu",
                 feat_select_idx: Union[str, list] = "all", layer_drop: Union[str, float] = 0.0, max_audio_len: int = -1,
                 reinit_layers: List[int] = [], unfreeze_layers: List[int] = [], normalize_hiddenstates: bool = False,
                 normalize_type: str = "s3prl", hubert_config: HubertConfig = None, **kwargs):
        super().__init__()
        assert name in self.MODEL2URL, "Model name({}) should be in {}".format(name, self.MODEL2URL.keys())
        assert normalize_type in ["s3prl", "method1", "method2"], normalize_type
        assert not (len(reinit_layers) > 0 and len(unfreeze_layers) > 0)            # speech_encoder_plus.py:415
        if (len(reinit_layers) > 0 or len(unfreeze_layers) > 0) and not trainable:
            raise AssertionError("reinit_layers / unfreeze_layers need trainable=True (speech_encoder_plus.py:418,:433)")
        if not (layer_drop == "original" or (isinstance(layer_drop, float) and 0.0 <= layer_drop <= 1.0)):
            raise ValueError(f"layer_drop = {layer_drop} is not supported.")
        self.name, self.pretrained, self.trainable = name, pretrained, trainable
        self.feat_select_idx, self.max_audio_len = feat_select_idx, max_audio_len
        self.reinit_layers, self.unfreeze_layers = reinit_layers, unfreeze_layers
        self.normalize_hiddenstates, self.normalize_type = normalize_hiddenstates, normalize_type
        if hubert_config is not None and not isinstance(hubert_config, HubertConfig):   # dict / OrderedNamespace from a YAML
            hc = dict(hubert_config.to_dict() if hasattr(hubert_config, "to_dict") else hubert_config)
            if "conv_layers" in hc:
                hc["conv_layers"] = [tuple(c) for c in hc["conv_layers"]]
            hubert_config = HubertConfig(**hc)
        cfg = hubert_config if hubert_config is not None else HubertConfig.from_name(name)
        ckpt_sd = None
        if pretrained:
            ckpt = _find_local_ckpt(self.MODEL2URL[name])
            if ckpt is None:
                logger.warning("pretrained=True but no local HuBERT checkpoint (set SPEECHCLIP_HUBERT_CKPT); using random init")
            else:
                # speech_encoder_plus.py:380-398: fairseq.checkpoint_utils.load_model_ensemble_and_task builds the model FROM THE CHECKPOINT'S cfg
                # (extractor_mode, conv_bias, layer_norm_first ...; task.cfg.normalize decides the wave layer-norm, :507-508), not from its name.
                # fairseq itself is not needed: util/checkpoint_io.py unpickles the file with inert stubs for fairseq's classes.
                from ..util.checkpoint_io import load_fairseq_hubert
                file_cfg, ckpt_sd, stubbed = load_fairseq_hubert(ckpt)
                if hubert_config is not None and file_cfg != cfg:
                    raise ValueError(f"{ckpt}: the checkpoint's own configuration {file_cfg} differs from the hubert_config that was passed in {cfg}")
                if file_cfg != cfg:
                    logger.info(f"{ckpt}: architecture read from the checkpoint's cfg: {file_cfg} (name `{name}` alone would give {cfg})")
                cfg = file_cfg
                logger.info(f"{ckpt}: {len(ckpt_sd)} tensors; {len(stubbed)} pickled classes replaced by inert stubs")
        self.encoder = HubertModel(cfg)
        self.encoder_task = type("Task", (), {"cfg": type("Cfg", (), {"normalize": cfg.normalize})()})()
        if ckpt_sd is not None:
            from ..util.checkpoint_io import HUBERT_UNUSED_KEYS, strict_load
            # every weight of the forward must be in the file; the pre-training head (final_proj, label_embs_concat) and mask_emb are never reached
            # by customHubertForward (speech_encoder_plus.py:67-107) and are the only keys that may be absent on either side
            strict_load(self.encoder, ckpt_sd, allow_missing=HUBERT_UNUSED_KEYS, allow_unexpected=HUBERT_UNUSED_KEYS, what=f"HuBERT checkpoint {ckpt}")
        if layer_drop != "original":          # speech_encoder_plus.py:405-412: a float overrides the checkpoint's rate, "original" keeps it
            self.encoder.encoder.layerdrop = float(layer_drop)
        for p in self.encoder.parameters():
            p.requires_grad = False
        self.encoder.eval()
        # speech_encoder_plus.py:416-446: the listed transformer layers train (reinit_layers: re-initialised first, as `layer.apply(init_weights)` does);
        # every other layer, pos_conv, layer_norm, the feature extractor and post_extract_proj stay frozen (feature_grad_mult = 0)
        # KNOWN DIVERGENCE (documented in EXPERIMENTS.md, old section 5c): the reference freezes the unlisted layers, pos_conv, the FEATURE
        # LayerNorm (`encoder.layer_norm`), the extractor and post_extract_proj -- but not `encoder.encoder.layer_norm`, the LayerNorm behind the
        # positional conv of post-LN HuBERT-base, nor the never-reached mask_emb / final_proj / label_embs_concat.  That LayerNorm sits below
        # every layer, so the reference back-propagates through ALL frozen layers just to train its 2 x 768 numbers.  Here it stays frozen and
        # the backward stops at the lowest listed layer (`HubertLayersTrainFn` gets a detached input): trainable set = listed layers + mix + head.
        self.train_layers = sorted(set(int(i) for i in (list(reinit_layers) + list(unfreeze_layers))))
        # speech_encoder_plus.py:399-401: trainable without layer lists -- nothing is frozen: the conv feature extractor (its gradient scaled by
        # the checkpoint's feature_grad_mult [3P fairseq forward_features]), layer_norm, post_extract_proj, the positional conv and all layers train
        self.train_front = bool(trainable) and not self.train_layers
        if self.train_front:
            base_arch = not cfg.layer_norm_first and cfg.extractor_mode == "default" and not cfg.conv_bias
            large_arch = cfg.layer_norm_first and cfg.extractor_mode == "layer_norm" and cfg.conv_bias
            if not (base_arch or large_arch):
                raise NotImplementedError("trainable=True is built for the two released architectures: GroupNorm extractor + post-LN layers (base) and "
                                          "LayerNorm extractor with conv biases + pre-LN layers (large); not for other combinations")
            self.train_layers = list(range(cfg.encoder_layers))
            # never reached by customHubertForward: torch's Adam skips their None gradients (pre-LN: encoder.layer_norm only touches the final x)
            unused = ("mask_emb", "final_proj", "label_embs_concat") + (("encoder.layer_norm",) if large_arch else ())
            for k, p in self.encoder.named_parameters():
                p.requires_grad = not k.startswith(unused)
            self.encoder.feature_grad_mult = cfg.feature_grad_mult
        elif self.train_layers:
            assert 0 <= self.train_layers[0] and self.train_layers[-1] < cfg.encoder_layers, self.train_layers
            for i in self.train_layers:
                lyr = self.encoder.encoder.layers[i]
                if i in reinit_layers:
                    # `layer.apply(init_weights)` (speech_encoder_plus.py:421, avssl/util/init_model.py): reset_parameters() of every sub-module,
                    # children first -- torch's Linear / LayerNorm defaults -- then [3P fairseq] MultiheadAttention.reset_parameters on top of its
                    # projections: xavier_uniform with gain 1/sqrt(2) on q / k / v, xavier_uniform on out_proj, out_proj.bias = 0 (the q / k / v
                    # biases keep torch's uniform init)
                    for m in lyr.modules():
                        if isinstance(m, (nn.Linear, nn.LayerNorm)):
                            m.reset_parameters()
                    a = lyr.self_attn
                    for proj in (a.k_proj, a.v_proj, a.q_proj):
                        nn.init.xavier_uniform_(proj.weight, gain=1 / math.sqrt(2))
                    nn.init.xavier_uniform_(a.out_proj.weight)
                    nn.init.constant_(a.out_proj.bias, 0.0)
                for p in lyr.parameters():
                    p.requires_grad = True
            self.encoder.feature_grad_mult = 0
        if self.train_front:
            assert 0.0 < float(self.encoder.feature_grad_mult) <= 1.0, "feature_grad_mult = 0 would freeze the extractor: use unfreeze_layers"
        self.downsample_rate = self.MODEL_DOWNSAMPLE_RATE[name]
        self.out_dim = cfg.encoder_embed_dim
        self.upstream_model_hiddenstates_len = cfg.encoder_layers + 1
        if self.feat_select_idx == FEAT_SELECT_IDX_WEIGHTED_SUM_MODE:
            self.weightedsum_layer = WeightedSumLayer(n_weights=self.upstream_model_hiddenstates_len,
                                                      normalize_features=self.normalize_hiddenstates and self.normalize_type == "s3prl")

    def trainable_params(self) -> list:
        params = [p for p in self.encoder.parameters() if p.requires_grad]            # speech_encoder_plus.py:636-648: encoder parameters when trainable
        if self.feat_select_idx == FEAT_SELECT_IDX_WEIGHTED_SUM_MODE:
            params += list(self.weightedsum_layer.parameters())
        return params

    @staticmethod
    def _to_list(wav, wav_len):
        if isinstance(wav, torch.Tensor):
            if wav.dim() == 2:
                if len(wav_len) > 0:
                    return [wav[b, : int(wav_len[b])] for b in range(len(wav))]
                return [wav[b] for b in range(len(wav))]
            if wav.dim() == 1:
                return [wav]
        return list(wav)

    def forward(self, wav: Union[torch.Tensor, list], wav_len: Union[torch.Tensor, list] = [],
                feat_select_idx: Union[str, list] = None, return_hidden_states: bool = False) -> Tuple:
        dev = next(self.encoder.parameters()).device
        # fast path: a padded [B, Lmax] device tensor in eval mode needs no per-utterance slicing
        if isinstance(wav, torch.Tensor) and wav.dim() == 2 and len(wav_len) > 0 and not self.training:
            lens = [int(l) for l in (wav_len.tolist() if torch.is_tensor(wav_len) else wav_len)]
            lmax = max(lens)
            padded = wav[:, :lmax].to(dev, torch.float32)
            # the reference rebuilds the batch from wav[b, :len] with zero right-padding: enforce the zeros
            if any(l < lmax for l in lens):
                keep = torch.arange(lmax, device=dev)[None, :] < ops.dev_ints(lens, torch.int64, dev)[:, None]
                padded = padded * keep
            padded = padded.contiguous()
        elif isinstance(wav, torch.Tensor) and wav.dim() == 2 and len(wav_len) > 0 and self.training and wav.is_cuda:
            # train mode on a padded device batch: the reference crops every utterance longer than max_audio_len at a random offset
            # (np.random.randint, one draw per cropped utterance, in batch order -- reproduced here) and re-pads; one kernel for the batch
            full = [min(int(l), wav.shape[1]) for l in (wav_len.tolist() if torch.is_tensor(wav_len) else wav_len)]
            starts, lens = [], []
            for n in full:
                if self.max_audio_len < 0 or n <= self.max_audio_len:
                    starts.append(0); lens.append(n)
                else:
                    starts.append(int(np.random.randint(n - self.max_audio_len))); lens.append(self.max_audio_len)
            padded = ops.crop_pad(wav.to(dev, torch.float32), starts, lens, max(lens))
        else:
            wavs = self._to_list(wav, wav_len)
            if self.training:
                wavs = [random_crop_max_length(w, self.max_audio_len, len(w)) for w in wavs]
            lens = [len(w) for w in wavs]
            padded = torch.zeros(len(wavs), max(lens), device=dev, dtype=torch.float32)
            for i, w in enumerate(wavs):
                padded[i, : lens[i]] = w.to(dev, torch.float32)
        # speech_encoder_plus.py:572-592: `normalize_type` method1 / method2 overwrite every hidden state before they are mixed / returned ("s3prl" instead
        # normalises inside WeightedSumLayer).  They need the reference's padded [B, T] layout (method2 averages over ALL T frames of the padded batch).
        norm_method = self.normalize_type if (self.normalize_hiddenstates and self.normalize_type.startswith("method")) else None
        if feat_select_idx is None:
            feat_select_idx = self.feat_select_idx
        # layerdrop (speech_encoder_plus.py:49-53): ONE np.random.random() per layer per forward, in layer order and in every mode (the
        # reference draws before it looks at self.training, so the draws also advance the stream the random crops use); a layer is skipped
        # when the module is in train mode and its draw is <= layerdrop -- and then contributes no hidden state.
        nl_ = self.encoder.cfg.encoder_layers
        draws = np.random.random(nl_)
        rate = float(self.encoder.encoder.layerdrop)
        drop = tuple(i for i in range(nl_) if self.training and not (draws[i] > rate))
        # train mode of the module = train mode of the frozen encoder's dropouts too (Lightning's model.train(); speech_encoder_plus.py:42, :87 and
        # the fairseq layers' dropout modules).  One seed per forward from torch's generator; SC_FROZEN_DROPOUT=0 keeps the eval arithmetic.
        drop_seed = None
        if self.training and os.environ.get("SC_FROZEN_DROPOUT", "1") != "0" and any(v > 0 for v in self.encoder.dropout_rates().values()) \
                and not self.encoder.cfg.layer_norm_first:
            drop_seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        if drop and feat_select_idx == FEAT_SELECT_IDX_WEIGHTED_SUM_MODE:
            # WeightedSumLayer.forward asserts one weight per hidden state (weighted_sum.py:36): the reference fails here too
            raise AssertionError(self.upstream_model_hiddenstates_len - len(drop))
        if self.train_layers and torch.is_grad_enabled():
            if norm_method is not None:
                raise NotImplementedError("fine-tuning the encoder under normalize_type method1 / method2 is not built (no shipped config uses them)")
            return self._forward_finetune(padded, lens, feat_select_idx, return_hidden_states, drop_seed)
        pack = self._pack_plan(padded, lens) if norm_method is None else None
        hidden, T, Tp, _valid = self.encoder.extract_all_layers(padded, lens, drop_layers=drop, dropout_seed=drop_seed, pack=pack)      # [n, B, Tp, d]
        if norm_method is not None:
            ops.hidden_normalize_(hidden, T, norm_method)           # in place on this forward's states, as the reference overwrites layer_results[i]
        # speech_encoder_plus.py:604-611: Python round() (banker's) of len / 320, clamped to T
        feat_len = ops.dev_ints([min(round(l / self.downsample_rate), T) for l in lens], torch.long, dev).clone()   # escapes to the caller
        if pack is not None:
            # Padding-free engine (module/hubert.py: packed_geometry): `hidden` is [n, sum_b rows_b, d].  The padded [B, T, d] layout of the
            # reference is restored at this boundary only -- for the mixed frames when nobody needs the states themselves, else for all states.
            # Rows beyond an utterance's own frames (the reference holds padded-frame outputs there, which the heads mask: :604-611) are zeros,
            # the engine's halo row (inexact: it reads the neighbouring utterance) included.
            off = ops.dev_ints(pack["row_off"], torch.int32, dev)
            ws = getattr(self, "weightedsum_layer", None)
            if (feat_select_idx == FEAT_SELECT_IDX_WEIGHTED_SUM_MODE and not return_hidden_states
                    and not (torch.is_grad_enabled() and ws.weights.requires_grad)):
                assert hidden.shape[0] == ws.n_weights, hidden.shape[0]
                mixed = ops.unpack_rows(ops.weighted_sum(hidden, ws.weights.detach().float(), ws.normalize_features), off, padded.shape[0], T, halo=1)
                return (mixed, feat_len)
            hidden = ops.unpack_rows(hidden, off, padded.shape[0], T, halo=1)                                                  # [n, B, T, d]
        escapes = return_hidden_states or feat_select_idx != FEAT_SELECT_IDX_WEIGHTED_SUM_MODE
        if pack is None and escapes:
            # the engine's `hidden` is a reused workspace: states handed to the CALLER (hidden_states / last_hidden_state / a layer list) are copied
            # out, as the reference returns fresh tensors (speech_encoder_plus.py:596-602) -- a second forward would silently rewrite them otherwise
            hidden = hidden.clone()
        layers = lambda: tuple(hidden[i, :, :T] for i in range(hidden.shape[0]))  # noqa: E731
        out = []
        if feat_select_idx == "all":
            hs = layers()
            out.extend([{"last_hidden_state": hs[-1], "hidden_states": hs}, feat_len])
        elif feat_select_idx == FEAT_SELECT_IDX_WEIGHTED_SUM_MODE:
            mixed = self.weightedsum_layer(hidden)[:, :T]
            if torch.is_grad_enabled() and self.weightedsum_layer.weights.requires_grad:
                # training of the tail (speechclip_amd/train_tail.py): the branch's backward produces the gradient of the mix weights
                # in the same pass over the frames, so it needs the frozen states the frames were mixed from
                mixed._mix_src = (hidden, self.weightedsum_layer)
            out.extend([mixed, feat_len])
        elif isinstance(feat_select_idx, list):
            hs = layers()
            out.extend([[hs[i] for i in feat_select_idx], feat_len])
        elif feat_select_idx == "last_hidden_state":
            out.extend([hidden[-1, :, :T], feat_len])
        elif feat_select_idx == "hidden_states":
            out.extend([layers(), feat_len])
        else:
            raise KeyError(feat_select_idx)
        if return_hidden_states:
            out.append(layers())
        return tuple(out)


def _pack_plan(self, padded, lens):
    """Packed (padding-free) geometry of this batch, or None for the uniform layout.  SC_VARLEN_PACK: "auto" (default: pack when it removes at
    least 10 % of the transformer rows), "1" (always, when the kernels cover the shape), "0" (never)."""
    mode = os.environ.get("SC_VARLEN_PACK", "auto")
    cfg = self.encoder.cfg
    cg = cfg.encoder_embed_dim // cfg.conv_pos_groups
    if mode == "0" or padded.shape[0] < 1 or cg not in (32, 48, 64) or cfg.conv_layers[0][0] % 64 or cfg.encoder_embed_dim // cfg.encoder_attention_heads != 64:
        return None
    T = self.encoder.frame_geometry(padded.shape[1])[1]
    pack = self.encoder.packed_geometry(lens, padded.shape[1], need_rows=[min(round(l / self.downsample_rate), T) for l in lens])
    if mode != "1" and pack["total"] > 0.9 * pack["padded_rows"]:
        return None
    return pack


FairseqSpeechEncoder_Hubert._pack_plan = _pack_plan


def _forward_finetune(self, padded, lens, feat_select_idx, return_hidden_states, drop_seed=None):
    """Training forward with encoder layers L0.. as ONE autograd node (train_hubert.HubertLayersTrainFn): the frozen part below the lowest
    trainable layer runs on the eval path, the layer mix carries the gradient to the hidden states (WeightedSumTrainFn)."""
    from ..train_hubert import HubertLayersTrainFn, WeightedSumTrainFn, layer_params
    if feat_select_idx != FEAT_SELECT_IDX_WEIGHTED_SUM_MODE:
        raise NotImplementedError("fine-tuning is wired for feat_select_idx = weighted_sum (every shipped config)")
    enc = self.encoder
    cfg = enc.cfg
    dev = padded.device
    L0, nl = self.train_layers[0], cfg.encoder_layers
    B, d = padded.shape[0], cfg.encoder_embed_dim
    if self.train_front:          # the wave -> hidden state 0 as an autograd node too (train_front.HubertFront[LN]TrainFn)
        from ..train_front import HubertFrontLNTrainFn, HubertFrontTrainFn, front_params, front_params_ln
        T0, T, P0, Tp = enc.frame_geometry(padded.shape[1])
        valid = enc.valid_frames(lens, padded.shape[1], T)
        fmeta = dict(conv_layers=[tuple(c) for c in cfg.conv_layers], T0=T0, P0=P0, Tp=Tp, d=d, G=cfg.conv_pos_groups, Kw=cfg.conv_pos,
                     grad_mult=float(enc.feature_grad_mult), normalize=bool(cfg.normalize))
        valid_dev = ops.dev_ints(valid, torch.int32, dev)
        if cfg.layer_norm_first:
            h_front = HubertFrontLNTrainFn.apply(fmeta, padded.contiguous(), ops.dev_ints(lens, torch.int32, dev), valid_dev, *front_params_ln(enc))
        else:
            if drop_seed is not None:
                r = enc.dropout_rates()
                fmeta["drop"] = dict(features=r["features"], hidden=r["hidden"], seed=int(drop_seed))
            h_front = HubertFrontTrainFn.apply(fmeta, padded.contiguous(), valid_dev, *front_params(enc))      # [B*Tp, d]
        hidden = None
    else:
        # (train-mode dropouts: the frozen layers below L0 through the engine, the trained nodes through their own masks)
        hidden, T, Tp, valid = enc.extract_all_layers(padded, lens, stop_layer=L0, dropout_seed=drop_seed)        # hidden[0 .. L0] are valid
    M = B * Tp
    params = []
    for i in range(L0, nl):
        params += layer_params(enc.encoder.layers[i])
    meta = dict(B=B, Tp=Tp, H=cfg.encoder_attention_heads, eps=1e-5, train=[i in self.train_layers for i in range(L0, nl)],
                pre_ln=bool(cfg.layer_norm_first))
    if drop_seed is not None:
        r = enc.dropout_rates()
        meta["drop"] = dict(hidden=r["hidden"], attention=r["attention"], activation=r["activation"], seed=(int(drop_seed) * 2654435761 + 97) & 0x7fffffff)
    # (the engine's hidden buffer is a reused workspace: the autograd node keeps its own copy)
    h_in = h_front if self.train_front else hidden[L0].reshape(M, d).clone()
    hi = HubertLayersTrainFn.apply(meta, h_in, ops.dev_ints(valid, torch.int32, dev), *params)      # [nl - L0, M, d]
    below = h_front.view(1, M, d) if self.train_front else hidden[:L0 + 1].reshape(L0 + 1, M, d).detach()
    hidden_all = torch.cat([below, hi], 0)
    ws = self.weightedsum_layer
    mixed = WeightedSumTrainFn.apply(hidden_all, ws.weights, ws.normalize_features).view(B, Tp, d)[:, :T]
    mixed._mix_src = (hidden_all.detach().view(nl + 1, B, Tp, d), ws)                  # the mix weights' gradient comes out of the head's backward
    feat_len = ops.dev_ints([min(round(l / self.downsample_rate), T) for l in lens], torch.long, dev).clone()
    out = [mixed, feat_len]
    if return_hidden_states:
        out.append(tuple(hidden_all[i].view(B, Tp, d)[:, :T] for i in range(nl + 1)))
    return tuple(out)


FairseqSpeechEncoder_Hubert._forward_finetune = _forward_finetune


class S3prlSpeechEncoderPlus(nn.Module):
    """Alternative loader via the s3prl hub (avssl/module/speech_encoder_plus.py:110-316).  No shipped config uses it
    (all have `type: FairseqHubert`) and s3prl is a network download; kept as an importable name only."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError("S3prlSpeechEncoderPlus is out of scope on MI355X; use audio_encoder.type: FairseqHubert")

"""ClipModel -- the reference's CLIP wrapper (avssl/module/clip_official.py:26-294) over the MI355X CLIP engine.

Same constructor arguments and attributes (`model`, `out_dim`, `tokenizer`, `device`, reduced sub-word vocabulary,
`encode_image`, `encode_text`, `encode_keywords`, `update_device`, `trainable_params`).  openai `clip` is not
required: `self.model` is a parameter tree with openai's key names (clip_model.CLIP) whose towers run on HIP kernels.
No network: weights are loaded only from a local state_dict ($SPEECHCLIP_CLIP_CKPT), else random init.
"""
import logging
import os

import numpy as np
import torch
import torch.nn as nn

from .clip_model import CLIP, ClipConfig

logger = logging.getLogger(__name__)
_clip_models = {"RN50", "RN101", "RN50x4", "RN50x16", "RN50x64", "ViT-B/32", "ViT-B/16", "ViT-L/14"}
SOT_ID, EOT_ID = 49406, 49407


class _IdDecoder(dict):
    """decoder[i] of the id-literal tokenizer: "<i></w>" for every id."""

    def __missing__(self, i):
        return "<{}></w>".format(int(i))


class _TokenizerIds:
    """Stand-in for openai's SimpleTokenizer (clip/simple_tokenizer.py [3P], not installed and its BPE vocabulary file is not
    available offline).  The hot path needs two ids (clip_official.py:95-101,235-240); the analysis surface (kwClip.py:277-466,
    :918-1001) needs `decoder[id]`, `decode(ids)` and `encode(text)`.  Without the vocabulary, sub-word i prints as the literal "<i>":
    decode / encode are exact inverses of each other on id lists, so hit rates and neighbour lists are computed on the true token ids and
    only their printed form differs.  Assign a real SimpleTokenizer-compatible object to `ClipModel.tokenizer` to get words."""

    def __init__(self, vocab_size=49408):
        self.encoder = {"<|startoftext|>": vocab_size - 2, "<|endoftext|>": vocab_size - 1}
        self.decoder = _IdDecoder()

    def decode(self, tokens) -> str:
        return "".join(self.decoder[t] for t in tokens).replace("</w>", " ")

    def encode(self, text: str) -> list:
        return [int(w[1:-1]) for w in text.split() if w.startswith("<") and w.endswith(">") and w[1:-1].isdigit()]


class ClipModel(nn.Module):
    def __init__(self, name: str, device: str = "cpu", image_encoder_trainable: bool = False, text_encoder_trainable: bool = False,
                 reduce_subword_embbedding: str = None, clip_config: ClipConfig = None, **kwargs):
        super().__init__()
        assert name in _clip_models
        if image_encoder_trainable or text_encoder_trainable:
            raise NotImplementedError("CLIP towers are frozen in every shipped config; fine-tuning needs backward kernels")
        self.name, self.device = name, device
        if clip_config is not None and not isinstance(clip_config, ClipConfig):
            clip_config = ClipConfig(**dict(clip_config.to_dict() if hasattr(clip_config, "to_dict") else clip_config))
        cfg = clip_config if clip_config is not None else ClipConfig.from_name(name)
        ckpt = os.environ.get("SPEECHCLIP_CLIP_CKPT", "")
        ckpt_sd = None
        if ckpt and os.path.isfile(ckpt):
            # clip_official.py:50 `clip.load(name, device)`: openai's files are TorchScript archives; read without the `clip` package, fp16 -> fp32 as
            # clip.load does on the CPU, architecture derived from the tensor shapes as clip.model.build_model does -- and it must be the NAMED one
            from ..util.checkpoint_io import clip_config_from_state_dict, load_clip_state_dict
            ckpt_sd = load_clip_state_dict(ckpt)
            file_cfg = clip_config_from_state_dict(ckpt_sd)
            if file_cfg != cfg:
                raise ValueError(f"{ckpt} holds {file_cfg}, but clip.name = {name!r} is {cfg}")
        elif ckpt:
            raise FileNotFoundError(f"SPEECHCLIP_CLIP_CKPT={ckpt} does not exist")
        self.model = CLIP(cfg)
        if ckpt_sd is not None:
            from ..util.checkpoint_io import CLIP_NON_WEIGHT_KEYS, strict_load
            strict_load(self.model, ckpt_sd, allow_unexpected=CLIP_NON_WEIGHT_KEYS, what=f"CLIP checkpoint {ckpt}")
        self.image_encoder_trainable, self.text_encoder_trainable = image_encoder_trainable, text_encoder_trainable
        self.out_dim = self.model.transformer.width
        from ..data.image_transforms import clip_preprocess
        self.image_preprocess = clip_preprocess(cfg.image_resolution)        # clip_official.py:50: the `preprocess` clip.load returns
        self.tokenizer = _TokenizerIds(cfg.vocab_size)
        for p in self.model.parameters():
            p.requires_grad = False
        self.selected_text_emb_ids = None
        if reduce_subword_embbedding is not None:
            if not os.path.exists(reduce_subword_embbedding):
                raise FileNotFoundError(reduce_subword_embbedding)
            data = np.load(reduce_subword_embbedding)
            self.selected_text_emb_ids = data[:, 0]
            dist = data[:, 1]
            self.selected_text_emb_ids_dist = torch.from_numpy(dist / np.sum(dist))
            self.original_text_emb_weight = self.model.token_embedding.weight
            reduced = self.model.token_embedding.weight[torch.as_tensor(self.selected_text_emb_ids, dtype=torch.long)]
            self.model.token_embedding = nn.Embedding.from_pretrained(reduced.detach().clone())
            self.model.token_embedding.weight.requires_grad = False
            self.original2Reduced = {int(o): n for n, o in enumerate(self.selected_text_emb_ids)}
            self.reducedl2Original = {n: int(o) for n, o in enumerate(self.selected_text_emb_ids)}
            self.startOfTxt_reduced = self.original2Reduced[self.tokenizer.encoder["<|startoftext|>"]]
            self.endOfTxt_reduced = self.original2Reduced[self.tokenizer.encoder["<|endoftext|>"]]

    def trainable_params(self) -> list:
        return []

    def update_device(self, device):
        self.device = device

    def to(self, *args, **kwargs):
        super().to(*args, **kwargs)
        self.device = self.model.token_embedding.weight.device
        return self

    def prep_image(self, paths: list) -> torch.Tensor:
        """clip_official.py:151-164: image files -> pre-processed tensor [B, 3, H, W] on self.device.  PIL open + bicubic resize + centre crop on the
        host (data/image_transforms.py), then the batch goes to the device as uint8 and is scaled / normalised there (sc_image_normalize_u8)."""
        from ..data.image_transforms import load_images_u8, normalize_u8
        return normalize_u8(load_images_u8(paths, self.model.cfg.image_resolution), self.device)

    def encode_image(self, image: torch.Tensor) -> torch.Tensor:
        return self.model.encode_image(image)

    def _special_ids(self):
        if self.selected_text_emb_ids is None:
            return self.tokenizer.encoder["<|startoftext|>"], self.tokenizer.encoder["<|endoftext|>"]
        return self.startOfTxt_reduced, self.endOfTxt_reduced

    def encode_text(self, text: torch.Tensor) -> torch.Tensor:
        """text: [B, 77] token ids (reduced ids if the vocabulary is reduced).  EOT = arg-max id position."""
        emb = self.model.token_embedding(text)
        return self.model.encode_text_embeddings(emb, text.argmax(dim=-1))

    def encode_keywords(self, keywords: torch.Tensor, keyword_num: int) -> torch.Tensor:
        """clip_official.py:220-264: [SOT, kw_1..kw_K, EOT, pad...] through the text tower, feature at position K+1.
        The causal mask makes positions > K+1 irrelevant, so only K+2 positions are evaluated."""
        if not isinstance(keywords, torch.Tensor):
            raise TypeError(f"Unknown keywords type {type(keywords)}")
        B, dev = keywords.size(0), keywords.device
        sot, eot = self._special_ids()
        tok = self.model.token_embedding.weight
        emb = torch.empty(B, keyword_num + 2, tok.shape[1], device=dev, dtype=torch.float32)
        emb[:, 0] = tok[sot]
        emb[:, 1:1 + keyword_num] = keywords
        emb[:, 1 + keyword_num] = tok[eot]
        return self.model.encode_text_embeddings(emb, keyword_num + 1)       # one EOT position for the batch: a host int, no gather / sync

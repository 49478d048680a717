"""fp32 CPU restatement of openai/CLIP's ViT image tower and text transformer (test oracle).

Published algorithm of openai/CLIP `clip/model.py` (requirements.txt:4 of the reference,
unpinned HEAD; sources NOT under /root/reference).  Parameter names follow openai's so the
reference's checkpoint keys `clip.model.*` load unchanged.  Call sites in
the reference: avssl/module/clip_official.py:50-55 (clip.load, attributes used),
:200-209 (encode_image), :220-264 (encode_keywords).
"""
from collections import OrderedDict
from dataclasses import dataclass

import torch
import torch.nn as nn


@dataclass
class ClipRefConfig:
    image_resolution: int = 224
    vision_patch: int = 32
    vision_width: int = 768
    vision_layers: int = 12
    embed_dim: int = 512
    context_length: int = 77
    vocab_size: int = 49408
    text_width: int = 512
    text_heads: int = 8
    text_layers: int = 12

    @staticmethod
    def vit_b32():
        return ClipRefConfig()

    @staticmethod
    def vit_l14():
        return ClipRefConfig(vision_patch=14, vision_width=1024, vision_layers=24, embed_dim=768,
                             text_width=768, text_heads=12)

    @staticmethod
    def tiny():
        return ClipRefConfig(image_resolution=64, vision_patch=16, vision_width=128, vision_layers=2,
                             embed_dim=64, context_length=77, vocab_size=512, text_width=64,
                             text_heads=1, text_layers=2)


class QuickGELU(nn.Module):
    def forward(self, x):
        return x * torch.sigmoid(1.702 * x)


class ResidualAttentionBlock(nn.Module):
    """x = x + MHA(ln_1(x)); x = x + c_proj(QuickGELU(c_fc(ln_2(x))))  (pre-LN)."""

    def __init__(self, d, heads, attn_mask=None):
        super().__init__()
        self.attn = nn.MultiheadAttention(d, heads)
        self.ln_1 = nn.LayerNorm(d)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(d, d * 4)), ("gelu", QuickGELU()),
                                              ("c_proj", nn.Linear(d * 4, d))]))
        self.ln_2 = nn.LayerNorm(d)
        self.attn_mask = attn_mask

    def forward(self, x):  # [L, N, D]
        m = self.attn_mask.to(dtype=x.dtype, device=x.device) if self.attn_mask is not None else None
        y = self.ln_1(x)
        x = x + self.attn(y, y, y, need_weights=False, attn_mask=m)[0]
        return x + self.mlp(self.ln_2(x))


class Transformer(nn.Module):
    def __init__(self, width, layers, heads, attn_mask=None):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads, attn_mask) for _ in range(layers)])

    def forward(self, x):
        return self.resblocks(x)


class VisionTransformer(nn.Module):
    def __init__(self, input_resolution, patch_size, width, layers, heads, output_dim):
        super().__init__()
        self.input_resolution, self.output_dim = input_resolution, output_dim
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch_size, stride=patch_size, bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn((input_resolution // patch_size) ** 2 + 1, width))
        self.ln_pre = nn.LayerNorm(width)
        self.transformer = Transformer(width, layers, heads)
        self.ln_post = nn.LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))

    def forward(self, x):  # [B,3,H,W]
        x = self.conv1(x)
        x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)           # [B, g*g, w]
        cls = self.class_embedding.to(x.dtype) + torch.zeros(x.shape[0], 1, x.shape[-1], dtype=x.dtype)
        x = torch.cat([cls, x], dim=1) + self.positional_embedding.to(x.dtype)
        x = self.ln_pre(x)
        x = self.transformer(x.permute(1, 0, 2)).permute(1, 0, 2)
        x = self.ln_post(x[:, 0, :])
        return x @ self.proj


class ClipRef(nn.Module):
    def __init__(self, cfg: ClipRefConfig):
        super().__init__()
        self.cfg = cfg
        self.context_length = cfg.context_length
        self.visual = VisionTransformer(cfg.image_resolution, cfg.vision_patch, cfg.vision_width,
                                        cfg.vision_layers, cfg.vision_width // 64 if cfg.vision_width >= 64 else 1,
                                        cfg.embed_dim)
        mask = torch.full((cfg.context_length, cfg.context_length), float("-inf")).triu_(1)
        self.transformer = Transformer(cfg.text_width, cfg.text_layers, cfg.text_heads, attn_mask=mask)
        self.vocab_size = cfg.vocab_size
        self.token_embedding = nn.Embedding(cfg.vocab_size, cfg.text_width)
        self.positional_embedding = nn.Parameter(torch.empty(cfg.context_length, cfg.text_width))
        self.ln_final = nn.LayerNorm(cfg.text_width)
        self.text_projection = nn.Parameter(torch.empty(cfg.text_width, cfg.embed_dim))
        self.logit_scale = nn.Parameter(torch.ones([]) * 2.6592600)
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.normal_(self.positional_embedding, std=0.01)
        nn.init.normal_(self.text_projection, std=cfg.text_width ** -0.5)
        self._init_towers()

    def _init_towers(self):
        for tower in (self.visual.transformer, self.transformer):
            proj_std = (tower.width ** -0.5) * ((2 * tower.layers) ** -0.5)
            attn_std = tower.width ** -0.5
            fc_std = (2 * tower.width) ** -0.5
            for blk in tower.resblocks:
                nn.init.normal_(blk.attn.in_proj_weight, std=attn_std)
                nn.init.normal_(blk.attn.out_proj.weight, std=proj_std)
                nn.init.normal_(blk.mlp.c_fc.weight, std=fc_std)
                nn.init.normal_(blk.mlp.c_proj.weight, std=proj_std)

    @property
    def dtype(self):
        return self.visual.conv1.weight.dtype

    def encode_image(self, image):
        return self.visual(image.type(self.dtype))

    def encode_text(self, text):
        x = self.token_embedding(text).type(self.dtype) + self.positional_embedding.type(self.dtype)
        x = self.transformer(x.permute(1, 0, 2)).permute(1, 0, 2)
        x = self.ln_final(x).type(self.dtype)
        return x[torch.arange(x.shape[0]), text.argmax(dim=-1)] @ self.text_projection

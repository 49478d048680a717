"""CLIP (ViT image tower + text transformer) on MI355X: openai checkpoint key names + HIP forward.

Replaces what the reference obtains from `clip.load(name, device)` (openai/CLIP, not vendored) and
calls at avssl/module/clip_official.py:209 (`model.encode_image`), :218 (`encode_text`) and
:249-262 (text transformer over injected keyword embeddings).  Parameter names follow openai's
`clip/model.py` so `clip.model.*` checkpoint keys load by name.

The image tower keeps an fp32 residual stream (pre-LN blocks; M = B*50 rows is tiny) and runs every
dense contraction through the bf16 MFMA GEMM; LayerNorms compute in fp32 like CLIP's LayerNorm.
"""
from dataclasses import dataclass

import torch
import torch.nn as nn

from .. import ops
from ..ops import ACT_QUICKGELU


@dataclass
class ClipConfig:
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
    def from_name(name: str) -> "ClipConfig":
        if name == "ViT-B/32":
            return ClipConfig()
        if name == "ViT-B/16":
            return ClipConfig(vision_patch=16)
        if name == "ViT-L/14":
            return ClipConfig(vision_patch=14, vision_width=1024, vision_layers=24, embed_dim=768, text_width=768, text_heads=12)
        raise NotImplementedError(f"CLIP variant {name} is not on the MI355X hot path (ViT-B/32, ViT-B/16, ViT-L/14 are)")


class _InProjAttn(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.out_proj = nn.Linear(d, d)


class _Mlp(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.c_fc = nn.Linear(d, 4 * d)
        self.c_proj = nn.Linear(4 * d, d)


class _ResBlock(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.attn = _InProjAttn(d)
        self.ln_1 = nn.LayerNorm(d)
        self.mlp = _Mlp(d)
        self.ln_2 = nn.LayerNorm(d)


class _Tower(nn.Module):
    def __init__(self, width, layers, heads):
        super().__init__()
        self.width, self.layers, self.heads = width, layers, heads
        self.resblocks = nn.ModuleList([_ResBlock(width) for _ in range(layers)])
        proj_std = (width ** -0.5) * ((2 * layers) ** -0.5)
        for blk in self.resblocks:
            nn.init.normal_(blk.attn.in_proj_weight, std=width ** -0.5)
            nn.init.normal_(blk.attn.out_proj.weight, std=proj_std)
            nn.init.normal_(blk.mlp.c_fc.weight, std=(2 * width) ** -0.5)
            nn.init.normal_(blk.mlp.c_proj.weight, std=proj_std)

    def pack(self, dev):
        bf, f32 = torch.bfloat16, torch.float32
        out = []
        for b in self.resblocks:
            out.append(dict(
                ln1=(b.ln_1.weight.detach().to(dev, f32), b.ln_1.bias.detach().to(dev, f32)),
                wqkv=b.attn.in_proj_weight.detach().to(dev, bf).contiguous(), bqkv=b.attn.in_proj_bias.detach().to(dev, f32),
                wo=b.attn.out_proj.weight.detach().to(dev, bf).contiguous(), bo=b.attn.out_proj.bias.detach().to(dev, f32),
                ln2=(b.ln_2.weight.detach().to(dev, f32), b.ln_2.bias.detach().to(dev, f32)),
                w1=b.mlp.c_fc.weight.detach().to(dev, bf).contiguous(), b1=b.mlp.c_fc.bias.detach().to(dev, f32),
                w2=b.mlp.c_proj.weight.detach().to(dev, bf).contiguous(), b2=b.mlp.c_proj.bias.detach().to(dev, f32)))
        return out


def run_tower(packed, x, B, L, heads, causal=False):
    """x: f32 residual stream [B*L, W], updated in place.  Pre-LN blocks with QuickGELU MLP."""
    M, W = x.shape
    dev = x.device
    bf = torch.bfloat16
    n = torch.empty(M, W, device=dev, dtype=bf)
    qkv = torch.empty(M, 3 * W, device=dev, dtype=bf)
    att = torch.empty(M, W, device=dev, dtype=bf)
    ffn = torch.empty(M, 4 * W, device=dev, dtype=bf)
    for L_ in packed:
        ops.layernorm(x, *L_["ln1"], out=n)
        ops.gemm(n, L_["wqkv"], L_["bqkv"], out=qkv)
        ops.attention(qkv, B, L, heads, None, out=att, causal=causal)
        ops.gemm(att, L_["wo"], L_["bo"], residual=x, out=x, out_f32=True)
        ops.layernorm(x, *L_["ln2"], out=n)
        ops.gemm(n, L_["w1"], L_["b1"], ACT_QUICKGELU, out=ffn)
        ops.gemm(ffn, L_["w2"], L_["b2"], residual=x, out=x, out_f32=True)
    return x


class _Visual(nn.Module):
    def __init__(self, cfg: ClipConfig):
        super().__init__()
        w, p, r = cfg.vision_width, cfg.vision_patch, cfg.image_resolution
        self.input_resolution, self.output_dim, self.patch = r, cfg.embed_dim, p
        self.conv1 = nn.Conv2d(3, w, kernel_size=p, stride=p, bias=False)
        scale = w ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(w))
        self.positional_embedding = nn.Parameter(scale * torch.randn((r // p) ** 2 + 1, w))
        self.ln_pre = nn.LayerNorm(w)
        self.transformer = _Tower(w, cfg.vision_layers, max(1, w // 64))
        self.ln_post = nn.LayerNorm(w)
        self.proj = nn.Parameter(scale * torch.randn(w, cfg.embed_dim))


class CLIP(nn.Module):
    """openai-compatible parameter tree.  encode_image / encode_text_embeddings run on HIP kernels."""

    def __init__(self, cfg: ClipConfig):
        super().__init__()
        self.cfg = cfg
        self.context_length = cfg.context_length
        self.visual = _Visual(cfg)
        self.transformer = _Tower(cfg.text_width, cfg.text_layers, cfg.text_heads)
        self.vocab_size = cfg.vocab_size
        self.token_embedding = nn.Embedding(cfg.vocab_size, cfg.text_width)
        self.positional_embedding = nn.Parameter(torch.empty(cfg.context_length, cfg.text_width))
        self.ln_final = nn.LayerNorm(cfg.text_width)
        self.text_projection = nn.Parameter(torch.empty(cfg.text_width, cfg.embed_dim))
        self.logit_scale = nn.Parameter(torch.ones([]) * 2.6592600)
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.normal_(self.positional_embedding, std=0.01)
        nn.init.normal_(self.text_projection, std=cfg.text_width ** -0.5)
        self._packed = None
        self._packed_bwd = None
        self.register_load_state_dict_post_hook(lambda module, keys: module.invalidate_packed())

    def invalidate_packed(self):
        self._packed = None
        self._packed_bwd = None

    def packed_bwd(self, dev):
        """Operands of the text tower's input-gradient pass (train_tail.TextTowerTrainFn): [W^T | W^T] as bf16 [in, 2 out] for dX = dY @ W on
        the MFMA GEMM with hi+lo split gradients (frozen weights, built once), fp32 LayerNorm gains."""
        if getattr(self, "_packed_bwd", None) is None:
            f32, bf = torch.float32, torch.bfloat16
            c = lambda t: t.detach().to(dev, f32).contiguous()          # noqa: E731
            def t16(t):       # [W^T | W^T] bf16 [in, 2*out]: pairs with the (hi | lo) split of the incoming gradient
                wt = t.detach().to(dev, f32).t().contiguous().to(bf)
                return torch.cat([wt, wt], dim=1).contiguous()

            self._packed_bwd = dict(
                txt=[dict(g1=c(b.ln_1.weight), wqkvt=t16(b.attn.in_proj_weight), wot=t16(b.attn.out_proj.weight), g2=c(b.ln_2.weight),
                          w1t=t16(b.mlp.c_fc.weight), w2t=t16(b.mlp.c_proj.weight)) for b in self.transformer.resblocks],
                ln_final_g=c(self.ln_final.weight), txt_proj=t16(self.text_projection.t()))
        return self._packed_bwd

    def _apply(self, fn, *a, **k):
        self._packed = None
        self._packed_bwd = None
        return super()._apply(fn, *a, **k)

    @property
    def dtype(self):
        return self.visual.conv1.weight.dtype

    def _pack(self, dev):
        bf, f32 = torch.bfloat16, torch.float32
        v, cfg = self.visual, self.cfg
        K = 3 * v.patch * v.patch
        Kpad = (K + 63) // 64 * 64
        w = torch.zeros(cfg.vision_width, Kpad, device=dev, dtype=bf)
        w[:, :K] = v.conv1.weight.detach().reshape(cfg.vision_width, K).to(dev, bf)
        P = dict(Kpad=Kpad, conv_w=w, cls=v.class_embedding.detach().to(dev, f32), pos=v.positional_embedding.detach().to(dev, f32).contiguous(),
                 ln_pre=(v.ln_pre.weight.detach().to(dev, f32), v.ln_pre.bias.detach().to(dev, f32)),
                 ln_post=(v.ln_post.weight.detach().to(dev, f32), v.ln_post.bias.detach().to(dev, f32)),
                 proj_t=v.proj.detach().t().to(dev, bf).contiguous(), vis=v.transformer.pack(dev),
                 txt=self.transformer.pack(dev), txt_pos=self.positional_embedding.detach().to(dev, f32).contiguous(),
                 ln_final=(self.ln_final.weight.detach().to(dev, f32), self.ln_final.bias.detach().to(dev, f32)),
                 txt_proj_t=self.text_projection.detach().t().to(dev, bf).contiguous())
        return P

    def packed(self, dev):
        if self._packed is None:
            self._packed = self._pack(dev)
        return self._packed

    @torch.no_grad()
    def encode_image(self, image: torch.Tensor) -> torch.Tensor:
        """[B,3,R,R] f32 (CLIP-normalised) -> [B, embed_dim] f32 (un-normalised), as VisionTransformer.forward."""
        assert image.is_cuda, "CLIP.encode_image runs on the HIP kernels only (no CPU fallback)"
        v = self.visual
        P = self.packed(image.device)
        B = image.shape[0]
        ntok = (v.input_resolution // v.patch) ** 2 + 1
        W = self.cfg.vision_width
        cols = ops.vit_patchify(image.float().contiguous(), v.patch, P["Kpad"])
        patch = ops.gemm(cols, P["conv_w"])
        x = ops.vit_embed(patch, P["cls"], P["pos"], *P["ln_pre"], B, ntok, W)
        run_tower(P["vis"], x, B, ntok, v.transformer.heads)
        cls = ops.layernorm(x, *P["ln_post"], rows=B, D=W, ld_in=ntok * W)
        return ops.gemm(cls, P["proj_t"], out_f32=True)

    def encode_text_embeddings(self, emb: torch.Tensor, take_pos) -> torch.Tensor:
        """emb: f32 [B, L, tw] token embeddings (before positional add), L <= context_length; take_pos: int64 [B] or one int for all rows.  Runs the causal text
        transformer on the first L positions (positions > max(take_pos) cannot influence it) and returns
        ln_final(x)[b, take_pos[b]] @ text_projection as f32 [B, embed_dim].  When `emb` carries a gradient (training the cascaded
        branch through the frozen tower) the differentiable path runs; all rows must then share one take position."""
        assert emb.is_cuda
        if torch.is_grad_enabled() and emb.requires_grad:
            from ..train_tail import TextTowerTrainFn
            if not isinstance(take_pos, int):
                pos = int(take_pos[0].item())
                assert bool((take_pos == pos).all()), "training path: one EOT position for the whole batch (K keywords => K + 1)"
                take_pos = pos
            return TextTowerTrainFn.apply(self, emb, take_pos)
        with torch.no_grad():
            return self._encode_text_embeddings_eval(emb, take_pos)

    def _encode_text_embeddings_eval(self, emb: torch.Tensor, take_pos: torch.Tensor) -> torch.Tensor:
        P = self.packed(emb.device)
        B, L, tw = emb.shape
        x = (emb.float() + P["txt_pos"][:L]).reshape(B * L, tw).contiguous()
        run_tower(P["txt"], x, B, L, self.transformer.heads, causal=True)
        if isinstance(take_pos, int):
            rows = x.view(B, L, tw)[:, take_pos].contiguous()
        else:
            rows = x.view(B, L, tw)[torch.arange(B, device=emb.device), take_pos].contiguous()
        n = ops.layernorm(rows, *P["ln_final"])
        return ops.gemm(n, P["txt_proj_t"], out_f32=True)

"""Pin the oracle's third-party restatements (fairseq HuBERT, openai CLIP) against the
architecture-equivalent `transformers` implementations with weights copied by key mapping
(part iii).  CPU only.

Known divergences (the reference / fairseq behaviour wins, so the cross-check avoids them):
  * HF derives the frame mask from conv-length arithmetic, fairseq from chunk-`all` -> equal-length
    (unpadded) batches only;
  * for the stable-LN (large) variant HF's last hidden state is post-final-LN while the reference's
    layer_results[-1] is pre-final-LN -> compare hidden_states[:-1] plus LN(last).
"""
import pytest
import torch

from oracle.clip_ref import ClipRef, ClipRefConfig
from oracle.hubert_ref import HubertModelRef, HubertRefConfig, hubert_forward, randomize_norm_affine

transformers = pytest.importorskip("transformers")


def _hf_hubert(cfg: HubertRefConfig):
    from transformers import HubertConfig, HubertModel
    hc = HubertConfig(
        hidden_size=cfg.encoder_embed_dim, num_hidden_layers=cfg.encoder_layers,
        num_attention_heads=cfg.encoder_attention_heads, intermediate_size=cfg.encoder_ffn_embed_dim,
        feat_extract_norm="layer" if cfg.extractor_mode == "layer_norm" else "group",
        conv_dim=[c for c, _, _ in cfg.conv_layers], conv_kernel=[k for _, k, _ in cfg.conv_layers],
        conv_stride=[s for _, _, s in cfg.conv_layers], conv_bias=cfg.conv_bias,
        num_conv_pos_embeddings=cfg.conv_pos, num_conv_pos_embedding_groups=cfg.conv_pos_groups,
        do_stable_layer_norm=cfg.layer_norm_first, feat_proj_layer_norm=True, hidden_dropout=0.0,
        attention_dropout=0.0, activation_dropout=0.0, feat_proj_dropout=0.0, layerdrop=0.0,
        apply_spec_augment=False, hidden_act="gelu", layer_norm_eps=1e-5)
    return HubertModel(hc).eval()


def _copy_hubert(ref: HubertModelRef, hf):
    sd = ref.state_dict()
    out = {}
    cfg = ref.cfg
    for i in range(len(cfg.conv_layers)):
        out[f"feature_extractor.conv_layers.{i}.conv.weight"] = sd[f"feature_extractor.conv_layers.{i}.0.weight"]
        if cfg.conv_bias:
            out[f"feature_extractor.conv_layers.{i}.conv.bias"] = sd[f"feature_extractor.conv_layers.{i}.0.bias"]
        if cfg.extractor_mode == "layer_norm":
            out[f"feature_extractor.conv_layers.{i}.layer_norm.weight"] = sd[f"feature_extractor.conv_layers.{i}.2.1.weight"]
            out[f"feature_extractor.conv_layers.{i}.layer_norm.bias"] = sd[f"feature_extractor.conv_layers.{i}.2.1.bias"]
        elif i == 0:
            out["feature_extractor.conv_layers.0.layer_norm.weight"] = sd["feature_extractor.conv_layers.0.2.weight"]
            out["feature_extractor.conv_layers.0.layer_norm.bias"] = sd["feature_extractor.conv_layers.0.2.bias"]
    out["feature_projection.layer_norm.weight"] = sd["layer_norm.weight"]
    out["feature_projection.layer_norm.bias"] = sd["layer_norm.bias"]
    out["feature_projection.projection.weight"] = sd["post_extract_proj.weight"]
    out["feature_projection.projection.bias"] = sd["post_extract_proj.bias"]
    out["encoder.pos_conv_embed.conv.bias"] = sd["encoder.pos_conv.0.bias"]
    out["encoder.pos_conv_embed.conv.parametrizations.weight.original0"] = sd["encoder.pos_conv.0.weight_g"]
    out["encoder.pos_conv_embed.conv.parametrizations.weight.original1"] = sd["encoder.pos_conv.0.weight_v"]
    out["encoder.layer_norm.weight"] = sd["encoder.layer_norm.weight"]
    out["encoder.layer_norm.bias"] = sd["encoder.layer_norm.bias"]
    for i in range(cfg.encoder_layers):
        p, q = f"encoder.layers.{i}.", f"encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            for w in ("weight", "bias"):
                out[f"{q}attention.{n}.{w}"] = sd[f"{p}self_attn.{n}.{w}"]
        for w in ("weight", "bias"):
            out[f"{q}layer_norm.{w}"] = sd[f"{p}self_attn_layer_norm.{w}"]
            out[f"{q}feed_forward.intermediate_dense.{w}"] = sd[f"{p}fc1.{w}"]
            out[f"{q}feed_forward.output_dense.{w}"] = sd[f"{p}fc2.{w}"]
            out[f"{q}final_layer_norm.{w}"] = sd[f"{p}final_layer_norm.{w}"]
    missing, unexpected = hf.load_state_dict(out, strict=False)
    assert not unexpected, unexpected
    assert all("masked_spec_embed" in m for m in missing), missing


@pytest.mark.parametrize("variant", ["base", "large"])
def test_hubert_ref_matches_hf(variant):
    torch.manual_seed(0)
    cfg = HubertRefConfig.tiny(layer_norm_first=(variant == "large"),
                               extractor_mode="layer_norm" if variant == "large" else "default",
                               conv_bias=(variant == "large"))
    ref = HubertModelRef(cfg).eval()
    randomize_norm_affine(ref, torch.Generator().manual_seed(1))
    hf = _hf_hubert(cfg)
    _copy_hubert(ref, hf)
    wav = torch.randn(3, 8000) * 0.3
    with torch.no_grad():
        o = hubert_forward(ref, wav, torch.zeros(3, 8000, dtype=torch.bool))
        h = hf(wav, output_hidden_states=True)
    ours, theirs = o["layer_results"], h.hidden_states
    assert len(ours) == len(theirs) == cfg.encoder_layers + 1
    n_cmp = len(ours) if not cfg.layer_norm_first else len(ours) - 1
    for i in range(n_cmp):
        torch.testing.assert_close(ours[i], theirs[i], atol=2e-5, rtol=1e-4)
    if cfg.layer_norm_first:   # HF's last state is post-final-LN; ours is pre-LN, `x` is post-LN
        torch.testing.assert_close(o["x"], theirs[-1], atol=2e-5, rtol=1e-4)


def test_clip_ref_matches_hf():
    from transformers import CLIPConfig, CLIPModel
    torch.manual_seed(0)
    cfg = ClipRefConfig(image_resolution=64, vision_patch=16, vision_width=128, vision_layers=2, embed_dim=64,
                        context_length=77, vocab_size=512, text_width=64, text_heads=1, text_layers=2)
    ref = ClipRef(cfg).eval()
    hc = CLIPConfig(
        text_config=dict(vocab_size=512, hidden_size=64, intermediate_size=256, num_hidden_layers=2,
                         num_attention_heads=1, max_position_embeddings=77, hidden_act="quick_gelu",
                         projection_dim=64, eos_token_id=511, bos_token_id=510, pad_token_id=0),
        vision_config=dict(hidden_size=128, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                           image_size=64, patch_size=16, hidden_act="quick_gelu", projection_dim=64),
        projection_dim=64)
    hf = CLIPModel(hc).eval()
    sd = ref.state_dict()
    out = {}
    out["vision_model.embeddings.class_embedding"] = sd["visual.class_embedding"]
    out["vision_model.embeddings.patch_embedding.weight"] = sd["visual.conv1.weight"]
    out["vision_model.embeddings.position_embedding.weight"] = sd["visual.positional_embedding"]
    out["vision_model.pre_layrnorm.weight"] = sd["visual.ln_pre.weight"]
    out["vision_model.pre_layrnorm.bias"] = sd["visual.ln_pre.bias"]
    out["vision_model.post_layernorm.weight"] = sd["visual.ln_post.weight"]
    out["vision_model.post_layernorm.bias"] = sd["visual.ln_post.bias"]
    out["visual_projection.weight"] = sd["visual.proj"].t()
    out["text_model.embeddings.token_embedding.weight"] = sd["token_embedding.weight"]
    out["text_model.embeddings.position_embedding.weight"] = sd["positional_embedding"]
    out["text_model.final_layer_norm.weight"] = sd["ln_final.weight"]
    out["text_model.final_layer_norm.bias"] = sd["ln_final.bias"]
    out["text_projection.weight"] = sd["text_projection"].t()
    out["logit_scale"] = sd["logit_scale"]
    for tower, src, width, n in (("vision_model", "visual.transformer", 128, 2), ("text_model", "transformer", 64, 2)):
        for i in range(n):
            s, d = f"{src}.resblocks.{i}.", f"{tower}.encoder.layers.{i}."
            w, b = sd[s + "attn.in_proj_weight"], sd[s + "attn.in_proj_bias"]
            for j, nme in enumerate(("q_proj", "k_proj", "v_proj")):
                out[d + f"self_attn.{nme}.weight"] = w[j * width:(j + 1) * width]
                out[d + f"self_attn.{nme}.bias"] = b[j * width:(j + 1) * width]
            for wb in ("weight", "bias"):
                out[d + f"self_attn.out_proj.{wb}"] = sd[s + f"attn.out_proj.{wb}"]
                out[d + f"layer_norm1.{wb}"] = sd[s + f"ln_1.{wb}"]
                out[d + f"layer_norm2.{wb}"] = sd[s + f"ln_2.{wb}"]
                out[d + f"mlp.fc1.{wb}"] = sd[s + f"mlp.c_fc.{wb}"]
                out[d + f"mlp.fc2.{wb}"] = sd[s + f"mlp.c_proj.{wb}"]
    missing, unexpected = hf.load_state_dict(out, strict=False)
    assert not unexpected, unexpected
    assert all("position_ids" in m for m in missing), missing
    img = torch.randn(3, 3, 64, 64)
    text = torch.randint(1, 500, (3, 77))
    text[:, 0] = 510
    text[:, 10] = 511          # EOT = max id -> argmax position (clip/model.py encode_text)
    text[:, 11:] = 0
    with torch.no_grad():
        hi = hf.visual_projection(hf.vision_model(pixel_values=img).pooler_output)
        ht = hf.text_projection(hf.text_model(input_ids=text).pooler_output)
        torch.testing.assert_close(ref.encode_image(img), hi, atol=2e-5, rtol=1e-4)
        torch.testing.assert_close(ref.encode_text(text), ht, atol=2e-5, rtol=1e-4)

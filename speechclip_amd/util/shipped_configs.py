"""The shipped SpeechCLIP configurations as config objects with the reference's YAML schema (config/speechCLIP/model_{base,large}/**/spchclp_{p,c}.yaml
keys and values): what `run_task.py --config ...` would hand to the model, built programmatically so bench.py / smoke() / the tests do not
need the YAML files of the reference tree.  Non-default `hubert_config` / `clip_config` give the tiny layouts the fixtures use."""
import dataclasses


def make_config(d_model=768, branch_heads=8, parallel=True, cascaded=False, hubert_name="hubert", clip_name="ViT-B/32",
                hubert_config=None, clip_config=None, normalize_hiddenstates=False, temperature_trainable=False,
                reduce_vocab=None, keyword_num=8):
    from ..base import OrderedNamespace
    targs = dict(n_layers=1, d_model=d_model, nhead=branch_heads, dim_feedforward=4 * d_model, dropout=0.1, activation="gelu",
                 layer_norm_eps=1e-5, batch_first=True, norm_first=False)
    cfg = {
        "model_settings": {
            "cascaded_objective_weight": 1.0 if cascaded else 0.0,
            "parallel_objective_weight": 1.0 if parallel else 0.0,
            "parallel_branch": {"transformer_type": "TransformerEncoder", "transformer_args": dict(targs), "need_projection": True},
            "cascaded_branch": {
                "type": "KW_CascadedBranch", "transformer_type": "MultiheadAttentionAndNorm",
                "transformer_args": dict(targs, nhead=1),
                "keyword": {"number": keyword_num, "detokenized_K_neighbors": 5, "retrieve_method": "cosine",
                            "batchnorms": {"type": "eachKw", "std_scale": 1.0, "learnable": True, "parallel": True}},
                "vq": {"bn_before_vq": True, "activation": "gelu", "type": "SimpleVectorQuantizer",
                       "args": {"temp": "fixed=0.1", "time_first": True, "use_gumbel": False, "hard": True}},
            },
        },
        "cl_loss": {"type": "MaskedContrastiveLoss",
                    "args": {"temperature": 0.07, "temperature_trainable": temperature_trainable, "margin": 0.0, "dcl": False,
                             "a2b": True, "b2a": True}},
        "retrieval": {"audio_feat_src": "cascaded" if cascaded and not parallel else "parallel", "recall_at": [1, 5, 10]},
        "clip": {"name": clip_name, "image_encoder_trainable": False, "text_encoder_trainable": False,
                 "reduce_subword_embbedding": reduce_vocab},
        "audio_encoder": {"type": "FairseqHubert", "name": hubert_name, "pretrained": False, "trainable": False,
                          "feat_select_idx": "weighted_sum", "layer_drop": 0.0, "max_audio_len": 102400,
                          "normalize_hiddenstates": normalize_hiddenstates,
                          "optim": {"name": "Adam", "args": {"lr": 1e-4, "weight_decay": 1e-6}},
                          "scheduler": {"name": "linear_warmup_decay", "warmup": 5000, "max_step": 50000, "final_lr": 1e-8}},
        "trainer": {"max_steps": 50000, "gradient_clip_val": 4, "precision": 16},
        "log_setting": {"log_detokenize_results": False},
    }
    if hubert_config is not None:
        cfg["audio_encoder"]["hubert_config"] = dataclasses.asdict(hubert_config)
    if clip_config is not None:
        cfg["clip"]["clip_config"] = dataclasses.asdict(clip_config)
    return OrderedNamespace(cfg)

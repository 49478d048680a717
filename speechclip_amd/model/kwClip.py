"""SpeechCLIP model layer on MI355X -- the `avssl.model` plugin surface.

Mirrors the reference's class names, constructor (`config: OrderedNamespace`), hook protocol and output tuple layout
(avssl/model/kwClip.py:49-1497, avssl/model/base_model.py:10-26) so that task code written against the reference
(`run_task.py`, `example.py`, Lightning's training/validation hooks) runs unchanged, while every tensor op of
the forward / contrastive path executes in libspeechclip_hip.so.

Multi-GPU: the reference runs single-process `nn.DataParallel` and computes the loss on device 0 after DP's gather
(`training_step_end`, kwClip.py:147-191).  Here there is one process per GPU; `*_step_end` all-gathers the per-rank
feature dicts over RCCL (speechclip_amd/parallel.py) in rank-major order (= DP's dim-0 concat order) and every rank
evaluates the loss on the global batch.
"""
import json
import logging
import os
from typing import List, Tuple, Union

import torch
from torch import nn

from .. import ops, parallel

_OVERLAP_IMAGE_TOWER = os.environ.get("SC_OVERLAP_VIT", "1") != "0"
_SIDE_PRIORITY = int(os.environ.get("SC_SIDE_PRIORITY", "0"))      # HIP stream priority of the image tower's side stream: 0 = default (lowest), -1 = high (A/B: profiles/r06_side_stream_priority_ab.txt)
_VIT_START = os.environ.get("SC_VIT_START", "")       # "" = the image tower starts with the step; "extractor" / "layer<i>": behind that stage of the speech tower (A/B);
                                                       # "head": behind the whole speech tower, beside the pooling / keyword head (the cascaded head is ~2.5 ms of small kernels)
_SIDE_STREAMS = {}
from ..base import OrderedNamespace
from ..module import ClipModel, FairseqSpeechEncoder_Hubert, MLPLayers, S3prlSpeechEncoderPlus, losses, mutualRetrieval
from ..module.kw_modules import TransformerModels
from ..module.speechclip_c_modules import vector_quantizers
from ..module.speechclip_c_modules.kw_bn import Kw_BatchNorm
from ..optim import get_scheduler
from ..util import get_keypadding_mask  # noqa: F401  (re-exported name used by downstream code)
from .base_model import BaseLightningModel

logger = logging.getLogger(__name__)
__all__ = ["KWClip_GeneralTransformer"]

# reduction applied to each logged metric gathered from the replicas (kwClip.py:41-46)
METRIC_REDUCEFN_MAPPING = {torch.Tensor: lambda x: torch.mean(x), float: lambda x: x, int: lambda x: x, str: lambda x: x}


class KWClipBase(BaseLightningModel):
    def __init__(self, config: OrderedNamespace):
        super().__init__(config)
        self.audio_encoder_type = config.audio_encoder.type
        if self.audio_encoder_type == "s3prl_plus":
            self.audio_encoder = S3prlSpeechEncoderPlus(**config.audio_encoder)
        elif self.audio_encoder_type == "FairseqHubert":
            self.audio_encoder = FairseqSpeechEncoder_Hubert(**config.audio_encoder)
        elif self.audio_encoder_type == "s3prl":
            raise DeprecationWarning("Please use s3prl_plus")
        else:
            logger.warning("No audio encoder loaded")
        self.clip = ClipModel(**config.clip)
        if hasattr(self, "audio_encoder"):
            self.audio_embd_dim = self.audio_encoder.out_dim
        self.subword_embd_dim = self.clip.model.token_embedding.weight.size(-1)
        self.recall_at = config.retrieval.recall_at
        self.criterion = getattr(losses, config.cl_loss.type)(**config.cl_loss.args)
        self.log_detokenize_results = config.log_setting.get("log_detokenize_results", True)
        self.keyword_num = self.config.model_settings.cascaded_branch.keyword.number

    # ---- encoders -------------------------------------------------------------------------------------
    def forward_audio(self, wav, wav_len=[], return_hidden_states: bool = False):
        if self.audio_encoder_type in ["s3prl_plus", "FairseqHubert"]:
            return self.audio_encoder(wav, wav_len, return_hidden_states=return_hidden_states)
        raise NotImplementedError("Unknown type:{}".format(self.audio_encoder_type))

    def forward_image(self, images: Union[list, torch.Tensor]) -> torch.Tensor:
        if isinstance(images, list):          # kwClip.py:517-518: paths -> ClipModel.prep_image (PIL + CLIP `_transform`) on the model's device
            self.clip.update_device(self.device)
            images = self.clip.prep_image(images)
        elif isinstance(images, torch.Tensor):
            if images.dim() != 4 or images.shape[1] != 3:
                raise ValueError(f"Incorrect image tensor shape {images.shape}")
        else:
            raise TypeError(f"Unknown image type {type(images)}")
        return self.clip.encode_image(images)

    def forward(self, batch: dict) -> tuple:
        raise NotImplementedError()

    def compute_loss(self, input_feats):
        raise NotImplementedError()

    # ---- Lightning hook protocol --------------------------------------------------------------------------
    def training_step(self, batch: dict, batch_idx: int = 0) -> dict:
        loss_feats, log_metrics = self.forward(batch)[:2]
        return {"loss_feats": loss_feats, "log_metrics": log_metrics}

    def _reduce_metrics(self, prefix, losses_, log_metrics):
        out = {f"{prefix}_{k}": losses_[k] for k in losses_}
        out.update({f"{prefix}_{k}": METRIC_REDUCEFN_MAPPING[type(log_metrics[k])](log_metrics[k]) for k in log_metrics})
        return out

    def training_step_end(self, outputs: dict) -> dict:
        if not isinstance(outputs, dict):
            raise NotImplementedError()
        if "loss" in outputs:
            return {"loss": torch.mean(outputs["loss"])}
        if "loss_feats" in outputs and "log_metrics" in outputs:
            from ..train_tail import gather_loss_feats_train
            losses_ = self.compute_loss(gather_loss_feats_train(outputs["loss_feats"]))
            self.log_dict(self._reduce_metrics("train", losses_, outputs["log_metrics"]), on_step=True, on_epoch=True,
                          prog_bar=True, logger=True, sync_dist=True)
            return {"loss": losses_["loss"]}
        raise NotImplementedError()

    def validation_step(self, batch: dict, batch_idx: int = 0) -> dict:
        loss_feats, log_metrics, others = self.forward(batch)
        audio_feat = others["cascaded_audio_feat"] if self.config.retrieval.audio_feat_src == "cascaded" else others["parallel_audio_feat"]
        ret = {"id": others["id"], "audio_feat": audio_feat}
        if others.get("image_feat") is not None:
            ret["image_feat"] = others["image_feat"]
        if others.get("text_feat") is not None:
            ret["text_feat"] = others["text_feat"]
        if others.get("keywords") is not None:
            ret["keywords"] = others["keywords"]
            ret["gold_text"] = batch.get("text")
        return {"loss_feats": loss_feats, "log_metrics": log_metrics, "others": ret}

    def validation_step_end(self, outputs: dict) -> dict:
        assert isinstance(outputs, dict)
        losses_ = self.compute_loss(parallel.gather_loss_feats(outputs["loss_feats"]))
        self.log_dict(self._reduce_metrics("val", losses_, outputs["log_metrics"]), on_step=True, on_epoch=True, prog_bar=True,
                      logger=True, sync_dist=True)
        others = parallel.gather_rows_dict(outputs["others"])     # all ranks' rows: recall is ranked against the full candidate pool
        for k in others:
            if isinstance(others[k], torch.Tensor):
                others[k] = others[k].detach().cpu()
        return others

    def detokenize_keywords(self, outputs: list):
        """Keyword de-tokenisation of validation_epoch_end (kwClip.py:277-466): the K nearest sub-words of every keyword embedding (cosine,
        or the pseudo-inverse read-out), the per-keyword hit rate against the gold caption's sub-word set, and the two JSON logs under
        <default_root_dir>/detokenizeText/.  The reference runs the [N*K, V] similarity + top-K on the CPU in dev-batch chunks; here they run
        on the device (sc_cosine_scores fp32 / sc_sgemm + sc_topk_rows_f32), the set logic stays on the host, chunking and output format
        unchanged.  Returns (hit_rate [keyword_num] in %, kw_top_ret, all_retok_outputs) -- the reference returns nothing and only logs."""
        root = os.path.join(self.config.trainer.default_root_dir, "detokenizeText")
        os.makedirs(root, exist_ok=True)
        epoch = int(getattr(self, "current_epoch", 0) or 0)
        if hasattr(self, "log_detokenize_results_every_n_epoch") and epoch % self.log_detokenize_results_every_n_epoch != 0:
            return None
        tok = self.clip.tokenizer
        gold_texts = [tok.decode(sent.squeeze().tolist()) for x in outputs for sent in x["gold_text"]]
        kw = torch.cat([x["keywords"] for x in outputs], dim=0)
        kw = kw.view(kw.shape[0], self.keyword_num, kw.shape[-1])
        assert kw.dim() == 3 and kw.shape[2] == self.subword_embd_dim, kw.shape
        emb = self.clip.model.token_embedding.weight.detach()
        kwcfg = self.config.model_settings.cascaded_branch.keyword
        K = kwcfg.get("detokenized_K_neighbors", 10)
        if not hasattr(kwcfg, "retrieve_method"):
            kwcfg.retrieve_method = "cosine"
        assert kwcfg.retrieve_method in ["cosine", "pseudo_inverse"]
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError("detokenize_keywords runs its similarity / top-K kernels on the GPU; move the model to cuda")
        emb_dev = emb.to(dev).float().contiguous()
        if kwcfg.retrieve_method == "pseudo_inverse":
            emb_pinv = torch.linalg.pinv(emb.detach().cpu().float().T).float().to(dev).contiguous()      # [V, D]
        reduced = self.clip.selected_text_emb_ids is not None
        orig = (lambda i: self.clip.reducedl2Original[i]) if reduced else (lambda i: i)
        hit_rate = [0] * self.keyword_num
        kw_top_ret = [[] for _ in range(self.keyword_num)]
        all_retok_outputs = []
        bs = self.config.data.dev_batch_size
        print("Detokenizing K={}".format(K))
        for i in range(0, len(gold_texts) + bs, bs):
            _gold_texts = gold_texts[i:i + bs]
            _bsz = len(_gold_texts)
            if _bsz == 0:
                break
            gold_sets = [set(tok.encode(_text)) for _text in _gold_texts]
            flat = kw[i:i + _bsz].reshape(-1, self.subword_embd_dim).float().to(dev).contiguous()
            if kwcfg.retrieve_method == "pseudo_inverse":
                score = ops.sgemm(flat, emb_pinv, transb=True)                     # (pinv @ kw^T)^T, kwClip.py:362-371
            else:
                score = ops.cosine_scores(flat, emb_dev, exact=True)               # F.cosine_similarity, fp32 (:372-379)
            k_values, k_indices = ops.topk_rows(score, K)
            k_values = k_values.view(_bsz, self.keyword_num, K).cpu()
            k_indices = k_indices.view(_bsz, self.keyword_num, K).cpu()
            for x in range(_bsz):
                tmp_outputs = {}
                for kw_i in range(self.keyword_num):
                    name = "keyword_{}".format(kw_i)
                    tmp_outputs[name] = []
                    top_k_toks = set(orig(_ind.item()) for _ind in k_indices[x, kw_i])
                    common = top_k_toks & gold_sets[x]
                    if bool(common):
                        hit_rate[kw_i] += 1
                        kw_top_ret[kw_i].append(int(list(common)[0]))
                    for _ind, _dist in zip(k_indices[x, kw_i], k_values[x, kw_i]):
                        tmp_outputs[name].append([tok.decoder[orig(_ind.item())], _dist.item()])
                # "gold" is the chunk's FIRST caption for every row of the chunk, as in the reference (kwClip.py:428 indexes with i)
                all_retok_outputs.append({"gold": gold_texts[i], "neighbors": tmp_outputs})
        hit_rate = torch.FloatTensor(hit_rate) / len(gold_texts) * 100
        print("kw_hit_rate", hit_rate)
        if getattr(self, "logger", None) is not None:
            self.log("kw_hit_rate", {"kw_{}".format(i): hit_rate[i].item() for i in range(self.keyword_num)}, sync_dist=True)
        if parallel.world()[0] == 0:      # every rank holds the gathered outputs (validation_step_end): one writer
            with open(os.path.join(root, "kw_hit_ep{}.json".format(epoch)), "w") as f:
                json.dump(kw_top_ret, f)
            with open(os.path.join(root, "keywords_ep{}.json".format(epoch)), "w") as f:
                json.dump(all_retok_outputs, f)
        return hit_rate, kw_top_ret, all_retok_outputs

    def validation_epoch_end(self, outputs: list):
        """kwClip.py:270-502: keyword de-tokenisation logging when the step outputs carry keywords (:277-466), then retrieval (:468-502)."""
        if "keywords" in outputs[0].keys():
            self.last_kw_hit_rate = self.detokenize_keywords(outputs)
        all_ids = torch.cat([x["id"] for x in outputs], dim=0)
        all_imgs = torch.cat([x["image_feat"] for x in outputs], dim=0)
        first = {}
        for i, _id in enumerate(all_ids.tolist()):       # last occurrence wins, first-seen order (dict semantics)
            first[_id] = i
        img_ids = torch.tensor(list(first.keys()), dtype=torch.long)
        img_feats = all_imgs[torch.tensor(list(first.values()), dtype=torch.long)]
        aud_feats = torch.cat([x["audio_feat"] for x in outputs], dim=0)
        print("Total #{} images, #{} audio".format(len(img_feats), len(aud_feats)))
        dev = self.device
        if dev.type != "cuda":
            from .._lib import SpeechClipHipError
            raise SpeechClipHipError("validation_epoch_end scores and ranks on the MI355X (sc_sgemm + sc_retrieval_ranks): move the model to the GPU; "
                                     "there is no host fallback")
        # fp32 similarity matrix on the device (kwClip.py:487-491), then the rank kernel
        score = ops.sgemm(aud_feats.float().to(dev).contiguous(), img_feats.float().to(dev).contiguous(), transb=True)
        return self.reportRetrieval(score_per_A=score, score_per_B=score.T.contiguous(), AB_answers=all_ids, BA_answers=img_ids)

    def reportRetrieval(self, score_per_A, score_per_B, AB_answers, BA_answers,
                        metadata={"modality_A_title": "audio", "modality_B_title": "image", "modality_A_logAbbr": "A",
                                  "modality_B_logAbbr": "I"}):
        for k in ("modality_A_title", "modality_B_title", "modality_A_logAbbr", "modality_B_logAbbr"):
            assert k in metadata
        r_ab, r_ba, r_mean = mutualRetrieval(score_per_A=score_per_A, score_per_B=score_per_B, AB_answers=AB_answers,
                                             BA_answers=BA_answers, recall_at=self.recall_at,
                                             modality_A_title=metadata["modality_A_title"], modality_B_title=metadata["modality_B_title"])
        ab, ba = metadata["modality_A_logAbbr"] + metadata["modality_B_logAbbr"], metadata["modality_B_logAbbr"] + metadata["modality_A_logAbbr"]
        print(f"val_recall_{ab}", r_ab)
        print(f"val_recall_{ba}", r_ba)
        print("val_recall_mean", r_mean)
        if getattr(self, "logger", None) is not None:
            self.log("val_recall_mean_10", r_mean.get("recall@10", 0.0), sync_dist=True)
        return r_ab, r_ba, r_mean

    def processWavs(self, wav):
        wav_len = [len(x) for x in wav]
        return wav, wav_len

    def feature_extractor_s3prl(self, wav):
        raise NotImplementedError()

    def getTrainableParams(self) -> list:
        params = []
        if hasattr(self, "audio_encoder"):
            params += self.audio_encoder.trainable_params()
            params += list(self.criterion.parameters())
        params += self.clip.trainable_params()
        return params

    def configure_optimizers(self) -> Tuple[list, list]:
        params = self.getTrainableParams()
        oc = self.config.audio_encoder.optim
        if oc.name == "Adam" and all(p.is_cuda for p in params):
            # same update rule as torch.optim.Adam, on one flat buffer (sc_grad_norm + sc_adam_step); Lightning's
            # trainer.gradient_clip_val (clip_grad_norm_) is folded into the step
            from ..train_tail import FusedAdam
            clip = float(self.config.trainer.get("gradient_clip_val", 0.0) or 0.0) if hasattr(self.config, "trainer") else 0.0
            opt = FusedAdam(params, max_grad_norm=clip, **oc.args)
        else:
            opt = getattr(torch.optim, oc.name)(params, **oc.args)
        sched = get_scheduler(optimizer=opt, **self.config.audio_encoder.scheduler)
        return [opt], [{"scheduler": sched, "interval": "step"}]


class KW_ParallelBranch(nn.Module):
    """kwClip.py:1004-1108: learned [CLS] + 1 post-LN encoder layer + final LN, CLS row, Linear(d -> E)."""

    def __init__(self, config: OrderedNamespace, audio_dim: int, out_dim: int) -> None:
        super().__init__()
        self.config, self.audio_dim, self.out_dim = config, audio_dim, out_dim
        pb = config.model_settings.parallel_branch
        self.need_projection = pb.get("need_projection", True)
        assert hasattr(TransformerModels, pb.transformer_type)
        self.self_att = getattr(TransformerModels, pb.transformer_type)(**pb.transformer_args)
        self.cls = torch.nn.Parameter(torch.randn([1, 1, pb.transformer_args.d_model]))
        if self.need_projection:
            self.linear_proj = nn.Linear(self.audio_dim, self.out_dim)

    def extract_hidden_states(self, audio_feat: torch.Tensor, audio_len: torch.Tensor) -> tuple:
        """kwClip.py:1049-1076: hidden representation of every layer of the branch over [CLS; frames], CLS position dropped."""
        return _branch_hidden_states(self, audio_feat, audio_len, 1)

    def forward(self, audio_feat: torch.Tensor, audio_len: torch.Tensor) -> torch.Tensor:
        if torch.is_grad_enabled() and self.cls.requires_grad:
            return self._forward_train(audio_feat, audio_len)
        out = self.self_att.forward_cls(self.cls, audio_feat, audio_len)            # f32 [B, d] (bf16 with SC_HEAD_PRECISE=0)
        if hasattr(self, "linear_proj"):
            if out.dtype == torch.float32:
                out = TransformerModels.hp_linear(out, self.linear_proj.weight, self.linear_proj.bias)
            else:
                out = ops.gemm(out, TransformerModels.cached_cast(self.linear_proj.weight, torch.bfloat16),
                               TransformerModels.cached_cast(self.linear_proj.bias, torch.float32), out_f32=True)
        return out


def _branch_hidden_states(branch, audio_feat: torch.Tensor, audio_len: torch.Tensor, n_cls: int) -> tuple:
    """Shared body of KW_ParallelBranch / KW_CascadedBranch.extract_hidden_states (kwClip.py:828-856, :1049-1076): full-row pass of
    [CLS tokens; frames] with the `len + n_cls` key-padding mask, then the CLS positions are cut off."""
    B, T, D = audio_feat.shape
    src = torch.cat([branch.cls.detach().to(audio_feat.device, torch.float32).expand(B, -1, -1), audio_feat.detach().float()], dim=1)
    mask = get_keypadding_mask(max_length=T + n_cls, data_lens=audio_len.to(audio_feat.device) + n_cls).to(audio_feat.device)
    hidden = branch.self_att.extract_hidden_states(src=src, key_padding_mask=mask)
    return tuple(x[:, n_cls:, ...] for x in hidden)


def _kw_parallel_forward_train(self, audio_feat: torch.Tensor, audio_len: torch.Tensor) -> torch.Tensor:
    """Differentiable path (train_tail.ParallelBranchTrainFn): fp32 master weights, dropout active in train() mode."""
    from ..train_tail import ParallelBranchTrainFn
    L = self.self_att.model.layers[0]
    sa = L.self_attn
    src = getattr(audio_feat, "_mix_src", None)
    hidden, mixw, normalize = (src[0], src[1].weights, src[1].normalize_features) if src is not None else (None, None, False)
    drop_p = float(L.dropout.p) if self.training else 0.0
    seed = int(torch.randint(0, 2 ** 31 - 8, (1,)).item()) if drop_p > 0 else 0
    meta = dict(heads=self.self_att.nhead, eps=self.self_att.eps, drop_p=drop_p, seed=seed, normalize=normalize)
    proj = (self.linear_proj.weight, self.linear_proj.bias) if hasattr(self, "linear_proj") else (None, None)
    frames = audio_feat if audio_feat.requires_grad else audio_feat.detach()       # requires_grad: fine-tuned encoder layers below (train_hubert.py)
    return ParallelBranchTrainFn.apply(meta, hidden, frames, audio_len, mixw, self.cls, sa.in_proj_weight, sa.in_proj_bias,
                                       sa.out_proj.weight, sa.out_proj.bias, L.norm1.weight, L.norm1.bias, L.linear1.weight, L.linear1.bias,
                                       L.linear2.weight, L.linear2.bias, L.norm2.weight, L.norm2.bias, self.self_att.model.norm.weight,
                                       self.self_att.model.norm.bias, *proj)


KW_ParallelBranch._forward_train = _kw_parallel_forward_train


class KW_CascadedBranch(nn.Module):
    """kwClip.py:697-916: K learned [CLS] keywords -> LN(MHA+x) -> Linear -> BatchNorm -> cosine vs sub-word embeddings ->
    VQ -> CLIP text encoder."""

    def __init__(self, config: OrderedNamespace, audio_dim: int, text_dim: int, clip: ClipModel) -> None:
        super().__init__()
        self.audio_dim, self.text_dim, self.clip, self.config = audio_dim, text_dim, clip, config
        cb = config.model_settings.cascaded_branch
        self.kw_projection_config = cb.keyword.get("kw_projection", None)
        self.keyword_num = cb.keyword.number
        self.cls = torch.nn.Parameter(torch.randn([1, self.keyword_num, cb.transformer_args.d_model]))
        assert hasattr(TransformerModels, cb.transformer_type), "transformer structure '{}' not supported".format(cb.transformer_type)
        self.self_att = getattr(TransformerModels, cb.transformer_type)(**cb.transformer_args)
        if self.kw_projection_config is None:                      # kwClip.py:749-756
            self.linear_proj = nn.Linear(cb.transformer_args.d_model, self.text_dim)
        else:                                                      # kwClip.py:757-771: an MLP instead of the single Linear (eval path; not trainable here)
            dims = list(self.kw_projection_config.dimensions)
            assert dims[0] == cb.transformer_args.d_model, f"first dim({dims[0]}) should match the audio encoder dim({cb.transformer_args.d_model})"
            assert dims[-1] == self.text_dim, f"last dim({dims[-1]}) should match the text encoder dim({self.text_dim})"
            self.linear_proj = MLPLayers(units=dims, dropout=self.kw_projection_config.dropout)
        self.vq_type = cb.vq.type
        if not hasattr(vector_quantizers, self.vq_type):
            raise NotImplementedError("Vq ({}) not implemented".format(self.vq_type))
        self.vector_quantizer = getattr(vector_quantizers, self.vq_type)(**cb.vq.args)
        if hasattr(cb.keyword, "batchnorms"):
            bn = cb.keyword.batchnorms
            emb = self.clip.model.token_embedding.weight
            self.bn_layer = Kw_BatchNorm(kw_num=self.keyword_num, kw_dim=self.text_dim, batchnorm_type=bn.type,
                                         init_bias=torch.mean(emb, dim=0), init_scale=torch.std(emb, dim=0), std_scale=bn.std_scale,
                                         learnable=bn.learnable if hasattr(bn, "learnable") else True,
                                         parallel=bn.parallel if hasattr(bn, "parallel") else False)

    def extract_hidden_states(self, audio_feat: torch.Tensor, audio_len: torch.Tensor) -> tuple:
        """kwClip.py:828-856."""
        return _branch_hidden_states(self, audio_feat, audio_len, self.keyword_num)

    @torch.no_grad()
    def getAttentionMap(self, audio_feat: torch.Tensor, audio_len: torch.Tensor):
        """kwClip.py:918-1001: (cls_weights, topk_kw, None) for visualisation.  cls_weights[i] = the per-head attention probabilities of the
        keyword rows of utterance i over its valid positions, fp32 [H, K, len_i + K]; topk_kw[i][k] = the 10 nearest sub-words (special
        tokens 0 / 2 / 3 pushed down by 100, :977-979) of keyword k as printed by the tokenizer's decoder.  Eval-mode arithmetic."""
        bsz, K = audio_feat.size(0), self.keyword_num
        total_max_len = audio_feat.size(1) + K
        dev = audio_feat.device
        src = torch.cat([self.cls.detach().to(dev, torch.float32).expand(bsz, -1, -1), audio_feat.detach().float()], dim=1)
        audio_len = torch.as_tensor(audio_len).to(dev)
        key_padding_mask = get_keypadding_mask(max_length=total_max_len, data_lens=audio_len + K).to(dev)
        out, attn = self.self_att.extract_attention_map(src=src, key_padding_mask=key_padding_mask, query_rows=K)
        lens = audio_len.tolist()
        cls_weights = [attn[i, :, :K, :lens[i] + K] for i in range(bsz)]
        keywords = out[:, :K].reshape(bsz * K, self.audio_dim)
        keywords = ops.gemm(keywords.to(torch.bfloat16), TransformerModels.cached_cast(self.linear_proj.weight, torch.bfloat16),
                            TransformerModels.cached_cast(self.linear_proj.bias, torch.float32), out_f32=True).view(bsz, K, self.text_dim)
        if hasattr(self, "bn_layer"):
            was_training = self.bn_layer.training
            self.bn_layer.eval()
            keywords = self.bn_layer(keywords)
            self.bn_layer.train(was_training)
        emb = self.clip.model.token_embedding.weight
        cos = ops.cosine_scores(keywords.reshape(bsz * K, self.text_dim), emb, exact=True).view(bsz, K, emb.shape[0])
        cos[..., 0] -= 100
        cos[..., 2] -= 100
        cos[..., 3] -= 100
        _, topk_ids = ops.topk_rows(cos, 10)
        topk_ids = topk_ids.cpu()
        dec = self.clip.tokenizer.decoder
        orig = (lambda i: self.clip.reducedl2Original[i]) if self.clip.selected_text_emb_ids is not None else (lambda i: i)
        topk_kw = [[[dec[orig(x.item())].replace("</w>", "") for x in topk_ids[b, k]] for k in range(K)] for b in range(bsz)]
        return cls_weights, topk_kw, None

    def forward(self, audio_feat: torch.Tensor, audio_len: torch.Tensor):
        if self.training and torch.is_grad_enabled() and self.cls.requires_grad:
            return self._forward_train(audio_feat, audio_len)
        B, K = audio_feat.shape[0], self.keyword_num
        kw = self.self_att.forward_cls(self.cls, audio_feat, audio_len)                       # f32 [B, K, d] (bf16 with SC_HEAD_PRECISE=0)
        if isinstance(self.linear_proj, MLPLayers):
            kw = self.linear_proj(kw.view(B * K, -1)).view(B, K, self.text_dim)
        elif kw.dtype == torch.float32:
            kw = TransformerModels.hp_linear(kw.view(B * K, -1), self.linear_proj.weight, self.linear_proj.bias).view(B, K, self.text_dim)
        else:
            kw = ops.gemm(kw.view(B * K, -1), TransformerModels.cached_cast(self.linear_proj.weight, torch.bfloat16),
                          TransformerModels.cached_cast(self.linear_proj.bias, torch.float32), out_f32=True).view(B, K, self.text_dim)
        if hasattr(self, "bn_layer"):
            kw = self.bn_layer(kw)
        emb = self.clip.model.token_embedding.weight
        cos = ops.cosine_scores(kw.reshape(B * K, self.text_dim), emb).view(B, K, emb.shape[0])   # fp32, exact arg-max
        vq_results = self.vector_quantizer(x=cos)
        assert emb.requires_grad is False
        keywords = self.vector_quantizer.embed(vq_results, emb)
        feat = self.clip.encode_keywords(keywords, K)
        return feat, vq_results, keywords


def _kw_cascaded_forward_train(self, audio_feat: torch.Tensor, audio_len: torch.Tensor):
    """Differentiable train-mode path (kwClip.py:868-916 under loss.backward()): train_tail.CascadedPoolTrainFn -> Kw_BatchNorm (batch
    statistics) -> cosine / straight-through VQ (KeywordSTFn) -> frozen CLIP text tower with input gradients (TextTowerTrainFn)."""
    from ..train_tail import CascadedPoolTrainFn, KeywordSTFn
    if isinstance(self.linear_proj, MLPLayers):
        raise NotImplementedError("training with a kw_projection MLP is not built (no shipped config has one); the eval forward is")
    B, K = audio_feat.shape[0], self.keyword_num
    mha = self.self_att.multihead_attn_layer
    src = getattr(audio_feat, "_mix_src", None)
    hidden, mixw, normalize = (src[0], src[1].weights, src[1].normalize_features) if src is not None else (None, None, False)
    drop_p = float(mha.dropout)
    seed = int(torch.randint(0, 2 ** 31 - 8, (1,)).item()) if drop_p > 0 else 0
    meta = dict(heads=self.self_att.nhead, eps=self.self_att.eps, drop_p=drop_p, seed=seed, normalize=normalize)
    n = self.self_att.attentionBlock_Norm
    frames = audio_feat if audio_feat.requires_grad else audio_feat.detach()       # requires_grad: fine-tuned encoder layers below (train_hubert.py)
    kw = CascadedPoolTrainFn.apply(meta, hidden, frames, audio_len, mixw, self.cls, mha.in_proj_weight, mha.in_proj_bias,
                                   mha.out_proj.weight, mha.out_proj.bias, n.weight, n.bias, self.linear_proj.weight,
                                   self.linear_proj.bias).view(B, K, self.text_dim)
    if hasattr(self, "bn_layer"):
        kw = self.bn_layer(kw)
    emb = self.clip.model.token_embedding.weight
    assert emb.requires_grad is False
    cos = ops.cosine_scores(kw.detach().reshape(B * K, self.text_dim), emb)                     # fp32 [B*K, V]
    vq_results = self.vector_quantizer(x=cos.view(B, K, emb.shape[0]))
    vq = self.vector_quantizer      # a learnable temperature (vq.temp: "learnable=...") goes in as the parameter itself: KeywordSTFn returns its gradient
    temp = vq.curr_temp if getattr(vq, "temp_type", "") == "learnable" else vq.temperature_value()
    keywords = KeywordSTFn.apply(kw.reshape(B * K, self.text_dim), cos, vq_results["targets"].reshape(-1), emb, temp, (0, 2, 3)).view(B, K, emb.shape[1])
    feat = self.clip.encode_keywords(keywords, K)
    return feat, vq_results, keywords


KW_CascadedBranch._forward_train = _kw_cascaded_forward_train


class KWClip_GeneralTransformer(KWClipBase):
    def __init__(self, config: OrderedNamespace) -> None:
        super().__init__(config)
        ms = self.config.model_settings
        self.cascaded_branch = None
        self.parallel_branch = None
        if ms.cascaded_objective_weight > 0:
            if ms.cascaded_branch.type != "KW_CascadedBranch":
                raise NotImplementedError()
            self.cascaded_branch = KW_CascadedBranch(config=self.config, audio_dim=self.audio_embd_dim,
                                                     text_dim=self.subword_embd_dim, clip=self.clip)
        if ms.parallel_objective_weight > 0:
            self.parallel_branch = KW_ParallelBranch(config=self.config, audio_dim=self.audio_embd_dim, out_dim=self.subword_embd_dim)
        self.img_enc_proj_net = self.p_branch_proj_net = self.c_branch_proj_net = None
        if ms.get("image_encoder_projection", None) is not None:
            p = ms.image_encoder_projection
            self.img_enc_proj_net = MLPLayers(units=p.dimensions, dropout=p.dropout)
        if ms.get("parallel_branch_projection", None) is not None:
            p = ms.parallel_branch_projection
            self.p_branch_proj_net = MLPLayers(units=p.dimensions, dropout=p.dropout)
            if ms.get("cascaded_branch_projection", None) is not None:   # the reference gates this on the parallel key (kwClip.py:1178)
                p = ms.cascaded_branch_projection
                self.c_branch_proj_net = MLPLayers(units=p.dimensions, dropout=p.dropout)

    def getTrainableParams(self) -> list:
        # the optional MLP projection heads (image_encoder_projection / parallel_branch_projection / ..., kwClip.py:1161-1190 of the
        # reference; no shipped YAML has them) are eval-only here: say so when the optimizer is built, not at the first training forward
        from ..module import MLPLayers
        # (only the four projection attributes of THIS class, not every MLPLayers nested somewhere in a sub-module -- the cascaded branch's own
        #  kw_projection MLP has its own check at its training forward)
        heads = [n for n in ("img_enc_proj_net", "p_branch_proj_net", "c_branch_proj_net") if isinstance(getattr(self, n, None), MLPLayers)]
        if heads:
            raise NotImplementedError(f"training with the MLP projection heads {heads} is not built on the MI355X path (eval / inference only); "
                                      "remove the *_projection sections from the config to train (README: gaps)")
        params = super().getTrainableParams()
        for m in (self.cascaded_branch, self.parallel_branch, self.img_enc_proj_net, self.p_branch_proj_net):
            if m is not None:
                params += [p for p in m.parameters() if p.requires_grad]
        return params

    def compute_loss(self, input_feats: dict):
        assert isinstance(input_feats, dict) and "id" in input_feats and "image_feat" in input_feats
        assert "cascaded_audio_feat" in input_feats or "parallel_audio_feat" in input_feats
        ms = self.config.model_settings
        image_feat, ids = input_feats["image_feat"].float(), input_feats["id"]
        out = {"loss": 0}
        if ms.cascaded_objective_weight > 0:
            out["c_cl_loss"] = self.criterion(feat_A=input_feats["cascaded_audio_feat"].float(), feat_B=image_feat, index=ids)
            out["loss"] = out["loss"] + ms.cascaded_objective_weight * out["c_cl_loss"]
        if ms.parallel_objective_weight > 0:
            out["p_cl_loss"] = self.criterion(feat_A=input_feats["parallel_audio_feat"].float(), feat_B=image_feat, index=ids)
            out["loss"] = out["loss"] + ms.parallel_objective_weight * out["p_cl_loss"]
        return out

    def _branches(self, audio_feat, audio_len):
        c_feat = p_feat = vq = kw = None
        if self.cascaded_branch is not None:
            c_feat, vq, kw = self.cascaded_branch(audio_feat=audio_feat, audio_len=audio_len)
            if c_feat.requires_grad:
                from ..train_tail import L2NormFn
                c_feat = L2NormFn.apply(c_feat)
            else:
                c_feat = ops.l2norm(c_feat)
        if self.parallel_branch is not None:
            p_feat = self.parallel_branch(audio_feat=audio_feat, audio_len=audio_len)
            if self.p_branch_proj_net is not None:
                p_feat = self.p_branch_proj_net(p_feat)
            if p_feat.requires_grad:
                from ..train_tail import L2NormFn
                p_feat = L2NormFn.apply(p_feat)
            else:
                p_feat = ops.l2norm(p_feat)
        return c_feat, p_feat, vq, kw

    def encode_speech(self, wav) -> dict:
        wav, wav_len = self.processWavs(wav)
        audio_feat, audio_len = self.forward_audio(wav, wav_len)
        c_feat, p_feat, vq, kw = self._branches(audio_feat, audio_len)
        return {"cascaded_audio_feat": c_feat, "parallel_audio_feat": p_feat, "vq_results": vq, "keywords": kw}

    def feature_extractor_s3prl(self, wav):
        """kwClip.py:1213-1247: encoder hidden states followed by the branches' (cascaded first, then parallel; the branch input itself,
        element [0] of each, is dropped)."""
        wav, wav_len = self.processWavs(wav)
        audio_feat, audio_len, hidden_states = self.forward_audio(wav, wav_len, return_hidden_states=True)
        assert isinstance(hidden_states, tuple)
        if self.cascaded_branch is not None:
            c_hidden = self.cascaded_branch.extract_hidden_states(audio_feat, audio_len)
            assert isinstance(c_hidden, tuple)
            hidden_states = hidden_states + tuple(c_hidden[1:])
        if self.parallel_branch is not None:
            p_hidden = self.parallel_branch.extract_hidden_states(audio_feat, audio_len)
            assert isinstance(p_hidden, tuple)
            hidden_states = hidden_states + tuple(p_hidden[1:])
        return hidden_states[-1], hidden_states

    def forward(self, batch) -> tuple:
        wav, wav_len, image, ids = batch["wav"], batch["wav_len"], batch["image"], batch["id"]
        self.clip.update_device(self.device)
        branches = None
        if _OVERLAP_IMAGE_TOWER and image.is_cuda:
            # The frozen image tower does not depend on the speech tower: it runs on a side HIP stream and fills the CUs the speech tower's
            # kernels leave idle (GEMM tails, HBM-bound conv0 / LayerNorm phases): 45.9 -> 44.7 ms per B = 256 step.  SC_OVERLAP_VIT=0: serial.
            cur = torch.cuda.current_stream()
            side = _SIDE_STREAMS.get(image.device.index)          # one side stream per device for the whole process (the library path keeps
            if side is None:                                      # one workspace half per stream: vendor_gemm.hip)
                side = _SIDE_STREAMS[image.device.index] = torch.cuda.Stream(device=image.device, priority=_SIDE_PRIORITY)
            def launch_image():
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    if ops.PROFILE is not None:        # bench instrumentation: the window in which two kernels may share the CUs
                        w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        w0.record()
                    tag = ops.PROFILE_TAG
                    ops.PROFILE_TAG = "image"
                    feat = self.forward_image(image)
                    ops.PROFILE_TAG = tag
                    if ops.PROFILE is not None:
                        w1.record()
                        ops.PROFILE_SIDE.append((w0, w1))
                return feat
            # Where the image tower enters the launch sequence.  Parallel-only models: with the step (it hides in the speech tower's HBM-bound phases and GEMM tails;
            # every later start measured worse, EXPERIMENTS.md R5-3b).  Models with a CASCADED branch: behind the speech tower, beside the keyword head + VQ + CLIP text
            # tower -- ~2.5 ms of small, serially dependent kernels that leave most CUs idle: 44.50 -> 44.00 ms per C-base step in 3 of 3 interleaved passes
            # (profiles/r06_vit_beside_cascaded_head_ab.txt; the same placement costs the parallel model +1.6 ms).  SC_VIT_START overrides ("step" = with the step).
            vit_start = _VIT_START or ("head" if self.cascaded_branch is not None else "")
            if vit_start == "step":
                vit_start = ""
            if not vit_start:
                image_feat = launch_image()
                audio_feat, audio_len = self.forward_audio(wav, wav_len)
            else:                                  # A/B: the image tower enters the launch sequence behind a stage of the speech tower (module/hubert.py STAGE_HOOK)
                from ..module import hubert as _hubert
                box = []

                def hook(name):
                    if name == vit_start and not box:
                        box.append(launch_image())
                _hubert.STAGE_HOOK = hook
                try:
                    audio_feat, audio_len = self.forward_audio(wav, wav_len)
                finally:
                    _hubert.STAGE_HOOK = None
                image_feat = box[0] if box else launch_image()
            if vit_start == "head":               # the head's kernels are enqueued first; the main stream joins the side stream in front of the first use of image_feat
                ops.PROFILE_TAG = "head"
                branches = self._branches(audio_feat, audio_len)
                ops.PROFILE_TAG = "speech"
            cur.wait_stream(side)
            image_feat.record_stream(cur)
        else:
            audio_feat, audio_len = self.forward_audio(wav, wav_len)
            ops.PROFILE_TAG = "image"
            image_feat = self.forward_image(image)
            ops.PROFILE_TAG = "speech"
        if self.img_enc_proj_net is not None:
            image_feat = self.img_enc_proj_net(image_feat)
        if branches is None:
            ops.PROFILE_TAG = "head"
            branches = self._branches(audio_feat, audio_len)
            ops.PROFILE_TAG = "speech"
        c_feat, p_feat, vq, kw = branches
        image_feat = ops.l2norm(image_feat)
        loss_feats = {"id": ids, "image_feat": image_feat}
        log_metrics = {}
        if c_feat is not None:
            loss_feats["cascaded_audio_feat"] = c_feat
        if p_feat is not None:
            loss_feats["parallel_audio_feat"] = p_feat
        if self.config.model_settings.cascaded_objective_weight > 0:
            log_metrics["softmax_temp"] = vq["temp"]
        log_metrics["cl_temp"] = self.criterion.current_temperature
        others = {"cascaded_audio_feat": c_feat, "parallel_audio_feat": p_feat, "image_feat": image_feat, "id": ids,
                  "vq_results": vq, "keywords": kw}
        return loss_feats, log_metrics, others

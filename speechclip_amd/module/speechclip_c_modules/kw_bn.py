"""Kw_BatchNorm (avssl/module/speechclip_c_modules/kw_bn.py:8-164), shipped mode: `eachKw` + `parallel` = one
BatchNorm1d over the flattened (dim, keyword) features, initialised from the CLIP token-embedding mean/std.
Eval mode (running statistics) is an affine map executed by sc_kw_affine; train mode (sc_kw_bn_train_fwd / sc_kw_bn_bwd) uses the batch
statistics of this rank's rows (not synced), as each replica does under the reference's DataParallel."""
import torch
from torch import nn

from ... import ops


class Kw_BatchNorm(nn.Module):
    def __init__(self, kw_num: int, kw_dim: int, batchnorm_type: str, init_bias: torch.Tensor, init_scale: torch.Tensor,
                 std_scale: float = 1, learnable: bool = True, parallel: bool = False) -> None:
        super().__init__()
        if batchnorm_type != "eachKw" or not parallel:
            raise NotImplementedError("MI355X path supports the shipped batchnorm mode: type eachKw, parallel true")
        self.batchnorm_type, self.kw_num, self.kw_dim, self.learnable, self.parallel = batchnorm_type, kw_num, kw_dim, learnable, parallel
        self.std_scale = std_scale if isinstance(std_scale, list) else [std_scale] * kw_num
        self.bn_layer = nn.BatchNorm1d(kw_dim * kw_num)
        with torch.no_grad():
            self.bn_layer.weight.copy_((init_scale * self.std_scale[0]).repeat(kw_num))
            self.bn_layer.bias.copy_(init_bias.repeat(kw_num))
        self.bn_layer.weight.requires_grad = learnable
        self.bn_layer.bias.requires_grad = learnable

    def forward(self, keywords: torch.Tensor, seq_lens: torch.Tensor = None) -> torch.Tensor:
        assert keywords.dim() == 3 and keywords.shape[2] == self.kw_dim and keywords.shape[1] == self.kw_num
        bn = self.bn_layer
        if self.training:       # batch statistics of this rank's rows + running-stat update (train_tail.KwBatchNormTrainFn)
            from ...train_tail import KwBatchNormTrainFn
            if not (bn.track_running_stats and bn.momentum is not None):
                raise NotImplementedError("Kw_BatchNorm training path expects nn.BatchNorm1d defaults (tracked running stats, momentum 0.1)")
            with torch.no_grad():
                bn.num_batches_tracked += 1
            return KwBatchNormTrainFn.apply(keywords, bn.weight, bn.bias, bn.running_mean, bn.running_var, float(bn.momentum), float(bn.eps))
        # flattened feature index of (k, d) is d*K + k  (kw_bn.py:122-131: permute(0,2,1).reshape(B,-1))
        K, D = self.kw_num, self.kw_dim
        scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).view(D, K).t().contiguous().float()
        shift = (bn.bias - bn.running_mean * (bn.weight / torch.sqrt(bn.running_var + bn.eps))).view(D, K).t().contiguous().float()
        return ops.kw_affine(keywords, scale.detach(), shift.detach())

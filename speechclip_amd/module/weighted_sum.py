"""WeightedSumLayer -- same constructor / call contract as avssl/module/weighted_sum.py:10-45, computed by
sc_weighted_sum_fwd (softmax over the n scalars, optional per-feature layer_norm, one streaming pass)."""
from typing import Sequence, Union

import torch
import torch.nn as nn

from .. import ops


class WeightedSumLayer(nn.Module):
    def __init__(self, n_weights: int, normalize_features: bool = False):
        super().__init__()
        self.n_weights = n_weights
        self.weights = nn.Parameter(torch.zeros((n_weights,), dtype=torch.float))
        self.normalize_features = normalize_features

    def forward(self, x: Union[Sequence[torch.Tensor], torch.Tensor]) -> torch.Tensor:
        """x: list of n tensors [B,T,D] (or one stacked [n,B,T,D] tensor, which avoids the stack copy)."""
        if not torch.is_tensor(x):
            assert len(x) == self.n_weights, len(x)
            base = getattr(x[0], "_base", None)
            if (base is not None and base.dim() == 4 and base.shape[0] == self.n_weights and base.is_contiguous()
                    and all(h._base is base for h in x) and x[0].data_ptr() == base.data_ptr()):
                x = base          # the layers are slices of one stacked buffer: no copy
            else:
                x = torch.stack(list(x), dim=0)
        n, B, T, D = x.shape
        assert n == self.n_weights, n
        out = ops.weighted_sum(x.reshape(n, B * T, D), self.weights.detach().float(), self.normalize_features)
        return out.view(B, T, D)

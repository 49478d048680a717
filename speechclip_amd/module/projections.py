"""MLPLayers (avssl/module/projections.py:6-29): the optional projection heads -- `image_encoder_projection`, `parallel_branch_projection`,
`cascaded_branch_projection` (kwClip.py:1147-1187) and the keyword `kw_projection` (kwClip.py:757-771) -- absent from every shipped YAML.

Same constructor and the reference's `state_dict` layout (`sequential.{0,3,6,...}.{weight,bias}`: nn.Sequential of [Linear, nonlin, Dropout] triples with the last
two modules cut off), so a checkpoint trained with such a head loads by key.  EVAL forward on the device: every Linear is a hi/lo-split MFMA GEMM on fp32 rows
(`hp_linear`, as the pooling heads' CLS rows), ReLU between them.  Training these optional heads (autograd + train-mode dropout) is not built."""
import torch
import torch.nn as nn

__all__ = ["MLPLayers"]


class MLPLayers(nn.Module):
    def __init__(self, units=(512, 512, 512), nonlin=None, dropout=0.1):
        super().__init__()
        self.nonlin = nn.ReLU() if nonlin is None else nonlin
        if not isinstance(self.nonlin, nn.ReLU):
            raise NotImplementedError("MLPLayers on the MI355X path: ReLU (the reference's default) only")
        self.dropout = dropout
        seq = []
        for u0, u1 in zip(units[:-1], units[1:]):
            seq += [nn.Linear(u0, u1), self.nonlin, nn.Dropout(self.dropout)]
        self.sequential = nn.Sequential(*seq[:-2])

    def forward(self, X: torch.Tensor) -> torch.Tensor:
        from .kw_modules.TransformerModels import hp_linear
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()) and self.training:
            raise NotImplementedError("training the optional MLP projection heads is not built (no shipped config has them)")
        shape = X.shape
        x = X.detach().float().reshape(-1, shape[-1]).contiguous()
        for m in self.sequential:
            if isinstance(m, nn.Linear):
                x = hp_linear(x, m.weight, m.bias)
            elif isinstance(m, nn.ReLU):
                x = torch.relu_(x)
            # nn.Dropout: identity in eval
        return x.view(*shape[:-1], x.shape[-1])

"""MLPLayers (avssl/module/projections.py:6-29): optional projection heads, absent from every shipped YAML.
Kept for API parity; runs through the MFMA GEMM (bias + ReLU applied on the fp32 result)."""
import torch
import torch.nn as nn

from .. import ops

__all__ = ["MLPLayers"]


class MLPLayers(nn.Module):
    def __init__(self, units=(512, 512, 512), nonlin=None, dropout=0.1):
        super().__init__()
        self.linears = nn.ModuleList([nn.Linear(a, b) for a, b in zip(units[:-1], units[1:])])
        self.dropout = dropout

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        for i, lin in enumerate(self.linears):
            x = ops.gemm(x.to(torch.bfloat16).contiguous(), lin.weight.detach().to(torch.bfloat16).contiguous(), lin.bias.detach().float(), out_f32=True)
            if i + 1 < len(self.linears):
                x = torch.relu(x)
        return x

"""MaskedContrastiveLoss -- constructor / forward contract of avssl/module/losses.py:129-245, computed by
sc_infonce_fwd.  Differences from the reference that are NOT numerics: no MAX_EYE=256 batch limit (the mask
is generated in-kernel for any batch size; the `eye_mat*` buffers are still registered so reference checkpoints
load strictly)."""
import math

import torch
import torch.nn as nn

from .. import ops

MAX_EYE = 256


class MaskedContrastiveLoss(nn.Module):
    def __init__(self, temperature: float = 0.07, temperature_trainable: bool = False, margin: float = 0.0,
                 dcl: bool = False, a2b: bool = True, b2a: bool = True):
        super().__init__()
        assert a2b or b2a, "Cannot set both `a2b` and `b2a` to False."
        self.temperature_trainable = temperature_trainable
        self.margin, self.dcl, self.a2b, self.b2a = margin, dcl, a2b, b2a
        if temperature_trainable:
            self.temperature = nn.Parameter(torch.ones([]) * math.log(1 / temperature))
        else:
            self.temperature = 1 / temperature
        eye = torch.eye(MAX_EYE, dtype=torch.bool)
        self.register_buffer("eye_mat", eye)
        self.register_buffer("neg_eye_mat", ~eye)
        self.register_buffer("eye_mat_fl", eye.float())

    @property
    def current_temperature(self) -> float:
        if self.temperature_trainable:
            return float(self.temperature.data.detach().float().exp().item())
        return float(self.temperature)

    def forward(self, feat_A: torch.Tensor, feat_B: torch.Tensor, index: torch.LongTensor = None) -> torch.Tensor:
        assert feat_A.shape == feat_B.shape, (feat_A.shape, feat_B.shape)
        if index is not None:
            assert index.shape[0] == feat_A.shape[0], (index.shape, feat_A.shape)
        if torch.is_grad_enabled() and (feat_A.requires_grad or feat_B.requires_grad or (self.temperature_trainable and self.temperature.requires_grad)):
            from ..train_tail import MaskedContrastiveFn
            return MaskedContrastiveFn.apply(feat_A, feat_B, index, self.temperature if self.temperature_trainable else None,
                                             self.current_temperature, self.margin, self.dcl, self.a2b, self.b2a)
        out = ops.infonce(feat_A.float(), feat_B.float(), index, self.current_temperature, self.margin, self.dcl, self.a2b, self.b2a)
        return out[0]

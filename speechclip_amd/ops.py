"""Tensor-level wrappers over the C ABI (include/speechclip_hip.h).

Only plumbing lives here: argument checks, output allocation, pointer extraction.  All arithmetic
happens inside libspeechclip_hip.so.  Every wrapper requires CUDA(HIP) tensors.
"""
from typing import Optional

import torch

from . import _lib
from ._lib import ACT_GELU, ACT_NONE, ACT_QUICKGELU, GEMM_OUT_F32, check, lib, ptr, stream

bf16 = torch.bfloat16


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.SpeechClipHipError("speechclip_amd ops need device tensors (no CPU fallback)")


def gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, act: int = ACT_NONE,
         residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, out_f32: bool = False,
         M: Optional[int] = None, K: Optional[int] = None, lda: Optional[int] = None) -> torch.Tensor:
    """out[M,N] = act(a[M,K] @ w[N,K]^T + bias) + residual.  `M/K/lda` override the logical A view
    (overlapping rows: conv-as-GEMM)."""
    _need_cuda(a, w)
    assert a.dtype == bf16 and w.dtype == bf16 and w.dim() == 2 and w.is_contiguous()
    N, Kw = w.shape
    if M is None:
        assert a.dim() == 2 and a.stride(1) == 1
        M, K, lda = a.shape[0], a.shape[1], a.stride(0)
    assert K == Kw, (K, Kw)
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=torch.float32 if out_f32 else bf16)
    assert out.stride(-1) == 1 and out.dtype == (torch.float32 if out_f32 else bf16)
    if residual is not None:
        assert residual.dtype == out.dtype and residual.stride(-1) == 1
    flags = act | (GEMM_OUT_F32 if out_f32 else 0)
    check(lib().sc_gemm_bf16(ptr(a), lda, ptr(w), w.stride(0), ptr(out), out.stride(-2), ptr(bias), ptr(residual),
                             residual.stride(-2) if residual is not None else 0, M, N, K, flags, stream()), "sc_gemm_bf16")
    return out

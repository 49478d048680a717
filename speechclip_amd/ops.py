"""Tensor-level wrappers over the C ABI (include/speechclip_hip.h).

Only plumbing lives here: argument checks, output allocation, pointer extraction.  All arithmetic
happens inside libspeechclip_hip.so.  Every wrapper requires CUDA(HIP) tensors.
"""
from typing import Optional

import torch

from . import _lib
from ._lib import ACT_GELU, ACT_NONE, ACT_QUICKGELU, ATTN_CAUSAL, ATTN_F16, GEMM_F16, GEMM_OUT_F32, check, lib, ptr, stream

bf16 = torch.bfloat16

# Optional HIP-event instrumentation of the dominant kernel (bench.py roofline leg): when PROFILE is a list, every
# sc_gemm_bf16 launch appends (start_event, end_event, flops, shape_tag) recorded on the launch stream.
PROFILE = None
# HBM-bound segments (conv layer 0, LayerNorm, layer mix): when PROFILE_HBM is a list, each launch appends
# (start, end, algorithmic bytes, tag, stream) -- bench.py reports their GB/s against the 8 TB/s HBM peak
PROFILE_HBM = None


class _HbmSpan:
    __slots__ = ("tag", "nbytes", "e0")

    def __init__(self, tag, nbytes):
        self.tag, self.nbytes, self.e0 = tag, nbytes, None

    def __enter__(self):
        if PROFILE_HBM is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *a):
        if self.e0 is not None and PROFILE_HBM is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            PROFILE_HBM.append((self.e0, e1, float(self.nbytes), self.tag, torch.cuda.current_stream().cuda_stream))
        return False
PROFILE_TAG = "speech"   # which part of the step is launching ("speech" / "image" / "head"): set by KWClip_GeneralTransformer.forward, stored as entry[5]
PROFILE_SIDE = []     # (start, end) HIP events of every side-stream window (the image tower running beside the speech tower) while PROFILE is on

_GEMM_WS = {}
# The plain GEMMs (QKV / out-proj / fc2 / ViT projections) run on the hand-written gemm256_kernel like everything else.  hipBLASLt behind
# the same entry (csrc/vendor_gemm.hip) is a COMPARATOR only: SC_GEMM_VENDOR=1 in the environment, or set_vendor_gemm(True) at run time
# (bench.py measures it beside the headline number); it is never on by default.
import os as _os
_VENDOR_GEMM = [_os.environ.get("SC_GEMM_VENDOR", "0") == "1"]


def set_vendor_gemm(enabled: bool) -> None:
    """Switch the vendor-library comparator path for plain GEMMs on / off (registers / drops the library workspace)."""
    _VENDOR_GEMM[0] = bool(enabled)
    if not enabled and _GEMM_WS:
        torch.cuda.synchronize()
        check(lib().sc_set_gemm_workspace(None, 0), "sc_set_gemm_workspace")
        _GEMM_WS.clear()


def vendor_gemm_enabled() -> bool:
    return _VENDOR_GEMM[0]


def _ensure_gemm_workspace(dev):
    """Comparator path only: one 64 MiB scratch buffer per process (one process per GPU) for the library (sc_set_gemm_workspace)."""
    if _VENDOR_GEMM[0] and dev not in _GEMM_WS:
        ws = torch.empty(64 << 20, device=dev, dtype=torch.uint8)
        check(lib().sc_set_gemm_workspace(ptr(ws), ws.numel()), "sc_set_gemm_workspace")
        _GEMM_WS.clear()          # the library keeps ONE registration: a process that hops devices re-registers
        _GEMM_WS[dev] = ws


# Parameter epoch: bumped by every optimizer step that writes parameters through raw pointers (train_tail.FusedAdam -> sc_adam_step does not
# move torch's per-tensor version counter).  Every parameter-derived cache (bf16 weight casts, pooling operands, cosine / VQ tables) carries
# it in its key, so an eval forward after training steps never sees pre-step weights.
_PARAM_EPOCH = [0]


def param_epoch(*tensors) -> int:
    """Current epoch; with tensors given, -1 if none of them is trainable (a frozen tensor cannot be written by an optimizer step, so its
    derived tables -- the 49408-row sub-word tables -- are not rebuilt after every step)."""
    if tensors and not any(t.requires_grad for t in tensors):
        return -1
    return _PARAM_EPOCH[0]


def bump_param_epoch() -> None:
    _PARAM_EPOCH[0] += 1


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
    # operand format: bf16, or IEEE half for both operands (SC_GEMM_F16: the frozen pre-LN encoder layers; 16-bit outputs / residuals are half too)
    f16 = a.dtype == torch.float16
    op16 = torch.float16 if f16 else bf16
    assert a.dtype == op16 and w.dtype == op16 and w.dim() == 2 and w.is_contiguous(), (a.dtype, w.dtype)
    N, Kw = w.shape
    if M is None:
        assert a.dim() == 2 and a.stride(1) == 1
        M, K, lda = a.shape[0], a.shape[1], a.stride(0)
    assert K == Kw, (K, Kw)
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=torch.float32 if out_f32 else op16)
    assert out.stride(-1) == 1 and out.dtype == (torch.float32 if out_f32 else op16)
    if residual is not None:
        assert residual.dtype == out.dtype and residual.stride(-1) == 1
    flags = act | (GEMM_OUT_F32 if out_f32 else 0) | (GEMM_F16 if f16 else 0)
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _ensure_gemm_workspace(a.device)
    check(lib().sc_gemm_bf16(ptr(a), lda, ptr(w), w.stride(0), ptr(out), out.stride(-2), ptr(bias), ptr(residual),
                             residual.stride(-2) if residual is not None else 0, M, N, K, flags, stream()), "sc_gemm_bf16")
    if PROFILE is not None:
        e1.record()
        PROFILE.append((e0, e1, 2.0 * M * N * K, (M, N, K, act, residual is not None, out_f32, int(lib().sc_gemm_last_path())),
                        torch.cuda.current_stream().cuda_stream, PROFILE_TAG))
    return out


def gemm_batched(a, lda, stride_a, w, stride_w, w_mod, out, ldc, stride_c, bias, M, N, K, batch, act=ACT_NONE, ldw=None):
    """`batch` products out_z[M,N] = a_z[M,K] w_{z % w_mod}[N,K]^T (+ bias); operand z at base + z * stride (elements); `a`, `w`, `out` may be
    views whose data_ptr is operand 0.  out dtype f32 => fp32 outputs."""
    _need_cuda(a, w, out)
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    flags = act | (GEMM_OUT_F32 if out.dtype == torch.float32 else 0)
    check(lib().sc_gemm_bf16_batched(ptr(a), lda, stride_a, ptr(w), K if ldw is None else ldw, stride_w, w_mod, ptr(out), ldc, stride_c, ptr(bias),
                                     M, N, K, batch, flags, stream()), "sc_gemm_bf16_batched")
    if PROFILE is not None:
        e1.record()
        PROFILE.append((e0, e1, 2.0 * M * N * K * batch, (M, N, K, act, False, False, 0, batch),   # tag[6] = path (0 hand-written), tag[7] = batch
                        torch.cuda.current_stream().cuda_stream, PROFILE_TAG))
    return out


LN_IN_F32, LN_OUT_F32, LN_GELU, LN_OUT_F16 = 1, 2, 4, 8


def layernorm(x, gamma, beta, eps=1e-5, out=None, out_f32=False, gelu=False, rows=None, D=None, ld_in=None):
    """Row LayerNorm.  x: [..., D] (bf16 or f32, last dim contiguous); optional strided-row view via rows/D/ld_in."""
    _need_cuda(x)
    if rows is None:
        D = x.shape[-1]
        x2 = x.reshape(-1, D) if x.is_contiguous() else x
        assert x2.dim() == 2 and x2.stride(1) == 1
        rows, ld_in = x2.shape[0], x2.stride(0)
        shape = x.shape
    else:
        shape = (rows, D)
    if out is None:
        out = torch.empty(shape, device=x.device, dtype=torch.float32 if out_f32 else bf16)
    flags = (LN_IN_F32 if x.dtype == torch.float32 else 0) | (LN_OUT_F32 if out.dtype == torch.float32 else 0) | (LN_GELU if gelu else 0)
    if out.dtype == torch.float16:           # IEEE-half operand format of the pre-LN encoder layers (fp32 residual stream in)
        assert x.dtype == torch.float32
        flags |= LN_OUT_F16
    assert x.dtype in (bf16, torch.float32) and out.dtype in (bf16, torch.float16, torch.float32)
    with _HbmSpan("layernorm", rows * D * (x.element_size() + out.element_size())):
        check(lib().sc_layernorm(ptr(x), ld_in, ptr(gamma), ptr(beta), ptr(out), D, rows, D, eps, flags, stream()), "sc_layernorm")
    return out


def weighted_sum(hidden, weights, normalize=False, eps=1e-5):
    """hidden: [n, rows, D] contiguous (bf16 or f32); weights f32 [n] (pre-softmax).  Returns bf16 [rows, D]."""
    _need_cuda(hidden, weights)
    n, rows, D = hidden.shape
    assert hidden.is_contiguous() and weights.dtype == torch.float32
    out = torch.empty(rows, D, device=hidden.device, dtype=bf16)
    flags = (1 if normalize else 0) | (2 if hidden.dtype == torch.float32 else 0)
    with _HbmSpan("layer_mix", rows * D * (n * hidden.element_size() + 2)):
        check(lib().sc_weighted_sum_fwd(ptr(hidden), rows * D, ptr(weights), ptr(out), n, rows, D, flags, eps, stream()), "sc_weighted_sum_fwd")
    return out


def hidden_normalize_(hidden, T, method):
    """In place: hidden [n, B, Tp, D] (bf16 / f32, contiguous) normalised as speech_encoder_plus.py:572-592 does for normalize_type "method1" / "method2"."""
    _need_cuda(hidden)
    assert hidden.dim() == 4 and hidden.is_contiguous() and hidden.dtype in (bf16, torch.float32) and method in ("method1", "method2")
    n, B, Tp, D = hidden.shape
    m = 1 if method == "method1" else 2
    ws = torch.empty(n * B * (Tp + 1), device=hidden.device, dtype=torch.float32) if m == 2 else None
    check(lib().sc_hidden_normalize(ptr(hidden), int(hidden.dtype == torch.float32), n, B, Tp, int(T), D, m, ptr(ws), stream()), "sc_hidden_normalize")
    return hidden


def gemm_splitk(a16, w16, K_chunk, bias=None, act=ACT_NONE, residual=None):
    """f32 [M, N] = act(a16 [M, K] @ w16 [N, K]^T + bias) + residual as a DETERMINISTIC split-K product: the K / K_chunk chunks run as one batched
    GEMM into fp32 partials [S, M, N] (S x more tiles for few-row, deep-K shapes), sc_splitk_reduce_f32 sums them in fixed order."""
    _need_cuda(a16, w16)
    assert a16.dtype == bf16 and w16.dtype == bf16 and a16.dim() == 2 and a16.is_contiguous() and w16.is_contiguous()
    M, K = a16.shape
    N = w16.shape[0]
    assert w16.shape[1] == K and K % K_chunk == 0 and K_chunk % 64 == 0 and N % 4 == 0
    S = K // K_chunk
    part = torch.empty(S, M, N, device=a16.device, dtype=torch.float32)
    gemm_batched(a16, K, K_chunk, w16, K_chunk, S, part, N, M * N, None, M, N, K_chunk, S, ldw=K)
    out = torch.empty(M, N, device=a16.device, dtype=torch.float32)
    if residual is not None:
        assert residual.dtype == torch.float32 and residual.stride(-1) == 1
    check(lib().sc_splitk_reduce_f32(ptr(part), S, M, N, ptr(bias), ptr(residual), residual.stride(-2) if residual is not None else 0, ptr(out), int(act),
                                     stream()), "sc_splitk_reduce_f32")
    return out


def l2norm(x, clamp=False):
    """x / ||x|| (no eps: kwClip.py:1436); clamp=True: x / max(||x||, 1e-8) (the operand normalisation of F.cosine_similarity)."""
    _need_cuda(x)
    assert x.dim() == 2 and x.stride(1) == 1
    out = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    check(lib().sc_l2norm_fwd(ptr(x), x.stride(0), ptr(out), x.shape[0], x.shape[1], int(x.dtype == torch.float32) | (2 if clamp else 0), stream()), "sc_l2norm_fwd")
    return out


def wave_layernorm(wav, lens_i32, eps=1e-5):
    _need_cuda(wav, lens_i32)
    assert wav.dtype == torch.float32 and wav.is_contiguous() and lens_i32.dtype == torch.int32
    out = torch.empty_like(wav)
    check(lib().sc_wave_layernorm(ptr(wav), ptr(out), ptr(lens_i32), wav.shape[0], wav.shape[1], eps, stream()), "sc_wave_layernorm")
    return out


def attention(qkv, B, T, H, klens_i32=None, out=None, scale=None, causal=False):
    """qkv: bf16 (or IEEE half: SC_ATTN_F16) [B*T, 3*H*64] packed (q|k|v); returns the same format [B*T, H*64]."""
    _need_cuda(qkv)
    D = H * 64
    assert qkv.dtype in (bf16, torch.float16) and qkv.shape == (B * T, 3 * D) and qkv.is_contiguous()
    if out is None:
        out = torch.empty(B * T, D, device=qkv.device, dtype=qkv.dtype)
    assert out.dtype == qkv.dtype
    esz = 2
    check(lib().sc_attention_fwd(qkv.data_ptr(), qkv.data_ptr() + D * esz, qkv.data_ptr() + 2 * D * esz, ptr(out), ptr(klens_i32),
                                 B, H, T, 64, 3 * D, D, 0.125 if scale is None else scale, (ATTN_CAUSAL if causal else 0) | (ATTN_F16 if qkv.dtype == torch.float16 else 0),
                                 stream()), "sc_attention_fwd")
    return out


def attention_dropout(qkv, B, T, H, klens_i32, drop_p, seed, out=None):
    """ops.attention with dropout on the attention probabilities (train-mode frozen encoder); drop_p == 0 is the plain kernel."""
    _need_cuda(qkv)
    D = H * 64
    assert qkv.dtype == bf16 and qkv.shape == (B * T, 3 * D) and qkv.is_contiguous()
    if out is None:
        out = torch.empty(B * T, D, device=qkv.device, dtype=bf16)
    check(lib().sc_attention_fwd_dropout(qkv.data_ptr(), qkv.data_ptr() + D * 2, qkv.data_ptr() + 2 * D * 2, ptr(out), ptr(klens_i32), B, H, T, 64,
                                         3 * D, D, 0.125, 0, float(drop_p), int(seed) & 0xffffffff, stream()), "sc_attention_fwd_dropout")
    return out


def dropout_bf16(x, drop_p, seed, residual=None, out=None):
    """out = [residual +] dropout(x) (bf16, counter-based mask from `seed`); out=x runs in place."""
    _need_cuda(x)
    assert x.dtype == bf16 and x.is_contiguous() and (residual is None or (residual.dtype == bf16 and residual.is_contiguous() and residual.shape == x.shape))
    if out is None:
        out = torch.empty_like(x)
    check(lib().sc_dropout_bf16(ptr(x), ptr(residual), ptr(out), x.numel(), float(drop_p), int(seed) & 0xffffffff, stream()), "sc_dropout_bf16")
    return out


def dropout_add_layernorm(x, residual, gamma, beta, drop_p, seed, eps=1e-5, out=None):
    """out = LayerNorm(residual + dropout(x)) for bf16 [rows, D]: one fused pass for D = 768, else sc_dropout_bf16 (into x, in place) + layernorm."""
    _need_cuda(x, residual)
    assert x.dtype == bf16 and residual.dtype == bf16 and x.is_contiguous() and residual.is_contiguous() and x.shape == residual.shape and x.dim() == 2
    if out is None:
        out = torch.empty_like(x)
    rc = lib().sc_dropout_add_layernorm_bf16(ptr(x), ptr(residual), ptr(gamma), ptr(beta), ptr(out), x.shape[0], x.shape[1], eps, float(drop_p),
                                             int(seed) & 0xffffffff, stream())
    if rc == 1:
        dropout_bf16(x, drop_p, seed, residual=residual, out=x)
        return layernorm(x, gamma, beta, eps, out=out)
    check(rc, "sc_dropout_add_layernorm_bf16")
    return out


def attention_rows(qkv, B, L, H, hd, key_padding_mask=None, scale=None):
    """Full-row MHA for any head dim: qkv bf16 [B*L, 3*H*hd] packed (q|k|v); key_padding_mask bool/uint8 [B, L] (True = padding) or None.
    Returns bf16 [B*L, H*hd] (heads concatenated, before out_proj)."""
    _need_cuda(qkv, key_padding_mask)
    D = H * hd
    assert qkv.dtype == bf16 and qkv.shape == (B * L, 3 * D) and qkv.is_contiguous()
    m = None
    if key_padding_mask is not None:
        assert key_padding_mask.shape == (B, L)
        m = key_padding_mask.to(torch.uint8).contiguous()
    out = torch.empty(B * L, D, device=qkv.device, dtype=bf16)
    check(lib().sc_attention_rows_fwd(qkv.data_ptr(), qkv.data_ptr() + D * 2, qkv.data_ptr() + 2 * D * 2, ptr(out), ptr(m), B, H, L, hd, 3 * D, D,
                                      hd ** -0.5 if scale is None else scale, stream()), "sc_attention_rows_fwd")
    return out


def attention_probs(qkv, B, L, H, hd, key_padding_mask=None, n_rows=None, scale=None):
    """Per-head attention probabilities (need_weights=True, average_attn_weights=False) of the first `n_rows` query rows (default: all):
    qkv bf16 [B*L, 3*H*hd] packed (q|k|v) -> f32 [B, H, n_rows, L]; exactly 0 at padded keys."""
    _need_cuda(qkv, key_padding_mask)
    D = H * hd
    assert qkv.dtype == bf16 and qkv.shape == (B * L, 3 * D) and qkv.is_contiguous()
    n_rows = L if n_rows is None else int(n_rows)
    m = None
    if key_padding_mask is not None:
        assert key_padding_mask.shape == (B, L)
        m = key_padding_mask.to(torch.uint8).contiguous()
    probs = torch.empty(B, H, n_rows, L, device=qkv.device, dtype=torch.float32)
    check(lib().sc_attention_probs_fwd(qkv.data_ptr(), qkv.data_ptr() + D * 2, ptr(probs), ptr(m), B, H, L, hd, n_rows, 3 * D,
                                       hd ** -0.5 if scale is None else scale, stream()), "sc_attention_probs_fwd")
    return probs


def topk_rows(x, k):
    """torch.topk(x, k, dim=-1) for f32 [..., V] on the device: (values f32 [..., k] descending, indices i64 [..., k]); ties -> lowest index."""
    _need_cuda(x)
    x2 = x.float().contiguous().view(-1, x.shape[-1])
    R, V = x2.shape
    vals = torch.empty(R, k, device=x.device, dtype=torch.float32)
    idx = torch.empty(R, k, device=x.device, dtype=torch.int32)
    check(lib().sc_topk_rows_f32(ptr(x2), V, R, V, int(k), ptr(vals), ptr(idx), stream()), "sc_topk_rows_f32")
    return vals.view(*x.shape[:-1], k), idx.long().view(*x.shape[:-1], k)


def cls_attention(cls_qkv, kv_x, lens_i32, B, T, NQ, H, hd):
    """cls_qkv bf16 [NQ, 3D]; kv_x bf16 [B*T, 2D]; returns bf16 [B, NQ, D]."""
    _need_cuda(cls_qkv, kv_x)
    D = H * hd
    assert cls_qkv.shape == (NQ, 3 * D) and kv_x.shape == (B * T, 2 * D) and cls_qkv.is_contiguous() and kv_x.is_contiguous()
    out = torch.empty(B, NQ, D, device=kv_x.device, dtype=bf16)
    check(lib().sc_cls_attention_fwd(ptr(cls_qkv), ptr(kv_x), 2 * D, ptr(lens_i32), ptr(out), B, T, NQ, H, hd, hd ** -0.5, stream()),
          "sc_cls_attention_fwd")
    return out


def cls_pool(x_rows, cls16, scores, cls_scores, lens_i32, B, T, NQ, R, D, split=0):
    """x_rows bf16 [B*T, D]; cls16 bf16 [NQ, D]; scores f32 [B*T, R]; cls_scores f32 [NQ, R] -> xbar bf16 [B, R, D], or with split = 2 / 3 the
    (hi | lo [| hi]) blocks bf16 [B, R, split*D] that keep the fp32 pooled sums to ~16 bits (operand of a depth-split*D GEMM against [W | W] /
    [W_hi | W_hi | W_lo])."""
    _need_cuda(x_rows, cls16, scores, cls_scores)
    assert x_rows.dtype == bf16 and x_rows.stride(1) == 1 and scores.dtype == torch.float32 and scores.shape == (B * T, R) and scores.is_contiguous()
    xbar = torch.empty(B, R, (split or 1) * D, device=x_rows.device, dtype=bf16)
    if split:
        check(lib().sc_cls_pool_fwd_split(ptr(x_rows), x_rows.stride(0), ptr(cls16), ptr(scores), ptr(cls_scores), ptr(lens_i32), ptr(xbar), B, T, NQ, R, D,
                                          int(split), stream()), "sc_cls_pool_fwd_split")
    else:
        check(lib().sc_cls_pool_fwd(ptr(x_rows), x_rows.stride(0), ptr(cls16), ptr(scores), ptr(cls_scores), ptr(lens_i32), ptr(xbar), B, T, NQ, R, D,
                                    stream()), "sc_cls_pool_fwd")
    return xbar


def conv0(wav, w, T0, P, gn_gamma=None, gn_beta=None, bias=None, eps=1e-5, out=None, ln_coef=None):
    """HuBERT conv layer 0.  wav f32 [B, L]; w f32 [C, 10].  GroupNorm+GELU if gn_gamma given, else raw conv + bias -- followed, when ln_coef
    (f32 [2 C + 1] = gamma | beta | eps) is given, by the LayerNorm over the channels + GELU of a "layer_norm" extractor in the same kernel.
    Returns channels-last bf16 [B, P, C] (+ (k-s) slack rows so the next conv-as-GEMM may over-read)."""
    _need_cuda(wav, w)
    B, L = wav.shape
    C = w.shape[0]
    assert wav.dtype == torch.float32 and wav.is_contiguous() and w.dtype == torch.float32 and w.is_contiguous()
    buf = out if out is not None else torch.zeros(B * P + 8, C, device=wav.device, dtype=bf16)
    assert buf.dtype == bf16 and buf.is_contiguous() and buf.shape[0] >= B * P and buf.shape[1] == C
    wfrag = torch.empty(lib().sc_conv0_wfrag_workspace_bytes(B), device=wav.device, dtype=torch.uint8)
    if gn_gamma is not None:
        ws = torch.empty(lib().sc_conv0_stats_workspace_bytes(B), device=wav.device, dtype=torch.uint8)
        coef = torch.empty(B, C, 2, device=wav.device, dtype=torch.float32)
        with _HbmSpan("conv0_gn_stats", B * L * 4):
            check(lib().sc_conv0_gn_coef(ptr(wav), L, ptr(w), ptr(gn_gamma), ptr(gn_beta), ptr(ws), ptr(coef), B, C, T0, eps, stream()), "sc_conv0_gn_coef")
        with _HbmSpan("conv0", B * L * 4 + B * P * C * 2):
            check(lib().sc_conv0_fwd(ptr(wav), L, L, ptr(w), None, ptr(coef), ptr(buf), B, C, T0, P, 0, ptr(wfrag), stream()), "sc_conv0_fwd")
    else:
        assert ln_coef is None or (ln_coef.dtype == torch.float32 and ln_coef.numel() == 2 * C + 1 and ln_coef.is_contiguous())
        with _HbmSpan("conv0", B * L * 4 + B * P * C * 2):
            check(lib().sc_conv0_fwd(ptr(wav), L, L, ptr(w), ptr(bias), ptr(ln_coef), ptr(buf), B, C, T0, P, 1 if ln_coef is None else 2, ptr(wfrag), stream()), "sc_conv0_fwd")
    return buf


def posconv(x, valid_i32, wg, bias, gamma, beta, B, Tp, D, G, Kw, out=None, out_f32=False, eps=1e-5):
    """x bf16 [B*Tp, D]; wg bf16 [G, D/G, Kw*D/G] (folded weight-norm, K index = tap*cg + c_in)."""
    _need_cuda(x, wg)
    cg = D // G
    conv = torch.empty(B * G * Tp * cg, device=x.device, dtype=bf16)
    rc = lib().sc_posconv_conv(ptr(x), ptr(valid_i32), ptr(wg), ptr(conv), B, Tp, D, G, Kw, stream())
    if rc == 1:      # group width not covered by the windowed kernel: pack the sliding windows and use the batched GEMM
        xg = torch.empty(B * G * (Tp + Kw) * cg + 64, device=x.device, dtype=bf16)
        check(lib().sc_posconv_pack(ptr(x), ptr(valid_i32), ptr(xg), B, Tp, D, G, Kw, stream()), "sc_posconv_pack")
        gemm_batched(xg, cg, (Tp + Kw) * cg, wg, cg * Kw * cg, G, conv, cg, Tp * cg, None, Tp, cg, Kw * cg, B * G)
    else:
        check(rc, "sc_posconv_conv")
    if out is None:
        out = torch.empty(B * Tp, D, device=x.device, dtype=torch.float32 if out_f32 else bf16)
    check(lib().sc_posconv_finish(ptr(x), ptr(valid_i32), ptr(conv), ptr(bias), ptr(gamma), ptr(beta), ptr(out), B, Tp, D, G,
                                  int(out.dtype == torch.float32), eps, stream()), "sc_posconv_finish")
    return out


# ---------------------------------------------------------------------------------------------- padding-free (packed) batches
# Utterance b owns rows [row_off[b], row_off[b + 1]) of every transformer-level tensor (module/hubert.py: extract_all_layers_packed).
def conv0_packed(wav, w, T0, row_off_i32, row_scale, rows_max, total_rows, gn_gamma=None, gn_beta=None, bias=None, eps=1e-5, out=None, ln_coef=None):
    """ops.conv0 writing utterance b at rows row_scale * row_off[b] ...; the GroupNorm statistics are those of the padded length T0."""
    _need_cuda(wav, w, row_off_i32)
    B, L = wav.shape
    C = w.shape[0]
    assert wav.dtype == torch.float32 and wav.is_contiguous() and w.dtype == torch.float32 and w.is_contiguous() and row_off_i32.dtype == torch.int32
    assert out is not None and out.dtype == bf16 and out.is_contiguous() and out.shape[0] >= row_scale * total_rows and out.shape[1] == C
    wfrag = torch.empty(lib().sc_conv0_wfrag_workspace_bytes(B), device=wav.device, dtype=torch.uint8)
    coef = None
    if gn_gamma is not None:
        ws = torch.empty(lib().sc_conv0_stats_workspace_bytes(B), device=wav.device, dtype=torch.uint8)
        coef = torch.empty(B, C, 2, device=wav.device, dtype=torch.float32)
        with _HbmSpan("conv0_gn_stats", B * L * 4):
            check(lib().sc_conv0_gn_coef(ptr(wav), L, ptr(w), ptr(gn_gamma), ptr(gn_beta), ptr(ws), ptr(coef), B, C, T0, eps, stream()), "sc_conv0_gn_coef")
    mode = 0 if gn_gamma is not None else 1
    if ln_coef is not None:      # conv + bias -> LayerNorm over the channels -> GELU in the same kernel (sc_conv0_fwd mode 2)
        assert gn_gamma is None and ln_coef.dtype == torch.float32 and ln_coef.numel() == 2 * C + 1 and ln_coef.is_contiguous()
        coef, mode = ln_coef, 2
    with _HbmSpan("conv0", row_scale * total_rows * (5 * 4 + C * 2)):
        check(lib().sc_conv0_fwd_packed(ptr(wav), L, L, ptr(w), None if gn_gamma is not None else ptr(bias), ptr(coef), ptr(out), B, C, T0,
                                        ptr(row_off_i32), row_scale, row_scale * rows_max, mode, ptr(wfrag), stream()),
              "sc_conv0_fwd_packed")
    return out


def posconv_packed(x, valid_i32, row_off_i32, wg, bias, gamma, beta, B, rows_max, total_rows, D, G, Kw, out=None, out_f32=False, eps=1e-5):
    """ops.posconv over packed rows: x bf16 [total_rows, D]."""
    _need_cuda(x, wg, row_off_i32)
    cg = D // G
    conv = torch.empty(total_rows * D, device=x.device, dtype=bf16)
    rc = lib().sc_posconv_conv_packed(ptr(x), ptr(valid_i32), ptr(row_off_i32), ptr(wg), ptr(conv), B, rows_max, D, G, Kw, stream())
    if rc == 1:
        raise SpeechClipHipError(f"packed batches need the windowed positional-conv kernel (D/G in 32/48/64), got D/G = {cg}")
    check(rc, "sc_posconv_conv_packed")
    if out is None:
        out = torch.empty(total_rows, D, device=x.device, dtype=torch.float32 if out_f32 else bf16)
    check(lib().sc_posconv_finish_packed(ptr(x), ptr(valid_i32), ptr(row_off_i32), ptr(conv), ptr(bias), ptr(gamma), ptr(beta), ptr(out), B, total_rows, D, G,
                                         int(out.dtype == torch.float32), eps, stream()), "sc_posconv_finish_packed")
    return out


def attention_packed(qkv, B, rows_max, H, klens_i32, row_off_i32, out=None, drop_p=0.0, seed=0):
    """ops.attention / ops.attention_dropout over packed rows: qkv bf16 [total_rows, 3*H*64]."""
    _need_cuda(qkv, klens_i32, row_off_i32)
    D = H * 64
    total = qkv.shape[0]
    assert qkv.dtype in (bf16, torch.float16) and qkv.shape == (total, 3 * D) and qkv.is_contiguous()
    if out is None:
        out = torch.empty(total, D, device=qkv.device, dtype=qkv.dtype)
    assert out.dtype == qkv.dtype
    check(lib().sc_attention_fwd_packed(qkv.data_ptr(), qkv.data_ptr() + D * 2, qkv.data_ptr() + 2 * D * 2, ptr(out), ptr(klens_i32), ptr(row_off_i32),
                                        B, H, rows_max, total, 64, 3 * D, D, 0.125, float(drop_p), int(seed) & 0xffffffff,
                                        ATTN_F16 if qkv.dtype == torch.float16 else 0, stream()), "sc_attention_fwd_packed")
    return out


def unpack_rows(src, row_off_i32, B, T_out, halo=0):
    """src [n, total_rows, D] or [total_rows, D] (packed) -> [n, B, T_out, D] / [B, T_out, D], zeros beyond each utterance's rows; the last `halo`
    rows of every utterance are not copied (the packed engine's receptive-field row)."""
    _need_cuda(src, row_off_i32)
    three = src.dim() == 3
    s3 = src if three else src.unsqueeze(0)
    n, total, D = s3.shape
    assert s3.is_contiguous() and (D * s3.element_size()) % 16 == 0
    out = torch.empty(n, B, T_out, D, device=src.device, dtype=src.dtype)
    rb = D * s3.element_size()
    check(lib().sc_unpack_rows(ptr(s3), total * rb, ptr(row_off_i32), ptr(out), B * T_out * rb, n, B, T_out, rb, int(halo), stream()), "sc_unpack_rows")
    return out if three else out[0]


_DEV_INTS = {}


def dev_ints(values, dtype, device):
    """Small host integer lists (utterance lengths, valid-frame counts) as a device tensor, cached by value: fixed-length batches upload
    them once instead of every step, and a step that only replays cached uploads can be captured in a HIP graph (a pageable
    host-to-device copy is not capturable).  The returned tensor is shared: callers must not write to it."""
    key = (tuple(int(v) for v in values), dtype, str(device))
    t = _DEV_INTS.get(key)
    if t is None:
        if len(_DEV_INTS) > 256:
            _DEV_INTS.clear()
        t = torch.tensor(list(key[0]), dtype=dtype).to(device)
        _DEV_INTS[key] = t
    return t


def crop_pad(wav, starts, lens, Lout):
    """wav f32 [B, L] (device); starts/lens host int lists -> f32 [B, Lout]: out[b, j] = wav[b, starts[b] + j] for j < lens[b], else 0."""
    _need_cuda(wav)
    assert wav.dtype == torch.float32 and wav.dim() == 2 and wav.stride(1) == 1
    B = wav.shape[0]
    assert all(0 <= s and s + l <= wav.shape[1] and l <= Lout for s, l in zip(starts, lens))
    meta = torch.tensor([list(starts), list(lens)], dtype=torch.int32).to(wav.device)      # random crop offsets: new every step
    out = torch.empty(B, Lout, device=wav.device, dtype=torch.float32)
    check(lib().sc_crop_pad(ptr(wav), wav.stride(0), ptr(meta[0]), ptr(meta[1]), ptr(out), B, Lout, stream()), "sc_crop_pad")
    return out


def vit_patchify(img, p, Kpad):
    _need_cuda(img)
    B, C, R, _ = img.shape
    assert C == 3 and img.dtype == torch.float32 and img.is_contiguous()
    npatch = (R // p) ** 2
    cols = torch.empty(B * npatch, Kpad, device=img.device, dtype=bf16)
    check(lib().sc_vit_patchify(ptr(img), ptr(cols), B, R, p, Kpad, stream()), "sc_vit_patchify")
    return cols


def vit_embed(patch, cls, pos, gamma, beta, B, ntok, D, eps=1e-5):
    out = torch.empty(B * ntok, D, device=patch.device, dtype=torch.float32)
    check(lib().sc_vit_embed(ptr(patch), ptr(cls), ptr(pos), ptr(gamma), ptr(beta), ptr(out), B, ntok, D, eps, stream()), "sc_vit_embed")
    return out


def infonce(feat_a, feat_b, ids=None, inv_temperature=1.0 / 0.07, margin=0.0, dcl=False, a2b=True, b2a=True):
    """Returns f32[3] device tensor: (loss, a2b term, b2a term)."""
    _need_cuda(feat_a, feat_b)
    assert feat_a.shape == feat_b.shape and feat_a.dtype == torch.float32 and feat_b.dtype == torch.float32
    feat_a, feat_b = feat_a.contiguous(), feat_b.contiguous()
    Bg, E = feat_a.shape
    if ids is not None:
        assert ids.dtype == torch.int64 and ids.shape[0] == Bg
        ids = ids.contiguous()
    ws = torch.empty(lib().sc_infonce_workspace_bytes(Bg), device=feat_a.device, dtype=torch.uint8)
    out = torch.empty(3, device=feat_a.device, dtype=torch.float32)
    check(lib().sc_infonce_fwd(ptr(feat_a), ptr(feat_b), ptr(ids), ptr(ws), ptr(out), Bg, E, inv_temperature, margin, int(dcl), int(a2b),
                               int(b2a), stream()), "sc_infonce_fwd")
    return out


def kw_affine(x, scale, shift):
    """x f32 [B,K,D]; scale/shift f32 [K,D]."""
    _need_cuda(x)
    B, K, D = x.shape
    x = x.float().contiguous()
    out = torch.empty_like(x)
    check(lib().sc_kw_affine(ptr(x), ptr(scale.to(x.device)), ptr(shift.to(x.device)), ptr(out), B * K, K, D, stream()), "sc_kw_affine")
    return out


_COS_TABLES = {}


def _cos_table(emb):
    """(e/|e|) of the frozen sub-word table as a three-block bf16 operand [V, 3E] = (hi | hi | lo): with the keyword side split as (hi | lo | hi),
    ONE bf16 MFMA GEMM of depth 3E returns a_hi.e_hi + a_lo.e_hi + a_hi.e_lo (~16 mantissa bits of the fp32 product).  Cached per table version."""
    import weakref
    key = emb.data_ptr()
    hit = _COS_TABLES.get(key)
    if hit is not None and hit[0] == (emb._version, param_epoch(emb), tuple(emb.shape)) and hit[2]() is emb:
        return hit[1]
    en = l2norm(emb.detach().float().contiguous(), clamp=True)
    hi = en.to(bf16)
    lo = (en - hi.float()).to(bf16)
    tab = torch.cat([hi, hi, lo], dim=1).contiguous()
    _COS_TABLES.clear()
    _COS_TABLES[key] = ((emb._version, param_epoch(emb), tuple(emb.shape)), tab, weakref.ref(emb))
    return tab


def cosine_scores(a, emb, eps=1e-8, exact=None):
    """a f32 [R,E], emb f32 [V,E] -> f32 [R,V] cosine similarities.
    Large problems (the 49408-entry sub-word table) run on the MFMA GEMM with three-term bf16 splits of the normalised operands (~1e-5
    accurate) followed by sc_cosine_refine, which recomputes every entry within 1e-3 of its row maximum in fp32 -- the arg-max is decided
    by fp32 arithmetic either way.  Small problems / `exact=True`: the fp32 SIMT kernel for the whole matrix."""
    _need_cuda(a, emb)
    a = a.float().contiguous()
    R, E = a.shape
    V = emb.shape[0]
    if exact is None:
        exact = not (E % 64 == 0 and V % 4 == 0 and R * V >= (1 << 22))
    out = torch.empty(R, V, device=a.device, dtype=torch.float32)
    if exact:
        embf = emb.detach().float().contiguous()
        ws = torch.empty(lib().sc_cosine_workspace_bytes(R, V), device=a.device, dtype=torch.uint8)
        check(lib().sc_cosine_scores(ptr(a), ptr(embf), ptr(ws), ptr(out), R, V, E, eps, stream()), "sc_cosine_scores")
        return out
    tab = _cos_table(emb)
    a3 = split_hilo(l2norm(a, clamp=True), nblk=3)
    gemm(a3, tab, out=out, out_f32=True)
    embf = emb.detach()
    embf = embf if (embf.dtype == torch.float32 and embf.is_contiguous()) else embf.float().contiguous()
    check(lib().sc_cosine_refine(ptr(out), ptr(a), ptr(embf), R, V, E, 1e-3, eps, stream()), "sc_cosine_refine")
    return out


def vq_fwd(scores, K, mask_ids=(0, 2, 3)):
    """scores f32 [R,V] -> (targets i64 [R], stats f32 [2] = (code_perplexity, prob_perplexity), ent_per_t f32 [K])."""
    import ctypes
    _need_cuda(scores)
    scores = scores.float().contiguous()
    R, V = scores.shape
    dev = scores.device
    targets = torch.empty(R, device=dev, dtype=torch.int64)
    stats = torch.empty(2, device=dev, dtype=torch.float32)
    ent = torch.empty(K, device=dev, dtype=torch.float32)
    ws = torch.empty(lib().sc_vq_workspace_bytes(R, V), device=dev, dtype=torch.uint8)
    ids = (ctypes.c_int32 * len(mask_ids))(*[int(i) for i in mask_ids])
    check(lib().sc_vq_fwd(ptr(scores), ptr(targets), ptr(stats), ptr(ent), ptr(ws), R, K, V, ctypes.cast(ids, ctypes.c_void_p), len(mask_ids),
                          stream()), "sc_vq_fwd")
    return targets, stats, ent


def gather_rows(src, idx):
    _need_cuda(src, idx)
    src = src.detach().float().contiguous()
    idx = idx.to(torch.int64).contiguous()
    out = torch.empty(idx.shape[0], src.shape[1], device=src.device, dtype=torch.float32)
    check(lib().sc_gather_rows(ptr(src), ptr(idx), ptr(out), idx.shape[0], src.shape[1], stream()), "sc_gather_rows")
    return out


# ---------------------------------------------------------------------------------------------------- trainable tail (fp32)
def _f32c(*ts):
    for t in ts:
        if t is not None:
            assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), (t.dtype, t.is_contiguous())


def sgemm(a, b, transa=False, transb=False, alpha=1.0, beta=0.0, out=None, bias=None):
    """out[M,N] = alpha * op(a) @ op(b) + beta * out (+ bias[N]).  fp32, 2-D row-major tensors (last stride 1).
    transa: a stored [K,M]; transb: b stored [N,K] (nn.Linear weight layout)."""
    _need_cuda(a, b)
    assert a.dtype == torch.float32 and b.dtype == torch.float32 and a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1
    M, K = (a.shape[1], a.shape[0]) if transa else a.shape
    N = b.shape[0] if transb else b.shape[1]
    assert (b.shape[1] if transb else b.shape[0]) == K, (a.shape, b.shape, transa, transb)
    if out is None:
        assert beta == 0.0
        out = torch.empty(M, N, device=a.device, dtype=torch.float32)
    assert out.shape == (M, N) and out.dtype == torch.float32 and out.stride(1) == 1
    check(lib().sc_sgemm(int(transa), int(transb), M, N, K, alpha, ptr(a), a.stride(0), ptr(b), b.stride(0), beta, ptr(out), out.stride(0),
                         ptr(bias), stream()), "sc_sgemm")
    return out


def sgemm_batched(M, N, K, a, lda, stride_a, b, ldb, stride_b, out, ldc, stride_c, batch, transa=False, transb=False, alpha=1.0, beta=0.0,
                  bias=None, stride_bias=0):
    """`batch` products out_i[M,N] = alpha op(a_i) op(b_i) + beta out_i (+ bias_i) with operand i at base + i*stride (elements): the
    per-head products of the attention block in one launch.  a / b / out / bias are tensors (or views) whose data_ptr is operand 0."""
    _need_cuda(a, b, out)
    assert a.dtype == torch.float32 and b.dtype == torch.float32 and out.dtype == torch.float32
    check(lib().sc_sgemm_batched(int(transa), int(transb), M, N, K, alpha, ptr(a), lda, stride_a, ptr(b), ldb, stride_b, beta, ptr(out), ldc, stride_c,
                                 ptr(bias), stride_bias, batch, stream()), "sc_sgemm_batched")
    return out


def cls_pool_train_fwd(x_rows, cls_tok, scores, cls_scores, lens_i32, B, T, NQ, R, D, drop_p=0.0, seed=0):
    """-> (p f32 [B,R,NQ+T], xbar f32 [B,R,D]).  x_rows bf16 [B*T, D]; cls_tok f32 [NQ,D]; scores f32 [B*T,R]; cls_scores f32 [NQ,R]."""
    _need_cuda(x_rows)
    _f32c(cls_tok, scores, cls_scores)
    assert x_rows.dtype == bf16 and x_rows.stride(1) == 1 and scores.shape == (B * T, R)
    p = torch.empty(B, R, NQ + T, device=x_rows.device, dtype=torch.float32)
    xbar = torch.empty(B, R, D, device=x_rows.device, dtype=torch.float32)
    check(lib().sc_cls_pool_train_fwd(ptr(x_rows), x_rows.stride(0), ptr(cls_tok), ptr(scores), ptr(cls_scores), ptr(lens_i32), ptr(p), ptr(xbar),
                                      B, T, NQ, R, D, float(drop_p), int(seed) & 0xFFFFFFFF, stream()), "sc_cls_pool_train_fwd")
    return p, xbar


def cls_pool_bwd(x_rows, cls_tok, hidden, p, dzbar, u, lens_i32, B, T, NQ, R, D, normalize=False, drop_p=0.0, seed=0, nsplit=None, return_ws=False):
    """-> (du f32 [B*S,R,D], dcls_key f32 [B*S,NQ,D], dalpha f32 [B*S,n] or None): PARTIAL rows (S = nsplit key-splits per utterance); the
    caller sums over rows.  hidden: bf16 / f32 [n, B*T, D] contiguous or None."""
    _need_cuda(x_rows)
    _f32c(cls_tok, p, dzbar, u)
    dev = x_rows.device
    n = 0 if hidden is None else hidden.shape[0]
    if hidden is not None:
        assert hidden.dtype in (bf16, torch.float32) and hidden.is_contiguous() and hidden.shape[1] == B * T and hidden.shape[2] == D
    if nsplit is None:
        nsplit = max(1, min(16, 512 // max(B, 1)))           # ~2 blocks of 8 waves per CU
    ds = torch.empty(B, R, NQ + T, device=dev, dtype=torch.float32)
    pp = torch.empty_like(ds)
    du = torch.empty(B * nsplit, R, D, device=dev, dtype=torch.float32)
    dck = torch.empty(B * nsplit, NQ, D, device=dev, dtype=torch.float32)
    dalpha = torch.empty(B * nsplit, n, device=dev, dtype=torch.float32) if n else None
    check(lib().sc_cls_pool_bwd(ptr(x_rows), x_rows.stride(0), ptr(cls_tok), ptr(hidden), int(n > 0 and hidden.dtype == torch.float32), hidden.stride(0) if n else 0, n,
                                int(normalize), ptr(p), ptr(dzbar), ptr(u), ptr(lens_i32), ptr(ds), ptr(pp), ptr(du), ptr(dck), ptr(dalpha), B, T, NQ, R, D,
                                int(nsplit), float(drop_p), int(seed) & 0xFFFFFFFF, stream()), "sc_cls_pool_bwd")
    if return_ws:          # (ds, pp): score gradients and post-dropout probabilities [B,R,NQ+T], what sc_cls_pool_dz needs
        return du, dck, dalpha, ds, pp
    return du, dck, dalpha


def layernorm_bwd(x, dy, gamma, dgamma=None, dbeta=None, eps=1e-5, dx=None, accumulate_dx=False):
    """fp32 [rows, D].  Returns dx; dgamma/dbeta are accumulated in place when given."""
    _f32c(x, dy, gamma, dgamma, dbeta)
    rows, D = x.shape
    if dx is None:
        assert not accumulate_dx
        dx = torch.empty_like(x)
    stats = torch.empty(rows, 2, device=x.device, dtype=torch.float32)
    check(lib().sc_layernorm_bwd(ptr(x), ptr(dy), ptr(gamma), ptr(dx), ptr(dgamma), ptr(dbeta), ptr(stats), rows, D, eps, int(accumulate_dx),
                                 stream()), "sc_layernorm_bwd")
    return dx


def gelu_f32(z):
    _f32c(z)
    y = torch.empty_like(z)
    check(lib().sc_gelu_f32(ptr(z), ptr(y), z.numel(), 0, stream()), "sc_gelu_f32")
    return y


def gelu_bwd_(z, dh):
    """dh *= gelu'(z) in place."""
    _f32c(z, dh)
    check(lib().sc_gelu_f32(ptr(z), ptr(dh), z.numel(), 1, stream()), "sc_gelu_f32")
    return dh


def colsum(x, out=None, accumulate=False):
    _need_cuda(x)
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    rows, cols = x.shape
    if out is None:
        out = torch.empty(cols, device=x.device, dtype=torch.float32)
        accumulate = False
    check(lib().sc_colsum(ptr(x), x.stride(0), rows, cols, ptr(out), int(accumulate), stream()), "sc_colsum")
    return out


def l2norm_bwd(x, dy):
    _f32c(x, dy)
    dx = torch.empty_like(x)
    check(lib().sc_l2norm_bwd(ptr(x), ptr(dy), ptr(dx), x.shape[0], x.shape[1], stream()), "sc_l2norm_bwd")
    return dx


def dropout_f32(x, drop_p, seed, out=None):
    _f32c(x)
    out = torch.empty_like(x) if out is None else out
    check(lib().sc_dropout_f32(ptr(x), ptr(out), x.numel(), float(drop_p), int(seed) & 0xFFFFFFFF, stream()), "sc_dropout_f32")
    return out


def add_rows(a, b, alpha=1.0, out=None):
    """out = alpha * a + b[r % b.shape[0]]  (fp32 2-D)."""
    _f32c(a, b)
    out = torch.empty_like(a) if out is None else out
    check(lib().sc_add_rows_f32(ptr(a), ptr(b), ptr(out), a.shape[0], a.shape[1], b.shape[0], float(alpha), stream()), "sc_add_rows_f32")
    return out


def mix_softmax_bwd(w, dalpha_b, dw):
    _f32c(w, dalpha_b, dw)
    check(lib().sc_mix_softmax_bwd(ptr(w), ptr(dalpha_b), dalpha_b.shape[0], dalpha_b.shape[1], ptr(dw), stream()), "sc_mix_softmax_bwd")
    return dw


def infonce_fwd_bwd(feat_a, feat_b, ids=None, inv_temperature=1.0 / 0.07, margin=0.0, dcl=False, a2b=True, b2a=True):
    """Loss and its gradients in one go: returns (out3, dfeat_a [Bg,E], dinv scalar tensor) -- d loss/d feat_a and d loss/d inv_temperature."""
    _f32c(feat_a, feat_b)
    Bg, E = feat_a.shape
    dev = feat_a.device
    if ids is not None:
        ids = ids.contiguous()
    ws = torch.empty(lib().sc_infonce_workspace_bytes(Bg), device=dev, dtype=torch.uint8)
    out = torch.empty(3, device=dev, dtype=torch.float32)
    check(lib().sc_infonce_fwd(ptr(feat_a), ptr(feat_b), ptr(ids), ptr(ws), ptr(out), Bg, E, inv_temperature, margin, int(dcl), int(a2b),
                               int(b2a), stream()), "sc_infonce_fwd")
    ws2 = torch.empty(lib().sc_infonce_bwd_workspace_bytes(Bg), device=dev, dtype=torch.uint8)
    G = torch.empty(Bg, Bg, device=dev, dtype=torch.float32)
    dinv = torch.empty(1, device=dev, dtype=torch.float32)
    check(lib().sc_infonce_bwd(ptr(feat_a), ptr(feat_b), ptr(ids), ptr(ws), ptr(ws2), ptr(G), ptr(dinv), Bg, E, inv_temperature, margin, int(dcl),
                               int(a2b), int(b2a), stream()), "sc_infonce_bwd")
    da = sgemm(G, feat_b, alpha=inv_temperature)
    return out, da, dinv


def grad_norm(flat_grad, max_norm=0.0):
    """-> f32[2] device tensor: (total L2 norm, clip coefficient)."""
    _f32c(flat_grad)
    ws = torch.empty(lib().sc_grad_norm_workspace_bytes(), device=flat_grad.device, dtype=torch.uint8)
    out = torch.empty(2, device=flat_grad.device, dtype=torch.float32)
    check(lib().sc_grad_norm(ptr(flat_grad), flat_grad.numel(), float(max_norm), ptr(ws), ptr(out), stream()), "sc_grad_norm")
    return out


def adam_step(p, g, m, v, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clip_coef=None):
    _f32c(p, g, m, v)
    check(lib().sc_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), ptr(clip_coef[1:] if clip_coef is not None else None), lr, betas[0], betas[1],
                             eps, weight_decay, int(step), stream()), "sc_adam_step")


def retrieval_ranks(score, own_ids, cand_ids):
    """score f32 [n, m] (device); own_ids i64 [n]; cand_ids i64 [m] -> int32 [n]: candidates ranked ahead of the row's best positive."""
    _need_cuda(score)
    assert score.dtype == torch.float32 and score.dim() == 2 and score.stride(1) == 1
    n, m = score.shape
    own = own_ids.to(score.device, torch.int64).contiguous()
    cand = cand_ids.to(score.device, torch.int64).contiguous()
    assert own.shape == (n,) and cand.shape == (m,)
    rank = torch.empty(n, device=score.device, dtype=torch.int32)
    check(lib().sc_retrieval_ranks(ptr(score), score.stride(0), ptr(own), ptr(cand), ptr(rank), n, m, stream()), "sc_retrieval_ranks")
    return rank


# ---- cascaded tail, training (train_cascaded.hip) ----
def attn_small_bwd(qkv16, dout, B, L, H, causal=True):
    """qkv bf16 [B*L, 3*H*64] (saved by the forward), dout f32 [B*L, H*64] -> dqkv f32 [B*L, 3*H*64]."""
    _need_cuda(qkv16, dout)
    W = H * 64
    assert qkv16.dtype == bf16 and qkv16.shape == (B * L, 3 * W) and qkv16.is_contiguous()
    _f32c(dout)
    dqkv = torch.empty(B * L, 3 * W, device=dout.device, dtype=torch.float32)
    check(lib().sc_attn_small_bwd(ptr(qkv16), ptr(dout), ptr(dqkv), B, L, H, 64, int(causal), stream()), "sc_attn_small_bwd")
    return dqkv


def split_hilo(a, nblk=2):
    """a f32 [M,K] (row stride allowed) -> bf16 [M, nblk*K] = (hi | lo) or (hi | lo | hi), hi + lo = a to ~16 bits."""
    _need_cuda(a)
    assert a.dtype == torch.float32 and a.dim() == 2 and a.stride(1) == 1
    M, K = a.shape
    out = torch.empty(M, nblk * K, device=a.device, dtype=bf16)
    check(lib().sc_split_hilo_bf16(ptr(a), a.stride(0), ptr(out), M, K, nblk, stream()), "sc_split_hilo_bf16")
    return out


def quickgelu_f32(z, out_bf16=False):
    _f32c(z)
    y = torch.empty(z.shape, device=z.device, dtype=bf16 if out_bf16 else torch.float32)
    check(lib().sc_quickgelu_f32(ptr(z), ptr(y), z.numel(), 0, int(out_bf16), stream()), "sc_quickgelu_f32")
    return y


def quickgelu_bwd_(z, dh):
    """dh *= quickgelu'(z) in place."""
    _f32c(z, dh)
    check(lib().sc_quickgelu_f32(ptr(z), ptr(dh), z.numel(), 1, 0, stream()), "sc_quickgelu_f32")
    return dh


def vq_st_bwd_(cos, dprob, temp, mask_ids=(0, 2, 3)):
    """dprob f32 [R,V] becomes d loss / d cos in place; returns rowdot f32 [R] = sum_v dcos cos."""
    import ctypes
    _f32c(cos, dprob)
    R, V = cos.shape
    rowdot = torch.empty(R, device=cos.device, dtype=torch.float32)
    ids = (ctypes.c_int32 * max(1, len(mask_ids)))(*[int(i) for i in mask_ids])
    check(lib().sc_vq_st_bwd(ptr(cos), ptr(dprob), ptr(rowdot), R, V, float(temp), ctypes.cast(ids, ctypes.c_void_p), len(mask_ids), stream()),
          "sc_vq_st_bwd")
    return rowdot


def cosine_bwd_finish(a, G, rowdot, eps=1e-8):
    _f32c(a, G, rowdot)
    da = torch.empty_like(a)
    check(lib().sc_cosine_bwd_finish(ptr(a), ptr(G), ptr(rowdot), ptr(da), a.shape[0], a.shape[1], eps, stream()), "sc_cosine_bwd_finish")
    return da


def kw_bn_train_fwd(x, gamma, beta, running_mean, running_var, momentum, eps):
    """x f32 [B,K,E]; gamma/beta/running_* f32 [E*K] in the reference's (e, k) order -> (y, mean [K*E], rstd [K*E]); running stats updated in place."""
    _f32c(x, gamma, beta)
    B, K, E = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(K * E, device=x.device, dtype=torch.float32)
    rstd = torch.empty(K * E, device=x.device, dtype=torch.float32)
    if running_mean is not None:
        _f32c(running_mean, running_var)
    check(lib().sc_kw_bn_train_fwd(ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), ptr(running_mean), ptr(running_var), B, K, E,
                                   float(momentum), float(eps), stream()), "sc_kw_bn_train_fwd")
    return y, mean, rstd


def kw_bn_bwd(x, dy, gamma, mean, rstd, want_param_grads=True):
    _f32c(x, dy, gamma, mean, rstd)
    B, K, E = x.shape
    dx = torch.empty_like(x)
    dg = torch.empty(K * E, device=x.device, dtype=torch.float32) if want_param_grads else None
    db = torch.empty(K * E, device=x.device, dtype=torch.float32) if want_param_grads else None
    check(lib().sc_kw_bn_bwd(ptr(x), ptr(dy), ptr(gamma), ptr(mean), ptr(rstd), ptr(dx), ptr(dg), ptr(db), B, K, E, stream()), "sc_kw_bn_bwd")
    return dx, dg, db


CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def image_normalize_u8(u8, mean=CLIP_MEAN, std=CLIP_STD):
    """uint8 [B,H,W,3] (already resized / centre-cropped on the host) -> f32 [B,3,H,W] = (x/255 - mean) / std  (ToTensor + Normalize)."""
    import ctypes
    _need_cuda(u8)
    assert u8.dtype == torch.uint8 and u8.dim() == 4 and u8.shape[-1] == 3 and u8.is_contiguous()
    B, H, W, _ = u8.shape
    out = torch.empty(B, 3, H, W, device=u8.device, dtype=torch.float32)
    m = (ctypes.c_float * 3)(*[float(x) for x in mean])
    sd = (ctypes.c_float * 3)(*[float(x) for x in std])
    check(lib().sc_image_normalize_u8(ptr(u8), ptr(out), B, H, W, ctypes.cast(m, ctypes.c_void_p), ctypes.cast(sd, ctypes.c_void_p), stream()),
          "sc_image_normalize_u8")
    return out


# ---- fine-tuning HuBERT layers (train_hubert.hip) ----
def transpose_bf16(src, ld_in, stride_in, rows, cols, batch, rows_padded=None, out=None, ld_out=None, stride_out=None):
    """batch of [rows, cols] bf16 matrices (row r of matrix z at src + z*stride_in + r*ld_in) -> out bf16 [batch, cols, rows_padded] (zero padded);
    ld_out / stride_out place matrix z at out + z*stride_out with rows of stride ld_out instead (e.g. side by side in one wide matrix)."""
    _need_cuda(src)
    rp = rows if rows_padded is None else rows_padded
    if out is None:
        out = torch.empty(batch, cols, rp, device=src.device, dtype=bf16)
    check(lib().sc_transpose_bf16(ptr(src), ld_in, stride_in, ptr(out), rp if ld_out is None else ld_out, cols * rp if stride_out is None else stride_out,
                                  rows, cols, rp, batch, stream()), "sc_transpose_bf16")
    return out


def attn_softmax_bwd(S, dP, dO_head, ld_do, O_head, ld_o, rows_per_batch, klens_i32, L, scale, drop=None):
    """S, dP f32 [Z, Lp, Lp]; dO_head / O_head: views at the head's first column, rows of stride ld_*; -> (P, dS) bf16 [Z, Lp, Lp].
    drop = (p, seed, H, h): the forward ran attention_dropout with (p, seed); this is head h of H."""
    _need_cuda(S, dP)
    Z, Lp, _ = S.shape
    P = torch.empty(Z, Lp, Lp, device=S.device, dtype=bf16)
    dS = torch.empty_like(P)
    if drop is not None and drop[0] > 0:
        check(lib().sc_attn_softmax_bwd_dropout(ptr(S), ptr(dP), Lp, Lp * Lp, ptr(dO_head), ld_do, ptr(O_head), ld_o, rows_per_batch, ptr(klens_i32), ptr(P),
                                                ptr(dS), L, Lp, Z, float(scale), float(drop[0]), int(drop[1]) & 0xffffffff, int(drop[2]), int(drop[3]),
                                                stream()), "sc_attn_softmax_bwd_dropout")
        return P, dS
    check(lib().sc_attn_softmax_bwd(ptr(S), ptr(dP), Lp, Lp * Lp, ptr(dO_head), ld_do, ptr(O_head), ld_o, rows_per_batch, ptr(klens_i32), ptr(P), ptr(dS),
                                    L, Lp, Z, float(scale), stream()), "sc_attn_softmax_bwd")
    return P, dS


def gelu_bwd_bf16(u, dh):
    _need_cuda(u, dh)
    assert u.dtype == bf16 and dh.dtype == bf16 and u.is_contiguous() and dh.is_contiguous() and u.shape == dh.shape
    du = torch.empty_like(u)
    check(lib().sc_gelu_bwd_bf16(ptr(u), ptr(dh), ptr(du), u.numel(), stream()), "sc_gelu_bwd_bf16")
    return du


def layernorm_bwd_bf16(x, dy, gamma, eps=1e-5, want_param_grads=True):
    """x, dy bf16 [rows, D] -> (dx bf16, dgamma f32 [D] | None, dbeta f32 [D] | None)."""
    _need_cuda(x, dy, gamma)
    rows, D = x.shape
    assert x.dtype == bf16 and dy.dtype == bf16 and x.is_contiguous() and dy.is_contiguous()
    dx = torch.empty_like(x)
    npart = int(lib().sc_layernorm_bwd_bf16_partials(rows))
    part = torch.empty(npart, 2 * D, device=x.device, dtype=torch.float32)
    check(lib().sc_layernorm_bwd_bf16(ptr(x), ptr(dy), ptr(gamma), ptr(dx), ptr(part), rows, D, eps, stream()), "sc_layernorm_bwd_bf16")
    if not want_param_grads:
        return dx, None, None
    sums = colsum(part)
    return dx, sums[:D].contiguous(), sums[D:].contiguous()


def colsum_bf16(x, out=None, accumulate=False):
    _need_cuda(x)
    assert x.dtype == bf16 and x.dim() == 2 and x.stride(1) == 1
    rows, cols = x.shape
    ws = torch.empty(int(lib().sc_colsum_bf16_workspace_bytes(rows, cols)), device=x.device, dtype=torch.uint8)
    if out is None:
        out = torch.empty(cols, device=x.device, dtype=torch.float32)
        accumulate = False
    check(lib().sc_colsum_bf16(ptr(x), x.stride(0), rows, cols, ptr(ws), ptr(out), int(accumulate), stream()), "sc_colsum_bf16")
    return out


def axpy_bf16(y, x, alpha):
    _need_cuda(y, x)
    assert y.dtype == bf16 and x.dtype == bf16 and y.is_contiguous() and x.is_contiguous() and y.numel() == x.numel()
    check(lib().sc_axpy_bf16(ptr(y), ptr(x), float(alpha), y.numel(), stream()), "sc_axpy_bf16")
    return y


def cls_pool_dz(pp, ds, dzbar, u, lens_i32, B, T, NQ, R, D):
    """-> dz bf16 [B*T, D]: gradient of the mixed frames from sc_cls_pool_bwd's workspaces."""
    _f32c(pp, ds, dzbar, u)
    dz = torch.empty(B * T, D, device=pp.device, dtype=bf16)
    check(lib().sc_cls_pool_dz(ptr(pp), ptr(ds), ptr(dzbar), ptr(u), ptr(lens_i32), ptr(dz), B, T, NQ, R, D, D, stream()), "sc_cls_pool_dz")
    return dz


# ---------------------------------------------------------------------------------------------- front-end backward (train_front.py)
def posconv_conv(x, valid_i32, wg, B, Tp, D, G, Kw):
    """The grouped positional conv alone: x bf16 [B*Tp, D] (frames >= valid[b] read as zero), wg bf16 [G, D/G, Kw*D/G] -> bf16 [B, G, Tp, D/G]."""
    _need_cuda(x, wg)
    cg = D // G
    conv = torch.empty(B * G * Tp * cg, device=x.device, dtype=bf16)
    rc = lib().sc_posconv_conv(ptr(x), ptr(valid_i32), ptr(wg), ptr(conv), B, Tp, D, G, Kw, stream())
    if rc == 1:
        xg = posconv_pack(x, valid_i32, B, Tp, D, G, Kw)
        gemm_batched(xg, cg, (Tp + Kw) * cg, wg, cg * Kw * cg, G, conv, cg, Tp * cg, None, Tp, cg, Kw * cg, B * G)
    else:
        check(rc, "sc_posconv_conv")
    return conv


def posconv_pack(x, valid_i32, B, Tp, D, G, Kw):
    """Sliding-window layout of the masked input: bf16 [B, G, Tp + Kw, D/G] (+64 slack elements), Kw/2 zero rows in front."""
    cg = D // G
    xg = torch.empty(B * G * (Tp + Kw) * cg + 64, device=x.device, dtype=bf16)
    check(lib().sc_posconv_pack(ptr(x), ptr(valid_i32), ptr(xg), B, Tp, D, G, Kw, stream()), "sc_posconv_pack")
    return xg


def posconv_finish_train(x, valid_i32, conv, bias, B, Tp, D, G):
    """-> (u, s) bf16 [B*Tp, D]: u = conv + bias regrouped, s = mask(x) + gelu(u)."""
    u = torch.empty(B * Tp, D, device=x.device, dtype=bf16)
    s = torch.empty_like(u)
    check(lib().sc_posconv_finish_train(ptr(x), ptr(valid_i32), ptr(conv), ptr(bias), ptr(u), ptr(s), B, Tp, D, G, stream()), "sc_posconv_finish_train")
    return u, s


def posconv_dgrad_finish(convT, ds, valid_i32, B, Tp, D, G):
    dx = torch.empty(B * Tp, D, device=ds.device, dtype=bf16)
    assert ds.dtype == bf16 and ds.is_contiguous()
    check(lib().sc_posconv_dgrad_finish(ptr(convT), ptr(ds), ptr(valid_i32), ptr(dx), B, Tp, D, G, stream()), "sc_posconv_dgrad_finish")
    return dx


def reverse_rows_bf16(x, B, T, D):
    assert x.dtype == bf16 and x.is_contiguous() and x.numel() == B * T * D
    out = torch.empty_like(x)
    check(lib().sc_reverse_rows_bf16(ptr(x), ptr(out), B, T, D, stream()), "sc_reverse_rows_bf16")
    return out


def conv0_bwd(wav, w, gamma, beta, dy, T0, P, eps=1e-5):
    """wav f32 [B, L]; w f32 [C, 10]; dy bf16 [B*P (+ slack), C] -> (dw f32 [C, 10], dgamma f32 [C], dbeta f32 [C])."""
    _need_cuda(wav, dy)
    B, L = wav.shape
    C = w.shape[0]
    assert wav.dtype == torch.float32 and wav.is_contiguous() and w.dtype == torch.float32 and w.is_contiguous() and dy.dtype == bf16 and dy.is_contiguous()
    part = torch.empty(B, C * 12, device=wav.device, dtype=torch.float32)
    check(lib().sc_conv0_bwd(ptr(wav), L, ptr(w), ptr(gamma), ptr(beta), ptr(dy), ptr(part), B, C, T0, P, eps, stream()), "sc_conv0_bwd")
    tot = colsum(part).view(C, 12)
    return tot[:, :10].contiguous(), tot[:, 10].contiguous(), tot[:, 11].contiguous()


def conv0_wgrad(wav, du, C, T0, P):
    """wav f32 [B, L]; du bf16 [B*P (+ slack), C] = gradient of conv layer 0's (pre-norm) output -> (dw f32 [C, 10], dbias f32 [C])."""
    _need_cuda(wav, du)
    B, L = wav.shape
    assert wav.dtype == torch.float32 and wav.is_contiguous() and du.dtype == bf16 and du.is_contiguous()
    part = torch.empty(B, C * 12, device=wav.device, dtype=torch.float32)
    check(lib().sc_conv0_wgrad(ptr(wav), L, ptr(du), ptr(part), B, C, T0, P, stream()), "sc_conv0_wgrad")
    tot = colsum(part).view(C, 12)
    return tot[:, :10].contiguous(), tot[:, 10].contiguous()


def gemm_batched2(a, lda, stride_a, stride_a2, w, ldw, stride_w, stride_w2, out, ldc, stride_c, stride_c2, M, N, K, outer, inner):
    """outer x inner products out_z[M,N] = a_z[M,K] w_z[N,K]^T, z = (zo, zi): operand at base + zo*stride + zi*stride2 (elements); out dtype f32 => fp32."""
    _need_cuda(a, w, out)
    flags = GEMM_OUT_F32 if out.dtype == torch.float32 else 0
    check(lib().sc_gemm_bf16_batched2(ptr(a), lda, stride_a, stride_a2, ptr(w), ldw, stride_w, stride_w2, ptr(out), ldc, stride_c, stride_c2, M, N, K,
                                      outer, inner, flags, stream()), "sc_gemm_bf16_batched2")
    return out


def attn_softmax_bwd_heads(S, dP, dO, O, rows_per_batch, klens_i32, L, scale, B, H, drop=None):
    """All heads at once: S, dP f32 [B*H, Lp, Lp]; dO / O bf16 [rows, H*64]; -> (P, dS) bf16 [B*H, Lp, Lp].  drop = (p, seed) of the forward."""
    _need_cuda(S, dP)
    Z, Lp, _ = S.shape
    assert Z == B * H and dO.stride(1) == 1 and O.stride(1) == 1
    P = torch.empty(Z, Lp, Lp, device=S.device, dtype=bf16)
    dS = torch.empty_like(P)
    p_, seed = (float(drop[0]), int(drop[1]) & 0xffffffff) if drop is not None else (0.0, 0)
    check(lib().sc_attn_softmax_bwd_heads(ptr(S), ptr(dP), Lp, Lp * Lp, ptr(dO), dO.stride(0), ptr(O), O.stride(0), rows_per_batch, ptr(klens_i32), ptr(P),
                                          ptr(dS), L, Lp, B, H, float(scale), p_, seed, stream()), "sc_attn_softmax_bwd_heads")
    return P, dS


def attn_bwd_probs(qkv, datt, att, klens_i32, B, T, H, drop=None):
    """P (dropped if the forward dropped) and dS bf16 [B*H, Lp, Lp] straight from the packed rows: qkv bf16 [>= B*T, 3*H*64], datt / att bf16 [B*T, H*64]."""
    _need_cuda(qkv, datt, att)
    d = H * 64
    Lp = -(-T // 64) * 64
    P = torch.empty(B * H, Lp, Lp, device=qkv.device, dtype=bf16)
    dS = torch.empty_like(P)
    p_, seed = (float(drop[0]), int(drop[1]) & 0xffffffff) if drop is not None else (0.0, 0)
    check(lib().sc_attn_bwd_probs(qkv.data_ptr(), qkv.data_ptr() + d * 2, qkv.data_ptr() + 2 * d * 2, 3 * d, ptr(datt), ptr(att), d, ptr(klens_i32), ptr(P),
                                  ptr(dS), B, H, T, Lp, 0.125, p_, seed, stream()), "sc_attn_bwd_probs")
    return P, dS

"""ctypes loader for libspeechclip_hip.so (the C-ABI product library).

There is deliberately NO fallback: if the shared object is missing or a call fails the product
raises.  PyTorch is used only for device memory and streams (tensor.data_ptr(), current stream).
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_uint32, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# SPEECHCLIP_HIP_LIB: developer override (A/B timing of two builds in one process launch each); the default is the in-tree build.
LIB_PATH = os.environ.get("SPEECHCLIP_HIP_LIB") or os.path.join(_HERE, "libspeechclip_hip.so")

ACT_NONE, ACT_GELU, ACT_QUICKGELU = 0, 1, 2
GEMM_OUT_F32 = 0x10
GEMM_F16 = 0x20          # SC_GEMM_F16: IEEE-half operands (and 16-bit outputs)
ATTN_CAUSAL, ATTN_F16 = 0x1, 0x2

_lib = None


class SpeechClipHipError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SpeechClipHipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C speechclip_amd/csrc).  There is no CPU/PyTorch fallback for the hot path.")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.sc_last_error.restype = c_char_p
        _declare(_lib)
        # SC_GEMM_KERNEL_MODE: developer override of the bf16 GEMM kernel choice (sc_debug_set_gemm_mode: 0 = gemm256_kernel only, 16 = gemm8p wherever
        # the shape allows, ...), for in-step A/B timing; unset = the dispatcher's rule
        mode = os.environ.get("SC_GEMM_KERNEL_MODE")
        if mode:
            try:
                _lib.sc_debug_set_gemm_mode(int(mode))
            except ValueError:
                import warnings
                warnings.warn(f"SC_GEMM_KERNEL_MODE={mode!r} is not an integer: ignored")
    return _lib


def _declare(L):
    P, I, L64, F, U32 = c_void_p, c_int, c_int64, c_float, c_uint32
    sigs = {
        "sc_abi_version": ([], c_int),
        "sc_set_gemm_workspace": ([P, L64], c_int),
        "sc_gemm_bf16": ([P, L64, P, L64, P, L64, P, P, L64, L64, I, I, I, P], c_int),
        "sc_gemm_bf16_batched": ([P, L64, L64, P, L64, L64, I, P, L64, L64, P, L64, I, I, I, I, P], c_int),
        "sc_layernorm": ([P, L64, P, P, P, L64, L64, I, F, I, P], c_int),
        "sc_weighted_sum_fwd": ([P, L64, P, P, I, L64, I, I, F, P], c_int),
        "sc_transpose_bf16": ([P, L64, L64, P, L64, L64, I, I, I, I, P], c_int),
        "sc_attn_softmax_bwd": ([P, P, L64, L64, P, L64, P, L64, L64, P, P, P, I, I, I, F, P], c_int),
        "sc_attn_softmax_bwd_dropout": ([P, P, L64, L64, P, L64, P, L64, L64, P, P, P, I, I, I, F, F, U32, I, I, P], c_int),
        "sc_attn_softmax_bwd_heads": ([P, P, L64, L64, P, L64, P, L64, L64, P, P, P, I, I, I, I, F, F, U32, P], c_int),
        "sc_attn_bwd_probs": ([P, P, P, L64, P, P, L64, P, P, P, I, I, I, I, F, F, U32, P], c_int),
        "sc_gemm_bf16_batched2": ([P, L64, L64, L64, P, L64, L64, L64, P, L64, L64, L64, L64, I, I, I, I, I, P], c_int),
        "sc_gelu_bwd_bf16": ([P, P, P, L64, P], c_int),
        "sc_layernorm_bwd_bf16_partials": ([L64], c_int64),
        "sc_layernorm_bwd_bf16": ([P, P, P, P, P, L64, I, F, P], c_int),
        "sc_colsum_bf16_workspace_bytes": ([L64, I], c_int64),
        "sc_colsum_bf16": ([P, L64, L64, I, P, P, I, P], c_int),
        "sc_axpy_bf16": ([P, P, F, L64, P], c_int),
        "sc_cls_pool_dz": ([P, P, P, P, P, P, I, I, I, I, I, L64, P], c_int),
        "sc_l2norm_fwd": ([P, L64, P, L64, I, I, P], c_int),
        "sc_splitk_reduce_f32": ([P, I, L64, I, P, P, L64, P, I, P], c_int),
        "sc_hidden_normalize": ([P, I, I, I, I, I, I, I, P, P], c_int),
        "sc_wave_layernorm": ([P, P, P, I, L64, F, P], c_int),
        "sc_attention_fwd": ([P, P, P, P, P, I, I, I, I, L64, L64, F, I, P], c_int),
        "sc_attention_fwd_dropout": ([P, P, P, P, P, I, I, I, I, L64, L64, F, I, F, U32, P], c_int),
        "sc_dropout_bf16": ([P, P, P, L64, F, U32, P], c_int),
        "sc_dropout_add_layernorm_bf16": ([P, P, P, P, P, L64, I, F, F, U32, P], c_int),
        "sc_cls_attention_fwd": ([P, P, L64, P, P, I, I, I, I, I, F, P], c_int),
        "sc_attention_rows_fwd": ([P, P, P, P, P, I, I, I, I, L64, L64, F, P], c_int),
        "sc_attention_probs_fwd": ([P, P, P, P, I, I, I, I, I, L64, F, P], c_int),
        "sc_posconv_finish_train": ([P, P, P, P, P, P, I, I, I, I, P], c_int),
        "sc_posconv_dgrad_finish": ([P, P, P, P, I, I, I, I, P], c_int),
        "sc_reverse_rows_bf16": ([P, P, I, I, I, P], c_int),
        "sc_conv0_bwd": ([P, L64, P, P, P, P, P, I, I, I, I, F, P], c_int),
        "sc_conv0_wgrad": ([P, L64, P, P, I, I, I, I, P], c_int),
        "sc_topk_rows_f32": ([P, L64, L64, I, I, P, P, P], c_int),
        "sc_cls_pool_fwd": ([P, L64, P, P, P, P, P, I, I, I, I, I, P], c_int),
        "sc_cls_pool_fwd_split": ([P, L64, P, P, P, P, P, I, I, I, I, I, I, P], c_int),
        "sc_conv0_stats_workspace_bytes": ([I], c_int64),
        "sc_conv0_gn_coef": ([P, L64, P, P, P, P, P, I, I, I, F, P], c_int),
        "sc_conv0_wfrag_workspace_bytes": ([I], c_int64),
        "sc_conv0_fwd": ([P, L64, L64, P, P, P, P, I, I, I, I, I, P, P], c_int),
        "sc_posconv_conv": ([P, P, P, P, I, I, I, I, I, P], c_int),
        "sc_posconv_pack": ([P, P, P, I, I, I, I, I, P], c_int),
        "sc_posconv_finish": ([P, P, P, P, P, P, P, I, I, I, I, I, F, P], c_int),
        "sc_crop_pad": ([P, L64, P, P, P, I, I, P], c_int),
        "sc_conv0_fwd_packed": ([P, L64, L64, P, P, P, P, I, I, I, P, I, I, I, P, P], c_int),
        "sc_posconv_conv_packed": ([P, P, P, P, P, I, I, I, I, I, P], c_int),
        "sc_posconv_finish_packed": ([P, P, P, P, P, P, P, P, I, L64, I, I, I, F, P], c_int),
        "sc_attention_fwd_packed": ([P, P, P, P, P, P, I, I, I, L64, I, L64, L64, F, F, U32, I, P], c_int),
        "sc_unpack_rows": ([P, L64, P, P, L64, I, I, I, I, I, P], c_int),
        "sc_image_normalize_u8": ([P, P, I, I, I, P, P, P], c_int),
        "sc_vit_patchify": ([P, P, I, I, I, I, P], c_int),
        "sc_vit_embed": ([P, P, P, P, P, P, I, I, I, F, P], c_int),
        "sc_infonce_workspace_bytes": ([I], c_int64),
        "sc_infonce_fwd": ([P, P, P, P, P, I, I, F, F, I, I, I, P], c_int),
        "sc_kw_affine": ([P, P, P, P, L64, I, I, P], c_int),
        "sc_cosine_workspace_bytes": ([I, I], c_int64),
        "sc_cosine_scores": ([P, P, P, P, I, I, I, F, P], c_int),
        "sc_vq_workspace_bytes": ([I, I], c_int64),
        "sc_vq_fwd": ([P, P, P, P, P, I, I, I, P, I, P], c_int),
        "sc_gather_rows": ([P, P, P, I, I, P], c_int),
        "sc_retrieval_ranks": ([P, L64, P, P, P, I, I, P], c_int),
        "sc_sgemm": ([I, I, I, I, I, F, P, L64, P, L64, F, P, L64, P, P], c_int),
        "sc_sgemm_batched": ([I, I, I, I, I, F, P, L64, L64, P, L64, L64, F, P, L64, L64, P, L64, I, P], c_int),
        "sc_cls_pool_train_fwd": ([P, L64, P, P, P, P, P, P, I, I, I, I, I, F, U32, P], c_int),
        "sc_cls_pool_bwd": ([P, L64, P, P, I, L64, I, I, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, F, U32, P], c_int),
        "sc_layernorm_bwd": ([P, P, P, P, P, P, P, I, I, F, I, P], c_int),
        "sc_gelu_f32": ([P, P, L64, I, P], c_int),
        "sc_gemm_last_path": ([], c_int),
        "sc_debug_set_gemm_mode": ([I], None),          # developer / test switch: exported, not in include/speechclip_hip.h
        "sc_debug_poison_gemm_sched": ([], c_int),      # test hook: exported, not in the header
        "sc_debug_vendor_stream_slot": ([P], c_int),
        "sc_split_hilo_bf16": ([P, L64, P, L64, I, I, P], c_int),
        "sc_cosine_refine": ([P, P, P, I, I, I, F, F, P], c_int),
        "sc_attn_small_bwd": ([P, P, P, I, I, I, I, I, P], c_int),
        "sc_quickgelu_f32": ([P, P, L64, I, I, P], c_int),
        "sc_vq_st_bwd": ([P, P, P, I, I, F, P, I, P], c_int),
        "sc_cosine_bwd_finish": ([P, P, P, P, I, I, F, P], c_int),
        "sc_kw_bn_train_fwd": ([P, P, P, P, P, P, P, P, I, I, I, F, F, P], c_int),
        "sc_kw_bn_bwd": ([P, P, P, P, P, P, P, P, I, I, I, P], c_int),
        "sc_colsum": ([P, L64, I, I, P, I, P], c_int),
        "sc_l2norm_bwd": ([P, P, P, I, I, P], c_int),
        "sc_dropout_f32": ([P, P, L64, F, U32, P], c_int),
        "sc_mix_softmax_bwd": ([P, P, I, I, P, P], c_int),
        "sc_add_rows_f32": ([P, P, P, I, I, I, F, P], c_int),
        "sc_infonce_bwd_workspace_bytes": ([I], c_int64),
        "sc_infonce_bwd": ([P, P, P, P, P, P, P, I, I, F, F, I, I, I, P], c_int),
        "sc_grad_norm_workspace_bytes": ([], c_int64),
        "sc_grad_norm": ([P, L64, F, P, P, P], c_int),
        "sc_adam_step": ([P, P, P, P, L64, P, F, F, F, F, F, I, P], c_int),
    }
    for name, (args, res) in sigs.items():
        fn = getattr(L, name)   # AttributeError here = header/library mismatch: fail loudly
        fn.argtypes = args
        fn.restype = res


def check(rc, what):
    if rc != 0:
        raise SpeechClipHipError(f"{what} failed (rc={rc}): {lib().sc_last_error().decode()}")


def ptr(t):
    return 0 if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def exported_symbols():
    """Names declared in include/speechclip_hip.h (parsed), for the CPU-side ABI test."""
    import re
    hdr = os.path.join(os.path.dirname(_HERE), "include", "speechclip_hip.h")
    text = open(hdr).read()
    return sorted(set(re.findall(r"\b(sc_[a-z0-9_]+)\s*\(", text)))

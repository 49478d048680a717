#!/usr/bin/env python3
"""PROBES build only: the 64-query-rows-per-wave attention forms (SC_ATTN_QB=2 / 3; measured +0.2 / +0.9 ms per step, not shipped) against the
fp32 reference.  usage: SPEECHCLIP_HIP_LIB=speechclip_amd/libspeechclip_hip_probes.so SC_ATTN_QB=2 python tools/probes_attn_qb_check.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_kernels_gpu import _attn_ref, _g, BF
from speechclip_amd import ops
assert "probes" in os.environ.get("SPEECHCLIP_HIP_LIB", ""), "needs the PROBES library (make -C speechclip_amd/csrc PROBES=1)"
for B, T, H, lens in [(3, 500, 12, [500, 499, 37]), (2, 129, 2, [1, 64]), (2, 319, 16, [319, 65]), (1, 700, 3, [513]), (2, 256, 1, None)]:
    qkv = torch.randn(B * T, 3 * H * 64, generator=_g(T + H)).to("cuda", BF)
    klens = torch.tensor(lens, dtype=torch.int32, device="cuda") if lens is not None else None
    y = ops.attention(qkv, B, T, H, klens)
    torch.testing.assert_close(y.float(), _attn_ref(qkv, B, T, H, klens), atol=2e-2, rtol=2e-2)
print("QB_OK", os.environ.get("SC_ATTN_QB"))

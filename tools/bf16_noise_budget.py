#!/usr/bin/env python3
"""CPU experiment (fp32 oracle + hooks): which bf16 roundings of the HIP path cost how much CENTRED cosine of the final audio embedding?  Each group of
rounding sites is injected into the fp32 oracle on its own and the embedding is compared with the clean oracle's.  usage: python tools/bf16_noise_budget.py [base|large] [B] [seconds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn as nn, torch.nn.functional as F
from helpers import centred_cos
from oracle.clip_ref import ClipRefConfig
from oracle.hubert_ref import HubertRefConfig, MultiheadAttentionRef, TransformerSentenceEncoderLayerRef
from oracle.speechclip_ref import SpeechClipRef, l2_normalize
kind = sys.argv[1] if len(sys.argv) > 1 else "large"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
secs = float(sys.argv[3]) if len(sys.argv) > 3 else 4.0
large = kind == "large"
torch.manual_seed(12)
hcfg = HubertRefConfig.large() if large else HubertRefConfig.base()
ref = SpeechClipRef(hcfg, ClipRefConfig.tiny(), parallel=True, branch_heads=8, normalize_hiddenstates=large).eval()
g = torch.Generator().manual_seed(21)
with torch.no_grad():
    ref.ws_weights.copy_(0.5 * torch.randn(ref.ws_weights.shape, generator=g))
    for m in ref.encoder.modules():
        if isinstance(m, (nn.LayerNorm, nn.GroupNorm)) and m.weight is not None:
            m.weight.add_(0.1 * torch.randn(m.weight.shape, generator=g)); m.bias.add_(0.1 * torch.randn(m.bias.shape, generator=g))
L = int(secs * 16000)
lens = [int(x) for x in torch.randint(L // 2, L + 1, (B,), generator=g)]; lens[0] = L
wav = torch.zeros(B, L)
for i, n in enumerate(lens): wav[i, :n] = 0.1 * torch.randn(n, generator=g) + 0.01
wl = torch.tensor(lens)
r16 = lambda t: t.to(torch.bfloat16).to(t.dtype)
def run():
    with torch.no_grad():
        feat, flen, _ = ref.forward_audio(wav, wl)
        return l2_normalize(ref.parallel_branch(feat, flen))
clean = run()
def with_hooks(make):
    hs = make()
    try: return run()
    finally:
        for h in hs: h.remove()
enc = ref.encoder
def conv_sites(conv_out=True, block_out=True):
    hs = []
    for blk in enc.feature_extractor.conv_layers:
        if conv_out: hs.append(blk[0].register_forward_hook(lambda m, a, o: r16(o)))
        if block_out: hs.append(blk.register_forward_hook(lambda m, a, o: r16(o)))
    return hs
def layer_inputs():      # LN output -> bf16 GEMM operand (q/k/v/fc1 inputs), attention output -> out_proj input, GELU output -> fc2 input
    hs = []
    for m in enc.modules():
        if isinstance(m, MultiheadAttentionRef):
            for p in (m.q_proj, m.k_proj, m.v_proj, m.out_proj): hs.append(p.register_forward_pre_hook(lambda mod, a: (r16(a[0]),)))
        if isinstance(m, TransformerSentenceEncoderLayerRef):
            hs.append(m.fc1.register_forward_pre_hook(lambda mod, a: (r16(a[0]),)))
            hs.append(m.fc2.register_forward_pre_hook(lambda mod, a: (r16(a[0]),)))
    hs.append(enc.post_extract_proj.register_forward_pre_hook(lambda mod, a: (r16(a[0]),)))
    return hs
def qkv_outputs():
    hs = []
    for m in enc.modules():
        if isinstance(m, MultiheadAttentionRef):
            for p in (m.q_proj, m.k_proj, m.v_proj): hs.append(p.register_forward_hook(lambda mod, a, o: r16(o)))
    return hs
def layer_outputs():     # post-LN base: the bf16 residual stream (each layer's output rounded)
    return [m.register_forward_hook(lambda mod, a, o: (r16(o[0]), o[1])) for m in enc.modules() if isinstance(m, TransformerSentenceEncoderLayerRef)]
def weights():
    saved = [(p, p.data.clone()) for m in enc.modules() if isinstance(m, (nn.Linear, nn.Conv1d)) for p in [m.weight]]
    for p, _ in saved: p.data = r16(p.data)
    class H:
        def remove(self):
            for p, d in saved: p.data = d
    return [H()]
groups = [("conv stack: conv outputs + block outputs -> bf16", lambda: conv_sites(True, True)),
          ("conv stack: block outputs only (conv outputs kept fp32)", lambda: conv_sites(False, True)),
          ("transformer GEMM operands (LN out, attention out, GELU out) -> bf16", layer_inputs),
          ("q / k / v -> bf16", qkv_outputs),
          ("layer outputs -> bf16 (the post-LN model's bf16 residual stream)", layer_outputs),
          ("encoder weights -> bf16", weights),
          ("ALL of the above", lambda: conv_sites() + layer_inputs() + qkv_outputs() + (layer_outputs() if not large else []) + weights())]
print(f"{kind}: B={B}, {secs} s; different utterances' raw cosine {F.cosine_similarity(clean[:1], clean[1:2]).item():.6f}")
for name, mk in groups:
    out = with_hooks(mk)
    cc = centred_cos(out, clean)
    print(f"{name:78s} centred cos min {cc.min().item():.4f} mean {cc.mean().item():.4f}   1-mean = {1 - cc.mean().item():.5f}")

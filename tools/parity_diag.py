"""Where does the centred-cosine gap of the T = 499 white-noise batch come from?  (diagnostic for tests/test_headline_parity_gpu.py)"""
import os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from helpers import centred_cos
from test_headline_parity_gpu import _share, _nontrivial_, _spread
from oracle.clip_ref import ClipRefConfig
from oracle.hubert_ref import HubertRefConfig
from oracle.speechclip_ref import SpeechClipRef, l2_normalize

torch.set_num_threads(16)
model = bench.build_model(); _nontrivial_(model, 11)
ref = SpeechClipRef(HubertRefConfig.base(), ClipRefConfig.vit_b32(), parallel=True, branch_heads=8).eval()
_share(model, ref); model = model.cuda()
B, L = int(os.environ.get("DIAG_B", "256")), 160000
batch, lens = bench.make_batch(B, L, 0, "cuda")
idx = _spread(B, lens)
with torch.no_grad():
    feat_h, flen, hid_h = model.forward_audio(batch["wav"], batch["wav_len"], return_hidden_states=True)
    lf, _, _ = model(batch)
    sub = {k: v[idx] for k, v in batch.items()}
    feat_s, _, hid_s = model.forward_audio(sub["wav"], sub["wav_len"], return_hidden_states=True)
    lf_s, _, _ = model(sub)
    subc = {k: v.cpu() for k, v in sub.items()}
    feat_o, flen_o, hid_o = ref.forward_audio(subc["wav"], subc["wav_len"])
    par_o = l2_normalize(ref.parallel_branch(feat_o, flen_o))
    # HIP head on the oracle's frames
    par_h_on_o = l2_normalize(model.parallel_branch(feat_o.cuda().to(feat_h.dtype), flen_o.cuda()).float())
    # oracle head on the HIP frames
    par_o_on_h = l2_normalize(ref.parallel_branch(feat_h[idx].float().cpu(), flen_o))
a = lf["parallel_audio_feat"][idx].float().cpu(); a_s = lf_s["parallel_audio_feat"].float().cpu()
def rep(name, got, want):
    cc = centred_cos(got, want); raw = F.cosine_similarity(got, want, dim=-1)
    err = (got - want).norm(dim=-1); sig = (want - want.mean(0, keepdim=True)).norm(dim=-1)
    print(f"{name}: centred cos min {cc.min():.4f} mean {cc.mean():.4f}; raw cos min {raw.min():.6f}; |err| mean {err.mean():.2e}; centred |ref| mean {sig.mean():.2e}")
rep("HIP(B=256 rows) vs oracle", a, par_o)
rep("HIP(subset alone) vs oracle", a_s, par_o)
rep("HIP(B=256 rows) vs HIP(subset alone)", a, a_s)
rep("HIP head on ORACLE frames vs oracle", par_h_on_o.cpu(), par_o)
rep("ORACLE head on HIP frames vs oracle", par_o_on_h, par_o)
print("oracle raw cos between different utterances:", F.cosine_similarity(par_o[:1], par_o[1:2]).item())
for li in (0, 1, 3, 6, 9, 12):
    h, o = hid_h[li][idx].float().cpu(), hid_o[li]
    n = h.shape[0]
    raw = F.cosine_similarity(h.reshape(n, -1), o.reshape(n, -1), dim=-1)
    # time-mean of the frames, centred over utterances (what a pooling head sees)
    cc = centred_cos(h.mean(1), o.mean(1))
    rel = ((h - o).norm() / o.norm()).item()
    print(f"hidden[{li}]: raw cos min {raw.min():.6f}; rel err {rel:.2e}; centred cos of time-mean min {cc.min():.4f}")
hm, om = feat_h[idx].float().cpu(), feat_o
print("audio_feat: rel err", ((hm - om).norm() / om.norm()).item(), "centred cos of time-mean", centred_cos(hm.mean(1), om.mean(1)).min().item())

import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ["SC_FROZEN_DROPOUT"] = "0"
import torch, tempfile
from forward_fuzz import build
from speechclip_amd.base import OrderedNamespace
model, ref = build(True, False)
model.eval()
for lens in ([8000], [400, 8000, 3000]):
    B = len(lens)
    g = torch.Generator().manual_seed(B)
    wav = torch.zeros(B, max(lens))
    for i, l in enumerate(lens): wav[i, :l] = 0.3 * torch.randn(l, generator=g)
    with torch.no_grad():
        feat, flen = model.forward_audio(wav.cuda(), torch.tensor(lens).cuda())
        cw, names, _ = model.cascaded_branch.getAttentionMap(feat, flen)
    assert len(cw) == B and all(w.shape[-1] == int(flen[i]) + 8 for i, w in enumerate(cw)) and all(torch.isfinite(w).all() for w in cw)
    print("getAttentionMap", lens, [tuple(w.shape) for w in cw], names[0][0][:3])
    model.config.trainer.default_root_dir = tempfile.mkdtemp()
    model.config.data = OrderedNamespace({"dev_batch_size": 16})
    batch = {"wav": wav.cuda(), "wav_len": torch.tensor(lens).cuda(), "image": torch.randn(B, 3, 64, 64, generator=g).cuda(), "id": torch.arange(B).cuda(),
             "text": torch.randint(4, 100, (B, 1, 77), generator=g).cuda()}
    with torch.no_grad():
        out = model.validation_step_end(model.validation_step(batch, 0))
        res = model.validation_epoch_end([out])
    print("validation_epoch_end", lens, model.last_kw_hit_rate[0].tolist(), res[2])
print("analysis fuzz OK")

"""One-process-per-GPU exchange step of the contrastive loss: RCCL all-gather of the per-rank embeddings over xGMI.

Replaces nn.DataParallel's gather-to-device-0 of the feature dicts (Lightning `strategy: dp`,
config/speechCLIP/model_base/spchclp_p.yaml:127; consumed by training_step_end, avssl/model/kwClip.py:147-191).
All float features and the int64 ids travel in ONE packed fp32 buffer per rank => a single collective of
B_local x (sum(E_k) + 2) floats (1.0 MiB sent / 8.4 MiB received per rank at global batch 2048, E = 512); at this
size the collective is latency-bound, so fewer, fused calls beat bandwidth tuning.  Rank-major concatenation equals
DataParallel's dim-0 gather order, so the global-batch loss is identical to the reference's.
"""
from typing import Dict

import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def gather_loss_feats(feats: Dict[str, torch.Tensor], force: bool = False) -> Dict[str, torch.Tensor]:
    """{id [B] i64, image_feat [B,E], parallel_audio_feat / cascaded_audio_feat [B,E]} -> same keys, global batch.
    `force`: run the packed collective even at world size 1 (single-GPU check of the RCCL path)."""
    rank, ws = world()
    if ws == 1 and not (force and dist.is_available() and dist.is_initialized()):
        return feats
    keys = [k for k in sorted(feats) if k != "id" and torch.is_tensor(feats[k])]
    ids = feats["id"].to(torch.int64).contiguous()
    B = ids.shape[0]
    dev = feats[keys[0]].device
    widths = [feats[k].shape[1] for k in keys]
    packed = torch.empty(B, sum(widths) + 2, device=dev, dtype=torch.float32)
    off = 0
    for k, w in zip(keys, widths):
        packed[:, off:off + w] = feats[k].float()
        off += w
    packed[:, off:off + 2] = ids.to(dev).view(torch.int32).view(B, 2).view(torch.float32)   # bit-cast, no value conversion
    out = torch.empty(ws * B, packed.shape[1], device=dev, dtype=torch.float32)
    dist.all_gather_into_tensor(out, packed)
    res, off = {}, 0
    for k, w in zip(keys, widths):
        res[k] = out[:, off:off + w].contiguous()
        off += w
    res["id"] = out[:, off:off + 2].contiguous().view(torch.int32).view(-1, 2).view(torch.int64).view(-1)
    return res

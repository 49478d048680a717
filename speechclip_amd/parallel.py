"""One-process-per-GPU exchange step of the contrastive loss: RCCL all-gather of the per-rank embeddings over xGMI.

Replaces nn.DataParallel's gather-to-device-0 of the feature dicts (Lightning `strategy: dp`,
config/speechCLIP/model_base/spchclp_p.yaml:127; consumed by training_step_end, avssl/model/kwClip.py:147-191).
All float features and the int64 ids travel in ONE packed fp32 buffer per rank => a single collective of
B_local x (sum(E_k) + 2) floats (1.0 MiB sent / 8.4 MiB received per rank at global batch 2048, E = 512); at this
size the collective is latency-bound, so fewer, fused calls beat bandwidth tuning.  Rank-major concatenation equals
DataParallel's dim-0 gather order, so the global-batch loss is identical to the reference's.
"""
from typing import Dict

import math
import torch
import torch.distributed as dist


# bench instrumentation: when a list, every exchange (pack + the ONE collective + unpack) appends its (start, end) HIP events -- the exchange
# measured INSIDE the step, on the stream it runs on
PROFILE_EXCHANGE = None


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def pack_feats(feats: Dict[str, torch.Tensor]):
    """-> (packed f32 [B, sum(E_k) + 2], keys, widths): every float feature side by side, the int64 ids bit-cast into the last two fp32 lanes."""
    keys = [k for k in sorted(feats) if k != "id" and torch.is_tensor(feats[k])]
    ids = feats["id"].to(torch.int64).contiguous()
    B = ids.shape[0]
    dev = feats[keys[0]].device
    widths = [feats[k].shape[1] for k in keys]
    packed = torch.empty(B, sum(widths) + 2, device=dev, dtype=torch.float32)
    off = 0
    for k, w in zip(keys, widths):
        packed[:, off:off + w] = feats[k].detach().float()
        off += w
    packed[:, off:off + 2] = ids.to(dev).view(torch.int32).view(B, 2).view(torch.float32)   # bit-cast, no value conversion
    return packed, keys, widths


def unpack_feats(out: torch.Tensor, keys, widths) -> Dict[str, torch.Tensor]:
    res, off = {}, 0
    for k, w in zip(keys, widths):
        res[k] = out[:, off:off + w].contiguous()
        off += w
    res["id"] = out[:, off:off + 2].contiguous().view(torch.int32).view(-1, 2).view(torch.int64).view(-1)
    return res


def all_gather_packed(packed: torch.Tensor) -> torch.Tensor:
    """THE exchange step: one all_gather_into_tensor (RCCL over xGMI) of the packed per-rank rows, rank-major."""
    rank, ws = world()
    out = torch.empty(ws * packed.shape[0], packed.shape[1], device=packed.device, dtype=packed.dtype)
    dist.all_gather_into_tensor(out, packed.contiguous())
    return out


def gather_loss_feats(feats: Dict[str, torch.Tensor], force: bool = False) -> Dict[str, torch.Tensor]:
    """{id [B] i64, image_feat [B,E], parallel_audio_feat / cascaded_audio_feat [B,E]} -> same keys, global batch.
    `force`: run the packed collective even at world size 1 (single-GPU check of the RCCL path)."""
    rank, ws = world()
    if ws == 1 and not (force and dist.is_available() and dist.is_initialized()):
        return feats
    if PROFILE_EXCHANGE is not None and feats["id"].is_cuda:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    packed, keys, widths = pack_feats(feats)
    out = unpack_feats(all_gather_packed(packed), keys, widths)
    if PROFILE_EXCHANGE is not None and feats["id"].is_cuda:
        e1.record()
        PROFILE_EXCHANGE.append((e0, e1))
    return out


ROW_KEYS_1D = ("row_score",)      # 1-D float entries of validation outputs that ARE per-row ([B] -> [B, 1]); anything else 1-D is passed through


def gather_rows_dict(d: dict, keys_1d=ROW_KEYS_1D) -> dict:
    """Validation outputs (`others`: id, audio_feat, image_feat, text_feat, keywords ...): every tensor with >= 2 dims whose dim 0 is the local
    batch is all-gathered rank-major (float tensors of any trailing shape in ONE packed collective with the ids; other integer tensors one
    collective each), non-tensors are passed through.  1-D float tensors are gathered only when named in `keys_1d` (a [1] scalar-like entry at
    B == 1 is not a row tensor: ADVICE r3).  The reference's DataParallel hands validation_epoch_end the outputs of ALL replicas
    (kwClip.py:193-275): recall is ranked against the full candidate pool, not a per-rank shard."""
    rank, ws = world()
    if ws == 1:
        return d
    B = d["id"].shape[0]
    dev = d["id"].device
    # all_gather_into_tensor needs the same row count on every rank.  A ragged last validation batch (the reference's DataParallel scatter hands the
    # first replicas one row more, kwClip.py:193-269; a Flickr8k validation epoch of 5 000 captions does not divide by 8 B) is padded to the largest
    # local batch and the pad rows are dropped after the gather: ONE small collective (the local row counts) and ONE device->host read per call
    # (it must precede the gather: ranks with different B would otherwise deadlock inside it).
    sizes_t = torch.empty(ws, device=dev, dtype=torch.int64)
    dist.all_gather_into_tensor(sizes_t, torch.tensor([B], device=dev, dtype=torch.int64))
    sizes = [int(v) for v in sizes_t.tolist()]
    bmax = max(sizes)
    ragged = any(v != bmax for v in sizes)

    def pad(v):
        if v.shape[0] == bmax:
            return v.contiguous()
        return torch.cat([v, v.new_zeros((bmax - v.shape[0],) + tuple(v.shape[1:]))], dim=0)

    keep = None
    if ragged:
        keep = torch.cat([torch.arange(r * bmax, r * bmax + n, device=dev) for r, n in enumerate(sizes)])      # valid rows, rank-major

    def trim(g):
        return g if keep is None else g.index_select(0, keep)

    flt = {k: v for k, v in d.items() if torch.is_tensor(v) and v.is_floating_point() and v.dim() >= 1 and v.shape[0] == B
           and (v.dim() >= 2 or k in keys_1d)}
    shapes = {k: v.shape[1:] for k, v in flt.items()}
    # (explicit width: reshape(0, -1) of an EMPTY local batch is ambiguous and would raise on that rank only, behind the row-count collective --
    #  the other ranks would then block in the packed gather)
    feats = {k: pad(v.reshape(B, int(math.prod(v.shape[1:])))) for k, v in flt.items()}
    feats["id"] = pad(d["id"])
    out = gather_loss_feats(feats)
    res = dict(d)
    for k in flt:
        res[k] = trim(out[k]).view(-1, *shapes[k]).to(flt[k].dtype)
    res["id"] = trim(out["id"])
    for k, v in d.items():
        if k != "id" and torch.is_tensor(v) and not v.is_floating_point() and v.dim() >= 2 and v.shape[0] == B:
            g = torch.empty(ws * bmax, *v.shape[1:], device=v.device, dtype=v.dtype)
            dist.all_gather_into_tensor(g, pad(v))
            res[k] = trim(g)
    return res

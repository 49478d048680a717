"""collate_general (avssl/data/collate_function.py:7-36): list of dataset rows -> batch dict.

Same contract as the reference: tensors named "wav" are zero-right-padded to the longest utterance and a "wav_len" LongTensor is
added; other tensors are stacked; plain numbers become a LongTensor.  `collate_to_device` is the MI355X side of the hand-over: ONE pinned
staging copy per key, and images that arrive as uint8 HWC crops are normalised on the device (sc_image_normalize_u8) instead of
crossing PCIe as fp32."""
from typing import Sequence

import torch


def collate_general(batch: Sequence[dict]) -> dict:
    keys = list(batch[0].keys())
    has_wav = "wav" in keys and isinstance(batch[0]["wav"], torch.Tensor)
    out = {}
    for k in keys:
        vals = [row[k] for row in batch]
        if isinstance(vals[0], torch.Tensor):
            if k == "wav":
                lmax = max(v.shape[0] for v in vals)
                buf = vals[0].new_zeros((len(vals), lmax) + tuple(vals[0].shape[1:]))
                for i, v in enumerate(vals):
                    buf[i, :v.shape[0]] = v
                out[k] = buf
            else:
                out[k] = torch.stack(vals, dim=0)
        else:
            out[k] = torch.LongTensor(vals)
    if has_wav:
        out["wav_len"] = torch.LongTensor([len(row["wav"]) for row in batch])
    return out


def collate_to_device(batch: dict, device, non_blocking: bool = True) -> dict:
    """Move a collated batch to the GPU.  `wav_len` stays on the host (the encoder's length arithmetic is host logic); a uint8
    [B,H,W,3] `image` (resized + centre-cropped on the host, not yet ToTensor/Normalize'd) is normalised on the device."""
    from .. import ops
    out = {}
    for k, v in batch.items():
        if not torch.is_tensor(v) or k == "wav_len":
            out[k] = v
            continue
        src = v.pin_memory() if (v.device.type == "cpu" and non_blocking) else v
        d = src.to(device, non_blocking=non_blocking)
        if k == "image" and d.dtype == torch.uint8:
            d = ops.image_normalize_u8(d.contiguous())
        out[k] = d
    return out

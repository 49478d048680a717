"""CLIP image pre-processing for `forward_image(list_of_paths)` (avssl/model/kwClip.py:504-519 -> ClipModel.prep_image, clip_official.py:151-164 ->
the `preprocess` that `clip.load` returns = clip/clip.py `_transform(n_px)` [3P openai/CLIP]):

    Resize(n_px, BICUBIC)  ->  CenterCrop(n_px)  ->  convert("RGB")  ->  ToTensor()  ->  Normalize(CLIP mean, CLIP std)

torchvision is not installed; its PIL code path IS `Image.resize(..., BICUBIC)` + `Image.crop`, so the geometry below reproduces it with PIL alone:
the SHORTER side becomes n_px and the longer one int(n_px * long / short) (torchvision.transforms.functional.resize, int size), the crop window starts at
int(round((H - n_px) / 2)), int(round((W - n_px) / 2)) (center_crop).  The host part ends at uint8 HWC; scaling by 1/255 and the normalisation run on the
device (sc_image_normalize_u8) when the model lives there, so a batch crosses PCIe as bytes."""
from typing import List, Sequence, Union

import numpy as np
import torch

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def clip_resize_center_crop_u8(img, n_px: int = 224) -> np.ndarray:
    """PIL image (any mode / size) -> uint8 [n_px, n_px, 3], the geometric half of clip's `_transform`."""
    from PIL import Image
    w, h = img.size
    if not (w == h == n_px):
        if w <= h:
            nw, nh = n_px, int(n_px * h / w)
        else:
            nw, nh = int(n_px * w / h), n_px
        if (nw, nh) != (w, h):
            img = img.resize((nw, nh), Image.BICUBIC)
        top, left = int(round((nh - n_px) / 2.0)), int(round((nw - n_px) / 2.0))
        img = img.crop((left, top, left + n_px, top + n_px))
    return np.asarray(img.convert("RGB"), dtype=np.uint8)


def load_images_u8(paths: Sequence[str], n_px: int = 224) -> torch.Tensor:
    """-> uint8 [B, n_px, n_px, 3] (host)."""
    from PIL import Image
    out = np.empty((len(paths), n_px, n_px, 3), dtype=np.uint8)
    for i, p in enumerate(paths):
        with Image.open(p) as im:
            out[i] = clip_resize_center_crop_u8(im, n_px)
    return torch.from_numpy(out)


def normalize_u8(u8: torch.Tensor, device: Union[str, torch.device] = "cpu") -> torch.Tensor:
    """uint8 [B, H, W, 3] -> f32 [B, 3, H, W] = (x / 255 - mean) / std on `device`: the HIP kernel on a GPU; plain tensor arithmetic on the host
    (DataLoader workers run this on CPUs -- data layer, avssl/data/base_dataset.py:101-108 -- not the hot path)."""
    device = torch.device(device)
    if device.type == "cuda":
        from .. import ops
        return ops.image_normalize_u8(u8.to(device, non_blocking=True).contiguous(), CLIP_MEAN, CLIP_STD)
    x = u8.to(torch.float32).div_(255.0).permute(0, 3, 1, 2)
    mean, std = torch.tensor(CLIP_MEAN).view(1, 3, 1, 1), torch.tensor(CLIP_STD).view(1, 3, 1, 1)
    return ((x - mean) / std).contiguous()


def clip_preprocess(n_px: int = 224):
    """The callable `clip.load` returns as `preprocess` (ClipModel.image_preprocess, clip_official.py:50): PIL image -> f32 [3, n_px, n_px]."""
    def _preprocess(img) -> torch.Tensor:
        return normalize_u8(torch.from_numpy(clip_resize_center_crop_u8(img, n_px).copy())[None])[0]
    return _preprocess

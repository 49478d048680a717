from .collate_function import collate_general, collate_to_device
from .image_transforms import clip_preprocess, load_images_u8, normalize_u8

__all__ = ["collate_general", "collate_to_device", "clip_preprocess", "load_images_u8", "normalize_u8"]

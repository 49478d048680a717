from .collate_function import collate_general, collate_to_device

__all__ = ["collate_general", "collate_to_device"]

"""Helpers on the hot-path boundary: key-padding mask and parameter freezing (avssl/util/data_utils.py:4-20,
avssl/util/model_utils.py:6-13) plus the general CLI flags (avssl/util/args.py:4-38)."""
import argparse

import torch
from torch import nn


def get_keypadding_mask(max_length: int, data_lens: torch.Tensor) -> torch.Tensor:
    """bool [B, max_length], True = padding (vectorised; the reference builds it with a Python loop on the CPU)."""
    lens = torch.as_tensor(data_lens).reshape(-1, 1)
    return torch.arange(max_length, device=lens.device)[None, :] >= lens


def freeze_model(m: nn.Module) -> None:
    for p in m.parameters():
        p.requires_grad = False


def unfreeze_model(m: nn.Module) -> None:
    for p in m.parameters():
        p.requires_grad = True


def add_general_arguments(parser: argparse.ArgumentParser) -> argparse.ArgumentParser:
    for flag, kw in (("--config", dict(type=str, default="")), ("--save_path", dict(type=str, default="")),
                     ("--train", dict(action="store_true")), ("--eval", dict(action="store_true")),
                     ("--test", dict(action="store_true")), ("--ckpt", dict(type=str, default="")),
                     ("--resume", dict(type=str, default="")), ("--njobs", dict(type=int, default=0)),
                     ("--gpus", dict(type=int, default=0)), ("--seed", dict(type=int, default=7122)),
                     ("--dataset_root", dict(type=str, default="")), ("--log_level", dict(type=str, default="info"))):
        parser.add_argument(flag, **kw)
    return parser

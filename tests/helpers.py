"""Shared test helper: the shipped configurations live in the package (speechclip_amd/util/shipped_configs.py)."""
from speechclip_amd.util.shipped_configs import make_config  # noqa: F401

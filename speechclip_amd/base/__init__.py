"""Configuration container of the plugin surface (checkpoints pickle `avssl.base.OrderedNamespace`; the alias package maps it here)."""
from . import ordered_namespace as _on

OrderedNamespace = _on.OrderedNamespace

__all__ = ["OrderedNamespace"]

"""Model classes of the plugin surface: what `avssl.model.<name>` resolves to."""
from . import kwClip as _kw

KWClip_GeneralTransformer = _kw.KWClip_GeneralTransformer
KWClipBase = _kw.KWClipBase

__all__ = ["KWClip_GeneralTransformer", "KWClipBase"]

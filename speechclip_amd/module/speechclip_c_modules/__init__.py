from . import vector_quantizers  # noqa: F401

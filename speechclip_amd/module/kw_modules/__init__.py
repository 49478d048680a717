from . import TransformerModels  # noqa: F401

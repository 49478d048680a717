from speechclip_amd.base import OrderedNamespace  # noqa: F401

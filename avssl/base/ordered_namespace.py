from speechclip_amd.base.ordered_namespace import OrderedNamespace  # noqa: F401  (pickle path of reference checkpoints)

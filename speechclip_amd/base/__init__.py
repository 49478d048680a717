from .ordered_namespace import OrderedNamespace

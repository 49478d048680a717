"""OrderedNamespace -- nested config object with attribute AND item access, list-merge construction, dict views and
pickling (it is stored in checkpoint hparams).  Same observable behaviour as avssl/base/ordered_namespace.py:7-153
(pinned by the reference's own test/test_dict.py semantics, restated in tests/test_host_logic.py); written
independently around a single recursive `_wrap` conversion."""
from argparse import Namespace
from collections import OrderedDict
from types import SimpleNamespace

_NS_TYPES = (SimpleNamespace, Namespace)


def _wrap(value):
    if isinstance(value, OrderedNamespace):
        return value
    if isinstance(value, (dict, OrderedDict)):
        return OrderedNamespace(value)
    if isinstance(value, _NS_TYPES):
        return OrderedNamespace(vars(value))
    if isinstance(value, list):
        return [OrderedNamespace(v) if isinstance(v, dict) else v for v in value]
    return value


class OrderedNamespace(object):
    def __init__(self, data=None, **kwargs):
        object.__setattr__(self, "_odict", OrderedDict())
        sources = []
        if isinstance(data, (tuple, list)):
            sources = list(data)
        elif data is not None:
            sources = [data]
        else:
            sources = [kwargs]
        for src in sources:
            if isinstance(src, OrderedNamespace):
                src = src._odict
            elif isinstance(src, _NS_TYPES):
                src = vars(src)
            for k, v in src.items():
                self._odict[k] = _wrap(v)

    # attribute / item access ------------------------------------------------------------------
    def __getattr__(self, key):
        od = object.__getattribute__(self, "_odict")
        if key in od:
            return od[key]
        raise AttributeError(key)

    def __setattr__(self, key, val):
        self._odict[key] = val

    def __getitem__(self, key):
        return self._odict[key]

    def __setitem__(self, key, val):
        self._odict[key] = val

    def __delitem__(self, key):
        del self._odict[key]

    def __contains__(self, key):
        return key in self._odict

    def __iter__(self):
        return iter(self._odict)

    def __len__(self):
        return len(self._odict)

    def keys(self):
        return self._odict.keys()

    def items(self):
        return self._odict.items()

    def values(self):
        return self._odict.values()

    def get(self, key, value=None):
        return self._odict.get(key, value)

    def copy(self):
        return self.__class__(self)

    # pickling / comparison --------------------------------------------------------------------
    def __getstate__(self):
        return self._odict

    def __setstate__(self, state):
        object.__setattr__(self, "_odict", OrderedDict())
        self._odict.update(state)

    def __eq__(self, other):
        return isinstance(other, OrderedNamespace) and self._odict == other._odict

    def __ne__(self, other):
        return not self.__eq__(other)

    # plain-container views ----------------------------------------------------------------------
    def _convert(self, factory):
        out = factory()
        for k, v in self._odict.items():
            out[k] = v._convert(factory) if isinstance(v, OrderedNamespace) else v
        return out

    def to_odict(self):
        return self._convert(OrderedDict)

    def to_dict(self):
        return self._convert(dict)

    odict = property(to_odict)
    pydict = property(to_dict)

    def __str__(self):
        return "OrderedNamespace(" + str(self.to_dict()) + ")"

    __repr__ = __str__

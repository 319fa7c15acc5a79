"""Stub of `numba` for importing the LibKGE reference as a CPU oracle.

Test infrastructure only (see oracle/README.md). `njit` becomes the identity
decorator, `typed.Dict` a plain dict; none of this is on the score arithmetic
path (reference uses numba only for index building / negative-sample filtering:
kge/indexing.py:59,115,415; kge/util/sampler.py:726).
"""
import types as _types

boolean = bool


def njit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]

    def deco(f):
        return f

    return deco


jit = njit
prange = range


class _TypedDict(dict):
    @classmethod
    def empty(cls, key_type=None, value_type=None):
        return cls()


typed = _types.SimpleNamespace(Dict=_TypedDict, List=list)
types = _types.SimpleNamespace(int32=int, int64=int, float32=float, float64=float)

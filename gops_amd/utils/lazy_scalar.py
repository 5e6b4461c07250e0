"""Loss scalars that stay on the GPU until somebody looks at them.

The reference's algorithms end every `local_update` with `loss.item()` for the log dict (gops/algorithm/fhadp.py:100-102,
infadp.py:153-157): a host sync per update.  On an MI355X an update of the fused kernels takes 0.2-0.7 ms, and the sync
leaves the GPU idle for the ~70 us the host needs to enqueue the next update (measured, profiles/r03_timeline_*).  The
trainers only read the log dict every `log_save_interval` iterations, so `tb_info` holds `LazyScalar`s: float-like
objects (float(), format, comparison, arithmetic, numpy conversion all work) that fetch their value from the device on
first use.  `GOPS_EAGER_LOG=1` restores plain Python floats (one sync per update, exactly like the reference).
"""
import os

import numpy as np

__all__ = ["LazyScalar", "lazy_enabled", "scalar"]


def lazy_enabled() -> bool:
    return os.environ.get("GOPS_EAGER_LOG", "0") in ("", "0")


class LazyScalar:
    __slots__ = ("_t", "_i", "_sign", "_mean", "_v", "_seen")

    def __init__(self, tensor, index=None, negate: bool = False, mean: bool = False, on_value=None):
        """`mean`: `tensor` is a vector whose mean is the value - the reduction kernel, too, only runs if the value is read.
        `on_value(v)`: called once, when the value is fetched (the algorithms hang their non-finite-loss watch on it: no extra sync)."""
        self._t, self._i, self._sign, self._mean, self._v, self._seen = tensor, index, (-1.0 if negate else 1.0), mean, None, on_value

    def _get(self) -> float:
        if self._v is None:
            t = self._t if self._i is None else self._t[self._i]
            if self._mean:
                t = t.mean()
            self._v = self._sign * float(t)   # the host sync happens here, once
            self._t = None
            if self._seen is not None:
                seen, self._seen = self._seen, None
                seen(self._v)
        return self._v

    def item(self) -> float:
        return self._get()

    __float__ = _get

    def __int__(self):
        return int(self._get())

    def __bool__(self):
        return bool(self._get())

    def __repr__(self):
        return repr(self._get())

    __str__ = __repr__

    def __format__(self, spec):
        return format(self._get(), spec)

    def __hash__(self):
        return hash(self._get())

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self._get(), dtype=dtype or np.float64)

    def __neg__(self):
        return -self._get()

    def __abs__(self):
        return abs(self._get())

    def __round__(self, n=None):
        return round(self._get(), n)

    def __hash__(self):
        return hash(self._get())

    def __reduce__(self):
        # copies, pickles and deep copies carry the VALUE (a plain float), not the device tensor behind it
        return (float, (self._get(),))


def _binary(name):
    def op(self, other):
        # like float: anything that is not a real number (None, a string, a tensor, ...) is NotImplemented - `entry == None` is
        # False and `entry in (None, ...)` works, as with the plain floats the reference's tb_info holds
        if isinstance(other, LazyScalar):
            other = other._get()
        elif isinstance(other, (bool, int, float, np.integer, np.floating)):
            other = float(other)
        else:
            return NotImplemented
        return getattr(float, name)(self._get(), other)
    op.__name__ = name
    return op


for _n in ("__eq__", "__ne__", "__lt__", "__le__", "__gt__", "__ge__", "__add__", "__radd__", "__sub__", "__rsub__", "__mul__",
           "__rmul__", "__truediv__", "__rtruediv__", "__pow__", "__rpow__", "__floordiv__", "__rfloordiv__", "__mod__", "__rmod__"):
    setattr(LazyScalar, _n, _binary(_n))


def scalar(tensor, index=None, negate: bool = False, mean: bool = False, on_value=None):
    """tb_info entry for a device scalar: lazy by default, a Python float (host sync now) under GOPS_EAGER_LOG=1."""
    s = LazyScalar(tensor, index, negate, mean, on_value)
    return s if lazy_enabled() else s.item()

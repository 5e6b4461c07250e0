"""One registry implementation behind all `create_*` factories.

GOPS's plugin surface is a set of string-keyed registries filled at import time by scanning package
directories, each with its own copy of the same Spec / register / lookup code.  Here that logic exists
once: a `Registry` knows what kind of thing it holds (for the error texts GOPS scripts may match on),
how to turn a module into entries, and how to look one up.
"""
import importlib
import os
from typing import Callable, Dict, Iterable, Optional, Tuple


class Entry:
    """A registered constructor plus default keyword arguments (GOPS calls this a Spec)."""

    __slots__ = ("key", "entry_point", "kwargs", "extra")

    def __init__(self, key: str, entry_point: Callable, kwargs: Optional[dict] = None, **extra):
        self.key, self.entry_point, self.kwargs, self.extra = key, entry_point, dict(kwargs or {}), extra

    def __getattr__(self, name):          # e.g. spec.approx_container_cls
        try:
            return self.extra[name]
        except KeyError:
            raise AttributeError(name) from None


class Registry(dict):
    """kind: the noun of the error messages ("algorithm", "apprfunc", "trainer", ...)."""

    def __init__(self, kind: str):
        super().__init__()
        self.kind = kind

    def add(self, key: str, entry_point: Callable, kwargs: Optional[dict] = None, **extra) -> Entry:
        self[key] = Entry(key, entry_point, kwargs, **extra)
        return self[key]

    def scan(self, directory: str, package: str, entries: Callable[[str, object], Iterable[Tuple[str, Callable, dict]]],
             keep: Callable[[str], bool] = lambda stem: True):
        """Import every public module `package.<stem>` of `directory` accepted by `keep` and register what
        `entries(stem, module)` yields as (key, constructor, extra-attributes)."""
        for file in sorted(os.listdir(directory)):
            stem, ext = os.path.splitext(file)
            if ext == ".py" and not stem.startswith("_") and stem != "base" and keep(stem):
                module = importlib.import_module(f"{package}.{stem}")
                for key, ctor, extra in entries(stem, module):
                    self.add(key, ctor, None, **extra)

    def lookup(self, key: str) -> Entry:
        entry = self.get(key)
        if entry is None:
            raise KeyError(f"No registered {self.kind} with id: {key}")
        return entry

    def build(self, key: str, *args, what: str = "entry_point", label: Optional[str] = None, **kwargs):
        """Construct `key` with its registered defaults overridden by `kwargs`."""
        entry = self.lookup(key)
        ctor = entry.entry_point if what == "entry_point" else getattr(entry, what, None)
        if not callable(ctor):
            raise RuntimeError(f"{label or entry.key} registered but {what} is not specified")
        return ctor(*args, **{**entry.kwargs, **kwargs})

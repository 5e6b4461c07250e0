"""create_apprfunc: registry `"<file>_<ClassName>"` -> class, filled from every module's
`__all__` in gops_amd/apprfunc (reference gops/create_pkg/create_apprfunc.py:43-72)."""
import importlib
import os
from dataclasses import dataclass, field
from typing import Callable, Dict

from gops_amd.utils.gops_path import apprfunc_path


@dataclass
class Spec:
    apprfunc: str
    name: str
    entry_point: Callable
    kwargs: dict = field(default_factory=dict)


registry: Dict[str, Spec] = {}


def register(apprfunc: str, name: str, entry_point: Callable, **kwargs):
    registry[apprfunc + "_" + name] = Spec(apprfunc=apprfunc, name=name, entry_point=entry_point, kwargs=kwargs)


for _file in sorted(os.listdir(apprfunc_path)):
    if _file.endswith(".py") and _file[0] != "_" and _file != "base.py":
        _mdl = importlib.import_module("gops_amd.apprfunc." + _file[:-3])
        for _name in _mdl.__all__:
            register(apprfunc=_file[:-3], name=_name, entry_point=getattr(_mdl, _name))


def create_apprfunc(**kwargs) -> object:
    apprfunc, name = kwargs["apprfunc"].lower(), kwargs["name"]
    spec_ = registry.get(apprfunc + "_" + name)
    if spec_ is None:
        raise KeyError(f"No registered apprfunc with id: {apprfunc}_{name}")
    _kwargs = spec_.kwargs.copy()
    _kwargs.update(kwargs)
    if not callable(spec_.entry_point):
        raise RuntimeError(f"{spec_.apprfunc}-{spec_.name} registered but entry_point is not specified")
    return spec_.entry_point(**_kwargs)

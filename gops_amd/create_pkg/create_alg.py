"""create_alg: registry of algorithms keyed by the upper-camel file name (`fhadp.py` -> "FHADP");
each module must export that class and `ApproxContainer` (reference
gops/create_pkg/create_alg.py:47-97).  Parallel trainers are one process per GPU here
(torch.distributed over RCCL), so every trainer kind receives a plain local algorithm object."""
import importlib
import os
from dataclasses import dataclass, field
from typing import Callable, Dict

from gops_amd.utils.gops_path import algorithm_path, underline2camel


@dataclass
class Spec:
    algorithm: str
    entry_point: Callable
    approx_container_cls: Callable
    kwargs: dict = field(default_factory=dict)


registry: Dict[str, Spec] = {}


def register(algorithm: str, entry_point: Callable, approx_container_cls: Callable, **kwargs):
    registry[algorithm] = Spec(algorithm=algorithm, entry_point=entry_point,
                               approx_container_cls=approx_container_cls, kwargs=kwargs)


for _file in sorted(os.listdir(algorithm_path)):
    if _file.endswith(".py") and _file[0] != "_" and _file != "base.py":
        _mdl = importlib.import_module("gops_amd.algorithm." + _file[:-3])
        _name = underline2camel(_file[:-3], first_upper=True)
        register(algorithm=_name, entry_point=getattr(_mdl, _name),
                 approx_container_cls=getattr(_mdl, "ApproxContainer"))


def _normalise(kwargs: dict) -> dict:
    if kwargs.get("seed") is None:
        kwargs["seed"] = 0
    if kwargs.get("cnn_shared") is None:
        kwargs["cnn_shared"] = False
    return kwargs


def create_alg(**kwargs) -> object:
    algorithm = kwargs["algorithm"]
    spec_ = registry.get(algorithm)
    if spec_ is None:
        raise KeyError(f"No registered algorithm with id: {algorithm}")
    _kwargs = _normalise({**spec_.kwargs, **kwargs})
    if not callable(spec_.entry_point):
        raise RuntimeError(f"{spec_.algorithm} registered but entry_point is not specified")
    trainer_name = _kwargs.get("trainer", None)
    known = ("off_serial", "on_serial", "on_sync", "off_sync", "off_async")
    if trainer_name is not None and not trainer_name.startswith(known):
        raise RuntimeError(f"trainer {trainer_name} not recognized")
    return spec_.entry_point(**_kwargs)


def create_approx_contrainer(algorithm: str, **kwargs) -> object:
    spec_ = registry.get(algorithm)
    if spec_ is None:
        raise KeyError(f"No registered algorithm with id: {algorithm}")
    _kwargs = _normalise({**spec_.kwargs, **kwargs})
    if not callable(spec_.approx_container_cls):
        raise RuntimeError(f"{spec_.algorithm} registered but approx_container_cls is not specified")
    return spec_.approx_container_cls(**_kwargs)

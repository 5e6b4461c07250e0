"""create_trainer: registry of `*trainer.py` modules keyed by file name, class =
underline2camel(name) (reference gops/create_pkg/create_trainer.py:44-73)."""
import importlib
import os
from dataclasses import dataclass, field
from typing import Callable, Dict

from gops_amd.utils.gops_path import trainer_path, underline2camel


@dataclass
class Spec:
    trainer: str
    entry_point: Callable
    kwargs: dict = field(default_factory=dict)


registry: Dict[str, Spec] = {}


def register(trainer: str, entry_point: Callable, **kwargs):
    registry[trainer] = Spec(trainer=trainer, entry_point=entry_point, kwargs=kwargs)


for _file in sorted(os.listdir(trainer_path)):
    if _file.endswith("trainer.py"):
        _name = _file[:-3]
        _mdl = importlib.import_module("gops_amd.trainer." + _name)
        register(trainer=_name, entry_point=getattr(_mdl, underline2camel(_name)))


def create_trainer(alg, sampler, buffer, evaluator, **kwargs) -> object:
    trainer_name = kwargs["trainer"]
    spec_ = registry.get(trainer_name)
    if spec_ is None:
        raise KeyError(f"No registered trainer with id: {trainer_name}")
    if not callable(spec_.entry_point):
        raise RuntimeError(f"{spec_.trainer} registered but entry_point is not specified")
    if spec_.trainer.startswith("off"):
        return spec_.entry_point(alg, sampler, buffer, evaluator, **kwargs)
    if spec_.trainer.startswith("on"):
        return spec_.entry_point(alg, sampler, evaluator, **kwargs)
    raise RuntimeError(f"trainer {spec_.trainer} not recognized")

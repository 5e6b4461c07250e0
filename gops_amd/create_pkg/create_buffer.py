"""create_buffer: registry of the buffer modules under gops_amd/trainer/buffer keyed by file name,
class = underline2camel(name) (reference gops/create_pkg/create_buffer.py:28-68).  On-policy trainers
get no buffer, as in the reference (:54-56)."""
import importlib
import os
from dataclasses import dataclass, field
from typing import Callable, Dict

from gops_amd.utils.gops_path import underline2camel

buffer_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "trainer", "buffer")


@dataclass
class Spec:
    buffer_name: str
    entry_point: Callable
    kwargs: dict = field(default_factory=dict)


registry: Dict[str, Spec] = {}


def register(buffer_name: str, entry_point: Callable, **kwargs):
    registry[buffer_name] = Spec(buffer_name=buffer_name, entry_point=entry_point, kwargs=kwargs)


for _file in sorted(os.listdir(buffer_path)):
    if _file.endswith(".py") and _file[0] != "_":
        _name = _file[:-3]
        _mdl = importlib.import_module("gops_amd.trainer.buffer." + _name)
        register(buffer_name=_name, entry_point=getattr(_mdl, underline2camel(_name)))


def create_buffer(**kwargs):
    trainer_name = kwargs.get("trainer", None)
    if trainer_name is not None and trainer_name.startswith("on"):
        return None
    buffer_name = kwargs["buffer_name"]
    spec_ = registry.get(buffer_name)
    if spec_ is None:
        raise KeyError(f"No registered buffer with id: {buffer_name}")
    if not callable(spec_.entry_point):
        raise RuntimeError(f"{spec_.buffer_name} registered but entry_point is not specified")
    return spec_.entry_point(**kwargs)

"""create_buffer: buffer modules under gops_amd/trainer/buffer keyed by file name, class =
underline2camel(name); on-policy trainers get no buffer (gops/create_pkg/create_buffer.py:28-68)."""
import os

from gops_amd.create_pkg._registry import Registry
from gops_amd.utils.gops_path import trainer_path, underline2camel

registry = Registry("buffer")
buffer_path = os.path.join(trainer_path, "buffer")


def register(buffer_name: str, entry_point, **kwargs):
    registry.add(buffer_name, entry_point, kwargs, buffer_name=buffer_name)


registry.scan(buffer_path, "gops_amd.trainer.buffer",
              lambda stem, module: [(stem, getattr(module, underline2camel(stem)), dict(buffer_name=stem))])


def create_buffer(**kwargs):
    trainer = kwargs.get("trainer")
    if trainer is not None and trainer.startswith("on"):
        return None
    return registry.build(kwargs["buffer_name"], **kwargs)

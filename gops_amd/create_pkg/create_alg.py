"""create_alg: algorithms keyed by the upper-camel file name (`fhadp.py` -> "FHADP"); each module exports
that class and `ApproxContainer` (surface of gops/create_pkg/create_alg.py:47-97).  Parallel trainers are
one process per GPU here (torch.distributed over RCCL): every trainer kind gets this rank's local object - for the
off_sync / off_async trainers behind the list-of-actor-handles shape the reference's scripts expect (`LocalActor`)."""
import inspect

from gops_amd.create_pkg._registry import Registry
from gops_amd.utils.gops_path import algorithm_path, underline2camel

registry = Registry("algorithm")
_TRAINER_KINDS = ("off_serial", "on_serial", "on_sync", "off_sync", "off_async")


def register(algorithm: str, entry_point, approx_container_cls, **kwargs):
    registry.add(algorithm, entry_point, kwargs, algorithm=algorithm, approx_container_cls=approx_container_cls)


def _entries(stem, module):
    name = underline2camel(stem, first_upper=True)
    yield name, getattr(module, name), dict(algorithm=name, approx_container_cls=getattr(module, "ApproxContainer"))


registry.scan(algorithm_path, "gops_amd.algorithm", _entries)


def _defaults(kwargs: dict) -> dict:
    out = dict(kwargs)
    if out.get("seed") is None:
        out["seed"] = 0
    if out.get("cnn_shared") is None:
        out["cnn_shared"] = False
    return out


class _RemoteMethod:
    """`handle.method.remote(*args)` of a Ray actor, in-process: returns the value itself."""

    def __init__(self, fn):
        self._fn = fn

    def __call__(self, *args, **kwargs):
        return self._fn(*args, **kwargs)

    def remote(self, *args, **kwargs):
        return self._fn(*args, **kwargs)


class LocalActor:
    """What the reference's example scripts expect from `create_alg` for the off_sync / off_async trainers: a list of
    actor handles whose methods are reached through `.remote(...)` (create_alg.py:87-93, e.g.
    `for alg_id in alg: alg_id.set_parameters.remote({...})`, example_train/mac/mac_mlp_cartpoleconti_async.py:153-154).
    One process per GPU here: the list has ONE entry, this rank's replica; the trainers unwrap it (`.unwrap()`)."""

    def __init__(self, obj):
        object.__setattr__(self, "_obj", obj)

    def unwrap(self):
        return self._obj

    def __getattr__(self, name):
        attr = getattr(self._obj, name)
        return _RemoteMethod(attr) if inspect.ismethod(attr) or inspect.isfunction(attr) else attr


def create_alg(**kwargs) -> object:
    registry.lookup(kwargs["algorithm"])          # unknown algorithm: KeyError before anything else
    trainer = kwargs.get("trainer")
    if trainer is not None and not trainer.startswith(_TRAINER_KINDS):
        raise RuntimeError(f"trainer {trainer} not recognized")
    alg = registry.build(kwargs["algorithm"], **_defaults(kwargs))
    if trainer is not None and trainer.startswith(("off_async", "off_sync")):
        return [LocalActor(alg)]                   # the reference returns a list of actor handles for these trainers
    return alg


def create_approx_contrainer(algorithm: str, **kwargs) -> object:
    return registry.build(algorithm, what="approx_container_cls", **_defaults(kwargs))

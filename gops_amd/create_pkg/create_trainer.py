"""create_trainer: `*trainer.py` modules of gops_amd/trainer keyed by file name, class =
underline2camel(name); off-policy trainers also receive the buffer (gops/create_pkg/create_trainer.py:44-73)."""
from gops_amd.create_pkg._registry import Registry
from gops_amd.utils.gops_path import trainer_path, underline2camel

registry = Registry("trainer")


def register(trainer: str, entry_point, **kwargs):
    registry.add(trainer, entry_point, kwargs, trainer=trainer)


registry.scan(trainer_path, "gops_amd.trainer",
              lambda stem, module: [(stem, getattr(module, underline2camel(stem)), dict(trainer=stem))],
              keep=lambda stem: stem.endswith("trainer"))


def create_trainer(alg, sampler, buffer, evaluator, **kwargs) -> object:
    name = kwargs["trainer"]
    registry.lookup(name)
    if name.startswith("off"):
        return registry.build(name, alg, sampler, buffer, evaluator, **kwargs)
    if name.startswith("on"):
        return registry.build(name, alg, sampler, evaluator, **kwargs)
    raise RuntimeError(f"trainer {name} not recognized")

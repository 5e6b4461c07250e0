"""Overlay `gops_amd` onto an installed GOPS tree: `import gops.<hot-path module>` resolves to the
MI355X implementation, everything else (`gops.trainer.sampler`, buffers, evaluator, plotting ...)
keeps coming from GOPS itself.  This is the "drop-in for that path and nothing else" switch:

    import gops_amd.overlay; gops_amd.overlay.install()      # before the script's `from gops...` lines
    from gops.create_pkg.create_alg import create_alg        # -> gops_amd.create_pkg.create_alg

so `example_train/fhadp/*.py` run unchanged apart from that first line (or with
`python -c "import gops_amd.overlay as o; o.install(); import runpy; runpy.run_path(...)"`).
"""
import importlib
import importlib.abc
import importlib.util
import sys

# reference module -> implementation (the functions on the hot path)
OVERLAY = {
    "gops.create_pkg.create_alg": "gops_amd.create_pkg.create_alg",
    "gops.create_pkg.create_apprfunc": "gops_amd.create_pkg.create_apprfunc",
    "gops.create_pkg.create_env_model": "gops_amd.create_pkg.create_env_model",
    "gops.create_pkg.create_trainer": "gops_amd.create_pkg.create_trainer",
    "gops.create_pkg.create_buffer": "gops_amd.create_pkg.create_buffer",
    "gops.algorithm.base": "gops_amd.algorithm.base",
    "gops.algorithm.fhadp": "gops_amd.algorithm.fhadp",
    "gops.algorithm.fhadp2": "gops_amd.algorithm.fhadp2",
    "gops.algorithm.fhadp_exterior": "gops_amd.algorithm.fhadp_exterior",
    "gops.algorithm.fhadp_interior": "gops_amd.algorithm.fhadp_interior",
    "gops.algorithm.fhadp_lagrangian": "gops_amd.algorithm.fhadp_lagrangian",
    "gops.algorithm.infadp": "gops_amd.algorithm.infadp",
    "gops.algorithm.mac": "gops_amd.algorithm.mac",
    "gops.algorithm.spil": "gops_amd.algorithm.spil",
    "gops.algorithm.mpg": "gops_amd.algorithm.mpg",
    "gops.apprfunc.mlp": "gops_amd.apprfunc.mlp",
    # NOT overlaid: gops.env.env_ocp.env_model.pyth_*_model.  The reference's DATA envs import helpers from those
    # modules (`from ...env_model.pyth_idpendulum_model import Dynamics`, pyth_idpendulum.py:20), and nothing on the
    # hot path imports them by name: `create_env_model` (overlaid above) builds this package's models from its own
    # registry.
    "gops.trainer.on_serial_trainer": "gops_amd.trainer.on_serial_trainer",
    "gops.trainer.on_sync_trainer": "gops_amd.trainer.on_sync_trainer",
    "gops.trainer.off_serial_trainer": "gops_amd.trainer.off_serial_trainer",
    "gops.trainer.off_sync_trainer": "gops_amd.trainer.off_sync_trainer",
    "gops.trainer.off_async_trainer": "gops_amd.trainer.off_async_trainer",
    "gops.trainer.buffer.replay_buffer": "gops_amd.trainer.buffer.replay_buffer",
}


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, target):
        self.target = target

    def create_module(self, spec):
        return importlib.import_module(self.target)   # the alias IS the implementation module

    def exec_module(self, module):
        pass


class _OverlayFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        impl = OVERLAY.get(fullname)
        if impl is None:
            return None
        return importlib.util.spec_from_loader(fullname, _AliasLoader(impl))


_finder = None


def install():
    """Idempotent; must run before the overlaid `gops.*` modules are first imported."""
    global _finder
    if _finder is None:
        _finder = _OverlayFinder()
        sys.meta_path.insert(0, _finder)
    already = [m for m in OVERLAY if m in sys.modules and sys.modules[m].__name__ != OVERLAY[m]]
    if already:
        raise RuntimeError(f"gops_amd.overlay.install() came too late, already imported: {already}")


def uninstall():
    global _finder
    if _finder is not None:
        sys.meta_path.remove(_finder)
        _finder = None
    for m in OVERLAY:
        sys.modules.pop(m, None)

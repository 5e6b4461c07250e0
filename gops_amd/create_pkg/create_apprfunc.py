"""create_apprfunc: `"<file>_<ClassName>"` -> class for every name in a module's `__all__` under
gops_amd/apprfunc (same keys, kwargs and error behaviour as gops/create_pkg/create_apprfunc.py:43-72)."""
from gops_amd.create_pkg._registry import Registry
from gops_amd.utils.gops_path import apprfunc_path

registry = Registry("apprfunc")


def register(apprfunc: str, name: str, entry_point, **kwargs):
    registry.add(f"{apprfunc}_{name}", entry_point, kwargs, apprfunc=apprfunc, name=name)


registry.scan(apprfunc_path, "gops_amd.apprfunc",
              lambda stem, module: ((f"{stem}_{cls}", getattr(module, cls), dict(apprfunc=stem, name=cls))
                                    for cls in module.__all__))


def create_apprfunc(**kwargs) -> object:
    apprfunc, name = kwargs["apprfunc"].lower(), kwargs["name"]
    return registry.build(f"{apprfunc}_{name}", label=f"{apprfunc}-{name}", **kwargs)

"""Alias of the :mod:`kvmini` package under the distribution's long name.

``kserve_vllm_mini_amd.ops`` (``.models``, ``.parallel``, ``.utils``, ``.engine``, …) are lazy
proxies of ``kvmini.ops`` etc.: attribute access resolves to the very same objects, importing
submodules on demand (``from kserve_vllm_mini_amd.ops import hist_quantile`` is
``kvmini.ops.hist_quantile``).  The framework itself is ONE package, ``kvmini`` — the name the
reference project's CLI and Python API use; this module only maps the long name onto it."""

import importlib
import sys
import types

import kvmini as _kvmini

__version__ = getattr(_kvmini, "__version__", "0")
_SUBPACKAGES = ("analysis", "cli", "cluster", "config", "cost", "deploy", "engine", "gates", "harness", "loadgen", "models", "ops", "parallel", "report", "server", "telemetry", "utils")


class _LazyAlias(types.ModuleType):
    """Stands in for ``kvmini.<name>`` until first use, then forwards every attribute."""

    def __init__(self, alias: str, real: str):
        super().__init__(alias, f"alias of {real}")
        self.__dict__["_real_name"] = real

    def _real(self) -> types.ModuleType:
        return importlib.import_module(self.__dict__["_real_name"])

    def __getattr__(self, name: str):
        real = self._real()
        try:
            return getattr(real, name)
        except AttributeError:
            try:  # a submodule that has not been imported yet
                return importlib.import_module(f"{real.__name__}.{name}")
            except ImportError:
                raise AttributeError(f"module {real.__name__!r} has no attribute {name!r}") from None

    def __dir__(self):
        return dir(self._real())


for _name in _SUBPACKAGES:
    _alias = _LazyAlias(f"{__name__}.{_name}", f"kvmini.{_name}")
    sys.modules[_alias.__name__] = _alias
    globals()[_name] = _alias


def __getattr__(name: str):
    return getattr(_kvmini, name)

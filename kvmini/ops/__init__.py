"""Bindings of the hand-written gfx950 (MI355X) HIP kernels.

``kvmini._C`` is a torch extension built in-tree by ``build_native.py`` (or
``__graft_entry__.build()``).  :func:`native` returns the module or raises with a
build hint; nothing here silently falls back to PyTorch on a GPU box.
"""

from __future__ import annotations

from typing import Any, Optional

_C: Any = None
_ERR: Optional[BaseException] = None


def native() -> Any:
    """The ``kvmini._C`` extension module (imported lazily)."""
    global _C, _ERR
    if _C is not None:
        return _C
    if _ERR is not None:
        raise RuntimeError(f"kvmini native extension unavailable: {_ERR}; run `python build_native.py`") from _ERR
    try:
        import torch  # noqa: F401  (libtorch must be loaded first)

        from kvmini import _C as mod  # type: ignore[attr-defined]
    except Exception as e:  # ImportError, OSError (missing .so / unresolved symbol)
        _ERR = e
        raise RuntimeError(f"kvmini native extension unavailable: {e}; run `python build_native.py`") from e
    _C = mod
    return mod


def native_built() -> bool:
    try:
        native()
        return True
    except RuntimeError:
        return False


def gpu_available() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False

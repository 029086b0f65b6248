"""Quantiles with one definition and three engines.

Semantics (reference ``analyze.py:59-81``): drop NaNs, sort, ``pos = p*(n-1)``,
linear interpolation between ``floor(pos)`` and ``ceil(pos)``; ``n == 1``
returns the value, empty input returns NaN.  The reference has two more,
inconsistent, definitions in its side tools — a floor-index rolling p95
(``scripts/fairness_dual_tenant.py:59-65``, ``k = int(0.95*(n-1))``) and
``sorted[int(p*n)]`` in the probes (``tools/net_storage_probe.py:39-40``).
Here they are all one routine with a ``method`` switch.

Engines:

* ``python`` — exact, dependency-free (the oracle).
* ``numpy``  — exact, vectorised (``np.partition``-free: a sort, like the oracle).
* ``hip``    — the gfx950 LDS-binned histogram + exact in-bin refinement kernel in
  ``csrc/hist_quantile.hip`` (``kvmini.ops.hist_quantile``).  Picked automatically
  for large sample sets when a GPU and the built extension are present.
"""

from __future__ import annotations

import bisect
import collections

import math
from typing import Iterable, List, Optional, Sequence

METHODS = ("linear", "floor", "nearest_rank")

# Below this many samples the GPU launch + copy costs more than a CPU sort.
HIP_MIN_SAMPLES = 1 << 16


def _clean(arr: Iterable[float]) -> List[float]:
    out = []
    for x in arr:
        if x is None:
            continue
        xf = float(x)
        if not math.isnan(xf):
            out.append(xf)
    return out


def _from_sorted(vals: Sequence[float], p: float, method: str) -> float:
    n = len(vals)
    if n == 0:
        return float("nan")
    if n == 1:
        return float(vals[0])
    p = max(0.0, min(float(p), 1.0))
    if method == "linear":
        pos = p * (n - 1)
        lo = int(math.floor(pos))
        hi = int(math.ceil(pos))
        if lo == hi:
            return float(vals[lo])
        frac = pos - lo
        return float(vals[lo] + frac * (vals[hi] - vals[lo]))
    if method == "floor":
        return float(vals[int(p * (n - 1))])
    if method == "nearest_rank":
        return float(vals[min(n - 1, int(p * n))])
    raise ValueError(f"unknown quantile method {method!r}; expected one of {METHODS}")


def percentile(arr: Iterable[float], p: float, method: str = "linear") -> float:
    """Exact p-quantile (0 <= p <= 1) of ``arr`` — the pure-Python oracle."""
    vals = _clean(arr)
    vals.sort()
    return _from_sorted(vals, p, method)


def percentiles(
    arr: Iterable[float],
    ps: Sequence[float] = (0.5, 0.95, 0.99),
    method: str = "linear",
    engine: str = "auto",
) -> List[float]:
    """Several quantiles of one sample set with a single sort / single GPU pass.

    ``engine``: ``auto`` | ``python`` | ``numpy`` | ``hip``.  ``auto`` uses the HIP
    kernel for >= ``HIP_MIN_SAMPLES`` samples when it is available, else numpy.
    """
    if engine not in ("auto", "python", "numpy", "hip"):
        raise ValueError(f"unknown engine {engine!r}")
    if engine == "python":
        vals = _clean(arr)
        vals.sort()
        return [_from_sorted(vals, p, method) for p in ps]

    import numpy as np

    a = np.asarray(arr if not isinstance(arr, (list, tuple)) else [np.nan if v is None else v for v in arr], dtype=np.float64)
    a = a[~np.isnan(a)]

    use_hip = engine == "hip"
    if engine == "auto" and a.size >= HIP_MIN_SAMPLES and method == "linear":
        from kvmini.ops import hist_quantile as hq

        use_hip = hq.available()
    if use_hip:
        from kvmini.ops import hist_quantile as hq

        if method != "linear":
            raise ValueError("the HIP engine implements the 'linear' definition only")
        return hq.quantiles(a, ps)

    a.sort()
    return [_from_sorted(a, p, method) for p in ps]


class RollingQuantile:
    """Windowed quantile over the most recent ``window`` observations.

    Default ``method='floor'`` reproduces the fairness guard's rolling p95
    (reference ``scripts/fairness_dual_tenant.py:46-65``, which sorts the window on every
    read).  Here the window is kept twice: a deque in arrival order and a list that stays
    sorted — an observation enters by binary insertion and the one that falls out of the window
    leaves by binary search, so ``add`` is O(log w) comparisons (plus the list shift) and
    ``value`` reads the order statistic directly; nothing is re-sorted per response."""

    def __init__(self, window: int = 50, p: float = 0.95, method: str = "floor"):
        if window < 1:
            raise ValueError("window must be >= 1")
        self.window = int(window)
        self.p = float(p)
        self.method = method
        self._fifo: "collections.deque[float]" = collections.deque()
        self._sorted: List[float] = []

    def add(self, x: float) -> None:
        x = float(x)
        self._fifo.append(x)
        bisect.insort(self._sorted, x)
        if len(self._fifo) > self.window:
            old = self._fifo.popleft()
            del self._sorted[bisect.bisect_left(self._sorted, old)]

    def value(self) -> Optional[float]:
        if not self._sorted:
            return None
        return _from_sorted(self._sorted, self.p, self.method)

    def __len__(self) -> int:
        return len(self._fifo)


def rolling_quantile(window: int = 50, p: float = 0.95, method: str = "floor", engine: str = "auto"):
    """A rolling quantile: the C++ one (``csrc/host_routines.cpp``; the window never touches the
    interpreter) for windows of 512 and more when the host runtime is built, else
    :class:`RollingQuantile`.  ``engine``: ``auto`` | ``python`` | ``native``."""
    if engine not in ("auto", "python", "native"):
        raise ValueError(f"unknown engine {engine!r}")
    if engine == "native" or (engine == "auto" and window >= 512):
        from . import native_host

        if native_host.available():
            return native_host.NativeRollingQuantile(window, p, method)
        if engine == "native":
            raise RuntimeError("libkvmini_rt.so is not built (python build_native.py --only rt)")
    return RollingQuantile(window, p, method)

"""ctypes front end of the C++ host routines in ``csrc/host_routines.cpp`` (part of
``libkvmini_rt.so``): non-dominated filter, cold/warm classification, exact quantiles and a
rolling quantile.

Every function here has a pure-Python / numpy twin that defines its semantics
(``ParetoAnalyzer.frontier_mask``, ``quantization_sweep.pareto_mask``,
``coldwarm.classify_requests_cold_warm``, ``quantiles.percentiles`` / ``RollingQuantile``); the
callers switch to the native routine above a size where it pays (``MIN_ROWS``) and fall back
to the twin when the library is not built, so results never depend on which one ran
(``tests/test_native_host.py`` pins equality).
"""

from __future__ import annotations

import ctypes
from typing import Optional, Sequence

import numpy as np

from kvmini.telemetry import native_rt

METHOD_CODES = {"linear": 0, "floor": 1, "nearest_rank": 2}
# below these sizes the numpy twin is as fast as the call overhead
MIN_ROWS_PARETO = 64
MIN_ROWS_COLD = 2048

_READY: Optional[bool] = None


def _dp(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _u8(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


def lib() -> Optional[ctypes.CDLL]:
    """The runtime library with the host routines' prototypes set, or None (not built, or built
    from a source tree that predates them)."""
    global _READY
    L = native_rt.load()
    if L is None or _READY is False:
        return None
    if _READY:
        return L
    try:
        dp, u8, i64 = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint8), ctypes.c_longlong
        L.kvh_pareto_mask.argtypes = [dp, i64, i64, ctypes.c_int, u8]
        L.kvh_pareto_mask.restype = ctypes.c_int
        L.kvh_classify_cold.argtypes = [dp, dp, i64, dp, i64, ctypes.c_double, ctypes.c_int, u8]
        L.kvh_classify_cold.restype = ctypes.c_int
        L.kvh_quantiles.argtypes = [dp, i64, dp, i64, ctypes.c_int, dp]
        L.kvh_quantiles.restype = i64
        L.kvh_rq_new.argtypes = [i64, ctypes.c_double, ctypes.c_int]
        L.kvh_rq_new.restype = ctypes.c_void_p
        L.kvh_rq_free.argtypes = [ctypes.c_void_p]
        L.kvh_rq_free.restype = None
        L.kvh_rq_add.argtypes = [ctypes.c_void_p, ctypes.c_double]
        L.kvh_rq_add.restype = ctypes.c_double
        L.kvh_rq_add_many.argtypes = [ctypes.c_void_p, dp, i64, dp]
        L.kvh_rq_add_many.restype = ctypes.c_int
        L.kvh_rq_value.argtypes = [ctypes.c_void_p]
        L.kvh_rq_value.restype = ctypes.c_double
        L.kvh_rq_len.argtypes = [ctypes.c_void_p]
        L.kvh_rq_len.restype = i64
    except AttributeError:
        _READY = False
        return None
    _READY = True
    return L


def available() -> bool:
    return lib() is not None


def pareto_mask(points: np.ndarray, nan_neutral: bool = False) -> np.ndarray:
    """bool[n]: rows of ``points[n, d]`` (lower is better on every axis) that no other row
    dominates."""
    L = lib()
    if L is None:
        raise RuntimeError("libkvmini_rt.so is not built (python build_native.py --only rt)")
    p = np.ascontiguousarray(points, dtype=np.float64)
    if p.ndim != 2:
        raise ValueError("points must be [n, d]")
    out = np.zeros(p.shape[0], dtype=np.uint8)
    if p.shape[0] and L.kvh_pareto_mask(_dp(p), p.shape[0], p.shape[1], int(bool(nan_neutral)), _u8(out)) != 0:
        raise RuntimeError("kvh_pareto_mask rejected its arguments")
    return out.astype(bool)


def classify_cold(start_s: Sequence[float], latency_s: Optional[Sequence[float]], cold_sorted: Sequence[float], window_s: float, spanning: bool) -> np.ndarray:
    """bool[n] cold flags; ``cold_sorted`` ascending."""
    L = lib()
    if L is None:
        raise RuntimeError("libkvmini_rt.so is not built (python build_native.py --only rt)")
    s = np.ascontiguousarray(start_s, dtype=np.float64)
    c = np.ascontiguousarray(cold_sorted, dtype=np.float64)
    lat = np.ascontiguousarray(latency_s, dtype=np.float64) if latency_s is not None else None
    if lat is not None and lat.shape != s.shape:
        raise ValueError("latency_s must match start_s")
    out = np.zeros(s.shape[0], dtype=np.uint8)
    rc = L.kvh_classify_cold(_dp(s), _dp(lat) if lat is not None else None, s.shape[0], _dp(c) if c.size else None, c.shape[0], float(window_s), int(bool(spanning)), _u8(out))
    if rc != 0:
        raise RuntimeError("kvh_classify_cold rejected its arguments")
    return out.astype(bool)


def quantiles(values: Sequence[float], ps: Sequence[float], method: str = "linear") -> list:
    L = lib()
    if L is None:
        raise RuntimeError("libkvmini_rt.so is not built (python build_native.py --only rt)")
    v = np.ascontiguousarray(values, dtype=np.float64)
    p = np.ascontiguousarray(ps, dtype=np.float64)
    out = np.zeros(p.shape[0], dtype=np.float64)
    if L.kvh_quantiles(_dp(v) if v.size else None, v.shape[0], _dp(p), p.shape[0], METHOD_CODES[method], _dp(out)) < 0:
        raise RuntimeError("kvh_quantiles rejected its arguments")
    return [float(x) for x in out]


class NativeRollingQuantile:
    """Same interface as :class:`kvmini.analysis.quantiles.RollingQuantile`, window held in C++."""

    def __init__(self, window: int = 50, p: float = 0.95, method: str = "floor"):
        if window < 1:
            raise ValueError("window must be >= 1")
        L = lib()
        if L is None:
            raise RuntimeError("libkvmini_rt.so is not built (python build_native.py --only rt)")
        self._lib = L
        self.window, self.p, self.method = int(window), float(p), method
        self._h = L.kvh_rq_new(self.window, self.p, METHOD_CODES[method])
        if not self._h:
            raise RuntimeError("kvh_rq_new failed")

    def add(self, x: float) -> float:
        return float(self._lib.kvh_rq_add(self._h, float(x)))

    def add_many(self, xs: Sequence[float]) -> np.ndarray:
        a = np.ascontiguousarray(xs, dtype=np.float64)
        out = np.zeros_like(a)
        if a.size and self._lib.kvh_rq_add_many(self._h, _dp(a), a.shape[0], _dp(out)) != 0:
            raise RuntimeError("kvh_rq_add_many rejected its arguments")
        return out

    def value(self) -> Optional[float]:
        if len(self) == 0:
            return None
        return float(self._lib.kvh_rq_value(self._h))

    def __len__(self) -> int:
        return int(self._lib.kvh_rq_len(self._h))

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.kvh_rq_free(self._h)
            self._h = None

    def __del__(self) -> None:
        try:
            self.close()
        except Exception:
            pass

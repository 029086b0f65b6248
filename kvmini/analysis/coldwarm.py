"""Cold/warm request classification and per-class metrics.

Semantics (reference ``analyze.py:402-460``): a request is *cold* when its
``start_ms`` falls inside ``[t_cold, t_cold + window]`` for any cold-start time
(30 s window by default).  The reference scans every (request, cold start)
pair; here cold starts are sorted once and each request does one bisect, which
is the same predicate in O((N + C) log C).

Per-class metrics keep the reference's asymmetric shape: an empty class emits
only ``_count`` and null ``_p50/_p95/_p99``.
"""

from __future__ import annotations

import bisect
import math
from typing import Any, Dict, List, Sequence

from .quantiles import percentiles


def is_success(row: Dict[str, Any]) -> bool:
    try:
        return int(float(row.get("status", 0) or 0)) == 200
    except (TypeError, ValueError):
        return False


_NATIVE_MIN_ROWS = 2048  # below this the per-row bisect is as fast as marshalling the columns


def _seconds(row: Dict[str, Any], key: str) -> float:
    try:
        return float(row.get(key, 0) or 0) / 1000.0
    except (TypeError, ValueError):
        if key == "start_ms":
            raise
        return 0.0


def _classify_native(rows: List[Dict[str, Any]], cold_sorted: List[float], window: float, spanning: bool):
    """The same predicate in C++ (csrc/host_routines.cpp) on the two columns; None when the
    host runtime library is not built."""
    from . import native_host

    if not native_host.available():
        return None
    starts = [_seconds(r, "start_ms") for r in rows]
    lats = [_seconds(r, "latency_ms") for r in rows] if spanning else None
    return native_host.classify_cold(starts, lats, cold_sorted, window, spanning)


def classify_requests_cold_warm(
    rows: List[Dict[str, Any]], cold_start_times: Sequence[float], cold_window_sec: float = 30.0, spanning: bool = True
) -> List[Dict[str, Any]]:
    """``spanning`` (an addition): a request that was already in flight when a cold start
    happened (``start < t_cold < start + latency``: it sat in the activator / waited for the
    restarted server) is cold as well — under the reference's predicate the request that
    paid for the whole cold start counts as warm."""
    cold_sorted = sorted(float(t) for t in cold_start_times)
    if len(rows) >= _NATIVE_MIN_ROWS and cold_sorted:
        flags = _classify_native(rows, cold_sorted, cold_window_sec, spanning)
        if flags is not None:
            for row, f in zip(rows, flags):
                row["is_cold_start"] = bool(f)
            return rows
    for row in rows:
        t = float(row.get("start_ms", 0) or 0) / 1000.0
        cold = False
        if cold_sorted:
            # rightmost cold start <= t
            i = bisect.bisect_right(cold_sorted, t) - 1
            if i >= 0 and t <= cold_sorted[i] + cold_window_sec:
                cold = True
            elif spanning:
                try:
                    lat = float(row.get("latency_ms", 0) or 0) / 1000.0
                except (TypeError, ValueError):
                    lat = 0.0
                j = bisect.bisect_right(cold_sorted, t)  # first cold start after the request began
                if lat > 0 and j < len(cold_sorted) and cold_sorted[j] < t + lat:
                    cold = True
        row["is_cold_start"] = cold
    return rows


def _subset_metrics(subset: List[Dict[str, Any]], prefix: str) -> Dict[str, Any]:
    if not subset:
        return {
            f"{prefix}_count": 0,
            f"{prefix}_p50_ms": None,
            f"{prefix}_p95_ms": None,
            f"{prefix}_p99_ms": None,
        }
    ok = [r for r in subset if is_success(r)]
    lats = [float(r.get("latency_ms", float("nan"))) for r in ok]
    ttfs = []
    for r in ok:
        v = r.get("ttfb_ms")
        if isinstance(v, (int, float)) and not math.isnan(v):
            ttfs.append(float(v))
    p50 = p95 = p99 = None
    if lats:
        p50, p95, p99 = percentiles(lats, (0.5, 0.95, 0.99))
    t50 = t95 = None
    if ttfs:
        t50, t95 = percentiles(ttfs, (0.5, 0.95))
    return {
        f"{prefix}_count": len(subset),
        f"{prefix}_success_count": len(ok),
        f"{prefix}_p50_ms": p50,
        f"{prefix}_p95_ms": p95,
        f"{prefix}_p99_ms": p99,
        f"{prefix}_ttft_p50_ms": t50,
        f"{prefix}_ttft_p95_ms": t95,
        f"{prefix}_error_rate": (len(subset) - len(ok)) / len(subset),
    }


def compute_cold_warm_metrics(rows: List[Dict[str, Any]]) -> Dict[str, Any]:
    cold = [r for r in rows if r.get("is_cold_start")]
    warm = [r for r in rows if not r.get("is_cold_start")]
    out = _subset_metrics(cold, "cold")
    out.update(_subset_metrics(warm, "warm"))
    c95, w95 = out.get("cold_p95_ms"), out.get("warm_p95_ms")
    if c95 and w95:
        out["cold_multiplier"] = c95 / w95
    return out

"""Equal-width histograms for the timing section of results.json.

Semantics (reference ``analyze.py:84-122``): NaNs dropped; 20 equal-width bins
over ``[min, max]``; the last bin is closed on the right; output keys
``bins`` (centres), ``counts``, ``bin_edges`` and — when not degenerate —
``min``, ``max``, ``total_samples``.  ``min == max`` yields a single bin.
"""

from __future__ import annotations

import math
from typing import Any, Dict, Iterable, List


def compute_histogram(data: Iterable[float], num_bins: int = 20, engine: str = "auto") -> Dict[str, Any]:
    clean: List[float] = []
    for x in data:
        if x is None:
            continue
        xf = float(x)
        if not math.isnan(xf):
            clean.append(xf)
    if not clean:
        return {"bins": [], "counts": [], "bin_edges": []}

    if engine in ("auto", "hip") and (engine == "hip" or len(clean) >= (1 << 16)):
        from kvmini.ops import hist_quantile as hq

        if hq.available():
            return hq.equal_width_histogram(clean, num_bins)
        if engine == "hip":
            raise RuntimeError("HIP histogram requested but kvmini native extension/GPU is unavailable")

    lo = min(clean)
    hi = max(clean)
    if lo == hi:
        return {"bins": [lo], "counts": [len(clean)], "bin_edges": [lo, hi]}
    width = (hi - lo) / num_bins
    edges = [lo + i * width for i in range(num_bins + 1)]
    counts = [0] * num_bins
    for v in clean:
        counts[min(num_bins - 1, int((v - lo) / width))] += 1
    centres = [(edges[i] + edges[i + 1]) / 2 for i in range(num_bins)]
    return {
        "bins": centres,
        "counts": counts,
        "bin_edges": edges,
        "min": lo,
        "max": hi,
        "total_samples": len(clean),
    }


# name used by the reference's inline CI checks (.github/workflows/lint-test.yml:112-150)
compute_histograms = compute_histogram

"""Unit pricing shared by the cost estimator, the planner and bench.py.

Accepts the reference's root schema (``gpu.{default,<product>}``,
``cpu.hourly_per_core``, ``memory.hourly_per_gib``, ``overhead.fraction``,
``calculation.{use_requests,include_sidecars}`` — ``cost.yaml:1-28``) *and* the
different schema its planner expects (``gpus.<name>.hourly``/``cpu_per_hour``/...,
``planner.py:29-43``), so one file serves both tools.  Adds per-partition price
fractions for MI355X compute partitions / SR-IOV VFs (the reference prices a MIG
slice as zero GPUs: ``cost_estimator.py:226-235``).
"""

from __future__ import annotations

import re
from dataclasses import dataclass, field
from typing import Any, Dict, Optional

DEFAULT_GPU_HOURLY = 3.45


@dataclass
class UnitPricing:
    gpu_default: float = DEFAULT_GPU_HOURLY
    gpu_map: Dict[str, float] = field(default_factory=dict)
    cpu_hourly_per_core: float = 0.04
    mem_hourly_per_gib: float = 0.005
    storage_hourly_per_gib: float = 0.00014
    overhead_fraction: float = 0.10
    use_requests: bool = True
    include_sidecars: bool = False
    partition_fraction: Dict[str, float] = field(default_factory=lambda: {"spx": 1.0, "dpx": 0.5, "qpx": 0.25, "cpx": 0.125, "vf": 0.125})
    regions: Dict[str, float] = field(default_factory=dict)

    def gpu_hourly(self, product_label: Optional[str]) -> float:
        """Best-effort label match: exact, then normalised-substring either way."""
        if not product_label:
            return self.gpu_default
        if product_label in self.gpu_map:
            return self.gpu_map[product_label]
        norm = lambda s: "-".join(t for t in re.split(r"[^A-Za-z0-9]+", s) if t).lower()  # noqa: E731
        key = norm(product_label)
        best = None
        for k, v in self.gpu_map.items():
            nk = norm(k)
            if key in nk or nk in key:
                if best is None or len(nk) > best[0]:
                    best = (len(nk), v)
        if best:
            return best[1]
        # model-number match ("MI355X" inside either string)
        m = re.search(r"mi\d{3}[a-z]?", key)
        if m:
            for k, v in self.gpu_map.items():
                if m.group(0) in norm(k):
                    return v
        return self.gpu_default


def pricing_from_dict(data: Dict[str, Any]) -> UnitPricing:
    data = data or {}
    p = UnitPricing()
    gpu = data.get("gpu") or {}
    if isinstance(gpu, dict):
        p.gpu_default = float(gpu.get("default", p.gpu_default))
        p.gpu_map = {str(k): float(v) for k, v in gpu.items() if k != "default" and isinstance(v, (int, float))}
    # planner-style schema
    for name, spec in (data.get("gpus") or {}).items():
        if isinstance(spec, dict) and "hourly" in spec:
            p.gpu_map[str(name)] = float(spec["hourly"])
        elif isinstance(spec, (int, float)):
            p.gpu_map[str(name)] = float(spec)
    p.cpu_hourly_per_core = float((data.get("cpu") or {}).get("hourly_per_core", data.get("cpu_per_hour", p.cpu_hourly_per_core)))
    p.mem_hourly_per_gib = float((data.get("memory") or {}).get("hourly_per_gib", data.get("memory_per_gb_hour", p.mem_hourly_per_gib)))
    p.storage_hourly_per_gib = float((data.get("storage") or {}).get("hourly_per_gib", data.get("storage_per_gb_hour", p.storage_hourly_per_gib)))
    p.overhead_fraction = float((data.get("overhead") or {}).get("fraction", p.overhead_fraction))
    calc = data.get("calculation") or {}
    p.use_requests = bool(calc.get("use_requests", True))
    p.include_sidecars = bool(calc.get("include_sidecars", False))
    if isinstance(data.get("partition_fraction"), dict):
        p.partition_fraction.update({str(k).lower(): float(v) for k, v in data["partition_fraction"].items()})
    if isinstance(data.get("regions"), dict):
        p.regions = {str(k): float(v) for k, v in data["regions"].items()}
    return p


def load_pricing(path: Optional[str]) -> UnitPricing:
    if not path:
        return UnitPricing()
    import os

    if not os.path.exists(path):
        return UnitPricing()
    from kvmini.utils import load_yaml

    return pricing_from_dict(load_yaml(path) or {})

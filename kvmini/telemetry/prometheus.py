"""Prometheus client re-keyed to ROCm exporter metric names.

The reference queries NVIDIA DCGM names with two fallbacks each
(``analyze.py:250-309``, ``energy/collector.py:44-48``).  The equivalents on an
AMD cluster come from the AMD Device Metrics Exporter (``gpu_gfx_activity``,
``gpu_used_vram``, ``gpu_package_power`` / ``gpu_average_package_power``), with the
older ``amd_smi_*`` / ``rocm_smi_*`` exporter names as fallbacks.  Pod scoping
uses the same ``<isvc>-predictor-.*`` regex (``analyze.py:258``).
"""

from __future__ import annotations

import json
import urllib.parse
import urllib.request
from typing import Any, Dict, List, Optional

GPU_UTIL_METRICS = ["gpu_gfx_activity", "amd_smi_gfx_activity", "rocm_smi_utilization_gpu", "amd_gpu_use_percent"]
GPU_MEM_METRICS = ["gpu_used_vram", "amd_smi_vram_used", "rocm_smi_memory_vram_used", "amd_gpu_memory_use_bytes"]
GPU_POWER_METRICS = ["gpu_package_power", "gpu_average_package_power", "amd_smi_average_socket_power", "rocm_smi_power_average", "amd_gpu_power"]
# Device Metrics Exporter reports used VRAM in MiB; the smi exporters in bytes.
_MIB_METRICS = {"gpu_used_vram"}


def prom_query(
    prom_url: str,
    query: str,
    start: Optional[float] = None,
    end: Optional[float] = None,
    step: Optional[int] = None,
    timeout: float = 15.0,
) -> Dict[str, Any]:
    if start is not None and end is not None and step is not None:
        endpoint = "/api/v1/query_range"
        params = {"query": query, "start": str(start), "end": str(end), "step": str(step)}
    else:
        endpoint = "/api/v1/query"
        params = {"query": query}
    url = prom_url.rstrip("/") + endpoint + "?" + urllib.parse.urlencode(params)
    with urllib.request.urlopen(url, timeout=timeout) as resp:
        return json.loads(resp.read())


def prom_vector_avg(result: Dict[str, Any]) -> Optional[float]:
    """Mean over series of an instant vector — or, for a range (matrix)
    response, mean over every sample of every series."""
    try:
        res = result["data"]["result"]
    except (KeyError, TypeError):
        return None
    vals: List[float] = []
    for s in res or []:
        try:
            if "value" in s:
                vals.append(float(s["value"][1]))
            for _t, v in s.get("values", []) or []:
                vals.append(float(v))
        except (TypeError, ValueError, IndexError):
            continue
    vals = [v for v in vals if v == v]
    if not vals:
        return None
    return sum(vals) / len(vals)


def _selector(namespace: str, isvc: str) -> str:
    return f'{{namespace="{namespace}",pod=~"{isvc}-predictor-.*"}}'


def _first_avg(prom_url: str, metrics: List[str], sel: str, dur: str) -> Optional[float]:
    for m in metrics:
        try:
            v = prom_vector_avg(prom_query(prom_url, f"avg_over_time({m}{sel}[{dur}])"))
        except Exception:
            v = None
        if v is not None:
            return v * (1024.0 * 1024.0) if m in _MIB_METRICS else v
    return None


def utilization_from_prom(prom_url: str, namespace: str, isvc: str, start: float, end: float) -> Dict[str, Optional[float]]:
    """Window-averaged GPU activity %, VRAM bytes, package power W (ROCm exporter)
    and cAdvisor CPU cores / memory bytes for the predictor pods."""
    window = max(1, int(end - start))
    step = max(5, int((end - start) / 60))
    dur = f"{window}s"
    sel = _selector(namespace, isvc)
    csel = f'{{namespace="{namespace}",pod=~"{isvc}-predictor-.*",container!="",container!="POD"}}'

    def _safe(q: str) -> Optional[float]:
        try:
            return prom_vector_avg(prom_query(prom_url, q, start, end, step))
        except Exception:
            return None

    return {
        "gpu_util_avg": _first_avg(prom_url, GPU_UTIL_METRICS, sel, dur),
        "gpu_mem_used_avg": _first_avg(prom_url, GPU_MEM_METRICS, sel, dur),
        "gpu_power_watts_avg": _first_avg(prom_url, GPU_POWER_METRICS, sel, dur),
        "cpu_util_avg": _safe(f"avg(rate(container_cpu_usage_seconds_total{csel}[2m]))"),
        "mem_used_avg": _safe(f"avg(container_memory_working_set_bytes{csel})"),
    }


def power_instant(prom_url: str, namespace: str, isvc: str) -> Optional[float]:
    """Summed instantaneous package power (W) of the service's GPUs — the
    collector's per-tick query (reference ``energy/collector.py:44-48,226-242``)."""
    sel = _selector(namespace, isvc)
    for m in GPU_POWER_METRICS:
        try:
            res = prom_query(prom_url, f"sum({m}{sel})")["data"]["result"]
            if res:
                return float(res[0]["value"][1])
        except Exception:
            continue
    return None


def cache_hit_ratio_from_prom(prom_url: str, namespace: str, isvc: str, start: float, end: float) -> Optional[float]:
    """vLLM prefix-cache hit ratio (reference ``analyze.py:312-333``; adds the
    ``vllm:gpu_prefix_cache_*`` names current vLLM exports)."""
    dur = f"{max(1, int(end - start))}s"
    sel = _selector(namespace, isvc)
    queries = [
        f"sum(rate(vllm:gpu_prefix_cache_hits_total{sel}[{dur}])) / sum(rate(vllm:gpu_prefix_cache_queries_total{sel}[{dur}]))",
        f"sum(rate(vllm_prompt_cache_hits_total{sel}[{dur}])) / sum(rate(vllm_prompt_cache_requests_total{sel}[{dur}]))",
        f"sum(rate(vllm_cache_hits_total{sel}[{dur}])) / (sum(rate(vllm_cache_hits_total{sel}[{dur}])) + sum(rate(vllm_cache_misses_total{sel}[{dur}])))",
    ]
    for q in queries:
        try:
            v = prom_vector_avg(prom_query(prom_url, q))
        except Exception:
            v = None
        if v is not None and 0.0 <= v <= 1.0:
            return v
    return None

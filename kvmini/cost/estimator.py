"""Resource-seconds cost estimator (``kvmini cost`` / stage 3 of ``bench``).

Model (reference ``cost_estimator.py:343-501``): for every pod of the service,
active interval ∩ measurement window x container resources gives GPU-, CPU-core-
and GiB-seconds; total = Σ resource-seconds x hourly price / 3600 x (1 +
overhead); per-request and per-1k-token costs divide by successes / total tokens;
cold/warm costs allocate the total by request count.  Differences:

* pods come from a :class:`~kvmini.cluster.ClusterBackend` (kubectl, local process
  registry, fake) — an unreachable cluster yields zero resource-seconds and a
  warning instead of a crash (reference: uncaught ``check_output``, ``:426``);
* GPU resource key is ``amd.com/gpu``; MI355X compute-partition / SR-IOV resources
  (``amd.com/gpu-cpx`` ...) are charged ``partition_fraction`` of a GPU instead of
  zero (reference prices MIG slices as 0 GPUs, ``:226-235``);
* the node product label is ``amd.com/gpu.product-name`` (fallback
  ``amd.com/gpu.family``, ``amd.com/gpu.device-id``);
* ``calculation.use_requests`` / ``include_sidecars`` from cost.yaml are honoured
  (parsed but ignored by the reference, ``:385``), and no 0.2 GiB-second seed.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
from typing import Any, Dict, List, Optional, Tuple

from kvmini.analysis.coldwarm import is_success
from kvmini.analysis.reduce import read_requests_csv, window_bounds
from kvmini.cluster import ClusterBackend, get_backend, parse_k8s_time
from kvmini.utils import merge_json_file, parse_k8s_quantity

from .pricing import UnitPricing, load_pricing

GPU_KEY = "amd.com/gpu"
SIDECARS = ("queue-proxy", "istio-proxy", "agent")
PRODUCT_LABELS = ("amd.com/gpu.product-name", "amd.com/gpu.family", "amd.com/gpu.device-id", "beta.amd.com/gpu.product-name")


def gpu_units(resources: Dict[str, Any], pricing: UnitPricing) -> float:
    """GPU-equivalents requested by one resources map: whole GPUs (``amd.com/gpu``)
    plus partition resources (``amd.com/gpu-<mode>`` / ``amd.com/<mode>...``) weighted
    by their price fraction."""
    total = 0.0
    for key, val in (resources or {}).items():
        if not key.startswith("amd.com/"):
            continue
        name = key[len("amd.com/") :].lower()
        try:
            n = float(val)
        except (TypeError, ValueError):
            continue
        if name == "gpu":
            total += n
            continue
        for mode, frac in pricing.partition_fraction.items():
            if mode in name.replace("gpu-", "").split("-")[0] or name.endswith(mode) or f"-{mode}" in name:
                total += n * frac
                break
    return total


def container_resources(container: Dict[str, Any], pricing: UnitPricing) -> Tuple[float, float, float]:
    """(cpu cores, memory GiB, GPU-equivalents) of one container spec."""
    rs = container.get("resources") or {}
    which = rs.get("requests" if pricing.use_requests else "limits") or {}
    cpu = parse_k8s_quantity(which.get("cpu"))
    mem = parse_k8s_quantity(which.get("memory")) / float(2**30)
    gpus = gpu_units(which, pricing) or gpu_units(rs.get("limits") or {}, pricing)  # GPUs usually live in limits only
    return cpu, mem, gpus


def container_start_end(pod_status: Dict[str, Any]) -> Tuple[Optional[float], Optional[float]]:
    start = end = None
    for s in pod_status.get("containerStatuses") or []:
        st = s.get("state") or {}
        for key in ("running", "terminated"):
            t = (st.get(key) or {}).get("startedAt")
            if t:
                ts = parse_k8s_time(t)
                start = ts if start is None else min(start, ts)
        fin = (st.get("terminated") or {}).get("finishedAt")
        if fin:
            te = parse_k8s_time(fin)
            end = te if end is None else max(end, te)
        if (st.get("running") or {}).get("startedAt"):
            end = None if fin is None else end  # a running container keeps the pod open
    # any running container means the pod has not ended
    if any(((s.get("state") or {}).get("running") or {}).get("startedAt") for s in pod_status.get("containerStatuses") or []):
        end = None
    return start, end


def sum_resource_seconds(pods: List[Dict[str, Any]], w0: float, w1: float, pricing: UnitPricing) -> Dict[str, float]:
    cpu_s = mem_s = gpu_s = 0.0
    for p in pods:
        spec, status = p.get("spec") or {}, p.get("status") or {}
        pst, pend = container_start_end(status)
        if pst is None and status.get("startTime"):
            pst = parse_k8s_time(status["startTime"])
        if pend is None:
            pend = w1
        if pst is None:
            continue
        start, end = max(pst, w0), min(pend, w1)
        if end <= start:
            continue
        secs = end - start
        for c in spec.get("containers") or []:
            if not pricing.include_sidecars and c.get("name") in SIDECARS:
                continue
            cpu, mem, gpus = container_resources(c, pricing)
            cpu_s += cpu * secs
            mem_s += mem * secs
            gpu_s += gpus * secs
    return {"cpu_core_seconds": cpu_s, "mem_gib_seconds": mem_s, "gpu_seconds": gpu_s}


def node_gpu_label(backend: ClusterBackend, pod: Dict[str, Any]) -> Optional[str]:
    node_name = (pod.get("spec") or {}).get("nodeName")
    if not node_name:
        return None
    try:
        labels = ((backend.get_node(node_name) or {}).get("metadata") or {}).get("labels") or {}
    except Exception:
        return None
    for key in PRODUCT_LABELS:
        if labels.get(key):
            return labels[key]
    return None


def calculate_cold_warm_costs(rows: List[Dict[str, Any]], total_cost: float, success: int) -> Dict[str, Any]:
    def tok(rs: List[Dict[str, Any]]) -> float:
        return sum(float(r["total_tokens"]) for r in rs if isinstance(r.get("total_tokens"), (int, float)))

    def cold_flag(r: Dict[str, Any]) -> bool:
        v = r.get("is_cold_start", False)
        return v if isinstance(v, bool) else str(v).strip().lower() == "true"

    ok = [r for r in rows if is_success(r)]
    cold = [r for r in ok if cold_flag(r)]
    warm = [r for r in ok if not cold_flag(r)]
    cold_total = total_cost * (len(cold) / success) if success else 0.0
    warm_total = total_cost * (len(warm) / success) if success else 0.0
    ct, wt = tok(cold), tok(warm)
    return {
        "cold_cost_per_request": cold_total / len(cold) if cold else None,
        "warm_cost_per_request": warm_total / len(warm) if warm else None,
        "cold_cost_per_1k_tokens": cold_total / ct * 1000.0 if ct > 0 else None,
        "warm_cost_per_1k_tokens": warm_total / wt * 1000.0 if wt > 0 else None,
        "cold_total_cost": cold_total,
        "warm_total_cost": warm_total,
        "cold_requests": len(cold),
        "warm_requests": len(warm),
        "cold_tokens": ct,
        "warm_tokens": wt,
    }


def estimate_cost(run_dir: str, namespace: str, service: str, cost_file: Optional[str] = "cost.yaml", backend: Optional[ClusterBackend] = None, quiet: bool = False) -> Dict[str, Any]:
    req_csv = os.path.join(run_dir, "requests_classified.csv")
    if not os.path.exists(req_csv):
        req_csv = os.path.join(run_dir, "requests.csv")
    if not os.path.exists(req_csv):
        raise FileNotFoundError(f"{req_csv} not found")
    rows = read_requests_csv(req_csv)
    if not rows:
        raise ValueError("no request rows")
    w0, w1 = window_bounds(rows)
    ok = [r for r in rows if is_success(r)]
    success = len(ok)
    total_tokens = sum(float(r["total_tokens"]) for r in ok if isinstance(r.get("total_tokens"), (int, float)))
    pricing = load_pricing(cost_file)
    backend = backend or get_backend()
    warnings: List[str] = []
    try:
        pods = backend.get_pods(namespace, service)
    except Exception as e:
        pods = []
        warnings.append(f"cluster backend {backend.name} failed: {e}")
    if not pods:
        warnings.append(f"no pods found for {namespace}/{service} via backend '{backend.name}': resource-seconds are zero")
    rsecs = sum_resource_seconds(pods, w0, w1, pricing)
    resource_source = "declared requests" if pricing.use_requests else "declared limits"
    usage_path = os.path.join(run_dir, "resource_usage.json")
    if os.path.exists(usage_path):
        # measured around the load test by the benchmark (cluster backends that can read the
        # processes): CPU-seconds really consumed, memory charged at the peak resident set
        try:
            with open(usage_path) as f:
                use = json.load(f)
            cpu_s, peak = float(use["cpu_core_seconds"]), float(use["peak_rss_bytes"])
            if cpu_s >= 0 and peak > 0:
                scale = (w1 - w0) / float(use["window_s"]) if float(use.get("window_s") or 0) > 0 else 1.0
                rsecs["cpu_core_seconds"] = cpu_s * min(scale, 1.0)  # (the snapshots bracket the request window)
                rsecs["mem_gib_seconds"] = peak / float(2**30) * (w1 - w0)
                resource_source = "measured: " + str(use.get("source", "process accounting"))
        except (OSError, ValueError, KeyError, TypeError) as e:
            warnings.append(f"resource_usage.json ignored: {e}")
    product = node_gpu_label(backend, pods[0]) if pods else None
    gpu_hr = pricing.gpu_hourly(product)
    total = (rsecs["gpu_seconds"] * gpu_hr + rsecs["cpu_core_seconds"] * pricing.cpu_hourly_per_core + rsecs["mem_gib_seconds"] * pricing.mem_hourly_per_gib) / 3600.0
    total *= 1.0 + pricing.overhead_fraction
    have_cost = bool(pods) and total > 0
    fields: Dict[str, Any] = {
        "cost_per_request": total / success if success and have_cost else None,
        "cost_per_1k_tokens": total / total_tokens * 1000.0 if total_tokens > 0 and have_cost else None,
        **calculate_cold_warm_costs(rows, total, success),
        "cost_breakdown": {
            "total_cost": total,
            **rsecs,
            "gpu_hourly": gpu_hr,
            "cpu_hourly_per_core": pricing.cpu_hourly_per_core,
            "mem_hourly_per_gib": pricing.mem_hourly_per_gib,
            "overhead_fraction": pricing.overhead_fraction,
            "gpu_product_label": product,
            "use_requests": pricing.use_requests,
            "include_sidecars": pricing.include_sidecars,
            "cluster_backend": backend.name,
            "cpu_memory_source": resource_source,
        },
    }
    if warnings:
        fields["cost_warnings"] = warnings
    merge_json_file(os.path.join(run_dir, "results.json"), fields)
    cw = fields
    summary = {
        k: cw.get(k)
        for k in ("cost_per_request", "cost_per_1k_tokens", "cold_cost_per_request", "warm_cost_per_request", "cold_cost_per_1k_tokens", "warm_cost_per_1k_tokens")
    }
    c, w = cw.get("cold_cost_per_request"), cw.get("warm_cost_per_request")
    summary["cold_vs_warm_multiplier"] = c / w if c and w else None
    if not quiet:
        for msg in warnings:
            print(f"WARNING: {msg}", file=sys.stderr)
        print(json.dumps(summary, indent=2))
    return fields


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(prog="kvmini cost", description="Estimate $ per request / per 1k tokens from pod resource-seconds")
    ap.add_argument("--run-dir", required=True)
    ap.add_argument("--namespace", default="default")
    ap.add_argument("--service", default="demo-llm")
    ap.add_argument("--cost-file", default="cost.yaml")
    ap.add_argument("--cluster-backend", default=None, choices=["auto", "kubectl", "local", "fake"])
    a = ap.parse_args(argv)
    try:
        estimate_cost(a.run_dir, a.namespace, a.service, a.cost_file, get_backend(a.cluster_backend))
    except (FileNotFoundError, ValueError) as e:
        print(f"ERROR: {e}", file=sys.stderr)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())

"""requests.csv -> results.json (``kvmini analyze``).

Key set and formulas follow the reference analyzer (``analyze.py:463-618``):
``p50/p95/p99_ms`` over successful latencies, ``throughput_rps`` = successes /
(max(start+latency) - min(start)), ``tokens_per_sec`` = completion tokens /
window, ``time_to_first_token_ms`` = mean TTFT, cold/warm split, token-timing
histograms, window, request counts, IO-probe merge, cache-hit ratio; the file is
merge-updated so later stages (cost, energy, quality) add to it.

Differences, all additive:

* utilisation/power come from the local amd-smi sampler (``power.json``) when no
  Prometheus URL is given — no DCGM, no exporter needed on a single node;
* per-token (inter-token-latency) statistics from ``token_events.npy`` — reduced
  by the HIP histogram/quantile kernel when the stream is large;
* alias keys for the two other results.json dialects the reference's own tools
  read (``runners/backends/tgi/invoke.sh:198-213``, ``scripts/compare_backends.py:313-339``).
"""

from __future__ import annotations

import argparse
import csv
import json
import math
import os
import statistics as stats
import sys
from typing import Any, Dict, List, Optional, Tuple

from kvmini.cluster import ClusterBackend, get_backend
from kvmini.utils import load_json, merge_json_file

from .coldwarm import classify_requests_cold_warm, compute_cold_warm_metrics, is_success
from .histograms import compute_histogram
from .quantiles import percentiles

NUMERIC_FIELDS = (
    "scheduled_ms",
    "start_ms",
    "ttfb_ms",
    "ttft_ms",
    "tllt_ms",
    "latency_ms",
    "status",
    "prompt_tokens",
    "completion_tokens",
    "total_tokens",
)

REQUESTS_CSV_HEADER = [
    "id",
    "scheduled_ms",
    "start_ms",
    "ttfb_ms",
    "tllt_ms",
    "latency_ms",
    "status",
    "prompt_tokens",
    "completion_tokens",
    "total_tokens",
    "error",
    "trace_id",
]


def read_requests_csv(path: str) -> List[Dict[str, Any]]:
    """Load requests.csv, coercing the known numeric columns to float; empty /
    'NaN' cells stay as the original string (treated as missing downstream)."""
    rows: List[Dict[str, Any]] = []
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            for k in NUMERIC_FIELDS:
                if k in row and row[k] not in (None, "", "NaN"):
                    try:
                        row[k] = float(row[k])
                    except ValueError:
                        pass
            rows.append(row)
    return rows


def _num(row: Dict[str, Any], key: str) -> float:
    v = row.get(key)
    if isinstance(v, (int, float)):
        return float(v)
    return float("nan")


def window_bounds(rows: List[Dict[str, Any]]) -> Tuple[float, float]:
    start = min((_num(r, "start_ms") if not math.isnan(_num(r, "start_ms")) else 0.0) for r in rows) / 1000.0
    end = max(((0.0 if math.isnan(_num(r, "start_ms")) else _num(r, "start_ms")) + (0.0 if math.isnan(_num(r, "latency_ms")) else _num(r, "latency_ms"))) for r in rows) / 1000.0
    return start, end


def compute_token_timing_analysis(rows: List[Dict[str, Any]]) -> Dict[str, Any]:
    """TTFB / TLLT / generation-time / per-token histograms and quantiles over
    successful requests (reference ``analyze.py:125-180``)."""
    ok = [r for r in rows if is_success(r)]
    ttfb = [_num(r, "ttfb_ms") for r in ok]
    tllt = [_num(r, "tllt_ms") for r in ok]
    ctoks = [(_num(r, "completion_tokens") if not math.isnan(_num(r, "completion_tokens")) else 0.0) for r in ok]

    gen: List[float] = []
    per_tok: List[float] = []
    for a, b, n in zip(ttfb, tllt, ctoks):
        if not math.isnan(a) and not math.isnan(b) and n > 0:
            gen.append(b - a)
            per_tok.append((b - a) / n)

    def q(arr: List[float], ps: Tuple[float, ...]) -> List[Optional[float]]:
        if not arr:
            return [None] * len(ps)
        return [None if math.isnan(v) else v for v in percentiles(arr, ps)]

    t50, t95, t99 = q(ttfb, (0.5, 0.95, 0.99))
    l50, l95, l99 = q(tllt, (0.5, 0.95, 0.99))
    g50, g95 = q(gen, (0.5, 0.95))
    k50, k95 = q(per_tok, (0.5, 0.95))
    return {
        "ttfb_histogram": compute_histogram(ttfb),
        "tllt_histogram": compute_histogram(tllt),
        "generation_time_histogram": compute_histogram(gen),
        "per_token_time_histogram": compute_histogram(per_tok),
        "ttfb_p50_ms": t50,
        "ttfb_p95_ms": t95,
        "ttfb_p99_ms": t99,
        "tllt_p50_ms": l50,
        "tllt_p95_ms": l95,
        "tllt_p99_ms": l99,
        "generation_p50_ms": g50,
        "generation_p95_ms": g95,
        "per_token_p50_ms": k50,
        "per_token_p95_ms": k95,
        "avg_tokens_per_response": stats.mean(ctoks) if ctoks else None,
    }


def inter_token_latency_metrics(run_dir: str, engine: str = "auto") -> Dict[str, Any]:
    """p50/p95/p99 + 20-bin histogram of *inter-token gaps* from the load
    generator's per-token event stream (``token_events.npy``: rows of
    ``[request_id, token_index, t_ms_since_request_start]``).  The reference only
    has the aggregate ``(tllt - ttfb) / completion_tokens`` (``analyze.py:143-146``)."""
    path = os.path.join(run_dir, "token_events.npy")
    if not os.path.exists(path):
        return {}
    import numpy as np

    ev = np.load(path)
    if ev.ndim != 2 or ev.shape[0] < 2:
        return {}
    # events are written grouped by request, in token order
    same_req = ev[1:, 0] == ev[:-1, 0]
    gaps = (ev[1:, 2] - ev[:-1, 2])[same_req]
    if gaps.size == 0:
        return {}
    p50, p95, p99 = percentiles(gaps, (0.5, 0.95, 0.99), engine=engine)
    return {
        "itl_p50_ms": p50,
        "itl_p95_ms": p95,
        "itl_p99_ms": p99,
        "itl_mean_ms": float(gaps.mean()),
        "itl_samples": int(gaps.size),
        "itl_histogram": compute_histogram(gaps, engine=engine),
    }


def utilization_from_power_json(run_dir: str, start: float, end: float) -> Dict[str, Optional[float]]:
    """Window averages from the local amd-smi sampler output (``power.json``)."""
    path = os.path.join(run_dir, "power.json")
    if not os.path.exists(path):
        return {}
    try:
        data = load_json(path)
    except (OSError, ValueError):
        return {}
    samples = data.get("samples", data) if isinstance(data, dict) else data
    watts, util, mem = [], [], []
    for s in samples or []:
        ts = s.get("ts_s", s.get("ts"))
        if ts is None or not (start - 1.0 <= float(ts) <= end + 1.0):
            continue
        if s.get("watts") is not None:
            watts.append(float(s["watts"]))
        if s.get("gfx_util") is not None:
            util.append(float(s["gfx_util"]))
        if s.get("vram_used") is not None:
            mem.append(float(s["vram_used"]))
    out: Dict[str, Optional[float]] = {}
    if watts:
        out["gpu_power_watts_avg"] = sum(watts) / len(watts)
    if util:
        out["gpu_util_avg"] = sum(util) / len(util)
    if mem:
        out["gpu_mem_used_avg"] = sum(mem) / len(mem)
    return out


def cache_hit_ratio(prom_url: Optional[str], backend: ClusterBackend, namespace: str, service: str, start: float, end: float) -> Optional[float]:
    """Prometheus first, then count 'cache hit' / 'cache miss' lines in the last
    200 log lines of the predictor pods (reference ``analyze.py:312-355``)."""
    if prom_url:
        from kvmini.telemetry.prometheus import cache_hit_ratio_from_prom

        v = cache_hit_ratio_from_prom(prom_url, namespace, service, start, end)
        if v is not None:
            return v
    try:
        text = backend.logs(namespace, service, tail=200)
    except Exception:
        return None
    hits = misses = 0
    for line in text.splitlines():
        ll = line.lower()
        if "cache hit" in ll:
            hits += 1
        if "cache miss" in ll:
            misses += 1
    return hits / (hits + misses) if hits + misses else None


def add_alias_keys(results: Dict[str, Any]) -> Dict[str, Any]:
    """Aliases for the adapter dialect and the README/comparator dialect, so a
    tool written against either finds real numbers instead of zeros."""
    r = results
    req = r.get("requests") or {}
    alias = {
        # adapter dialect
        "p50_total_ms": r.get("p50_ms"),
        "p95_total_ms": r.get("p95_ms"),
        "p99_total_ms": r.get("p99_ms"),
        "throughput_req_per_sec": r.get("throughput_rps"),
        "mean_ttfb_ms": r.get("time_to_first_token_ms"),
        "p95_ttfb_ms": r.get("ttft_p95_ms"),
        "gpu_utilization_avg": r.get("gpu_util_avg"),
        "total_requests": req.get("total"),
        "successful_requests": req.get("success"),
        # README / comparator dialect
        "p50_latency_ms": r.get("p50_ms"),
        "p95_latency_ms": r.get("p95_ms"),
        "p99_latency_ms": r.get("p99_ms"),
        "avg_ttft_ms": r.get("time_to_first_token_ms"),
        "energy_per_1k_tokens_wh": r.get("energy_wh_per_1k_tokens"),
        "avg_gpu_utilization_pct": r.get("gpu_util_avg"),
        "success_rate": (1.0 - r["error_rate"]) if isinstance(r.get("error_rate"), (int, float)) else None,
    }
    for k, v in alias.items():
        if v is not None or k not in r:
            r[k] = v
    return r


def analyze_rows(
    rows: List[Dict[str, Any]],
    cold_start_times: Optional[List[float]] = None,
    util: Optional[Dict[str, Optional[float]]] = None,
    io_probe: Optional[Dict[str, Any]] = None,
    cache_ratio: Optional[float] = None,
    cold_window_sec: float = 30.0,
) -> Dict[str, Any]:
    """Pure reduction (no IO): the function the golden-vector tests pin."""
    start, end = window_bounds(rows)
    cold_start_times = list(cold_start_times or [])
    classify_requests_cold_warm(rows, cold_start_times, cold_window_sec)

    ok_rows = [r for r in rows if is_success(r)]
    lats = [_num(r, "latency_ms") for r in ok_rows]
    # the true first-content-token time when the load generator recorded it,
    # else first byte (the reference's definition, loadtest.py:325-329)
    ttfs = [v for v in (_num(r, "ttfb_ms") for r in ok_rows) if not math.isnan(v)]
    ttft_tok = [v for v in (_num(r, "ttft_ms") for r in ok_rows) if not math.isnan(v)]

    total = len(rows)
    success = len(ok_rows)
    error_rate = (total - success) / total if total else None
    duration = end - start if end > start else None
    throughput = success / duration if duration else None
    ctoks = [(0.0 if math.isnan(_num(r, "completion_tokens")) else _num(r, "completion_tokens")) for r in ok_rows]
    total_tokens = sum((0.0 if math.isnan(_num(r, "total_tokens")) else _num(r, "total_tokens")) for r in ok_rows)
    tokens_per_sec = sum(ctoks) / duration if duration else None

    util_full: Dict[str, Optional[float]] = {
        "gpu_util_avg": None,
        "gpu_mem_used_avg": None,
        "gpu_power_watts_avg": None,
        "cpu_util_avg": None,
        "mem_used_avg": None,
    }
    util_full.update(util or {})

    energy: Dict[str, Any] = {}
    if util_full.get("gpu_power_watts_avg") and duration:
        wh = float(util_full["gpu_power_watts_avg"]) * duration / 3600.0
        energy = {
            "energy_wh": wh,
            "energy_wh_per_request": wh / success if success else None,
            "energy_wh_per_1k_tokens": wh / total_tokens * 1000.0 if total_tokens > 0 else None,
        }

    p50, p95, p99 = percentiles(lats, (0.5, 0.95, 0.99)) if lats else (float("nan"),) * 3
    t50 = t95 = None
    if ttfs:
        t50, t95 = percentiles(ttfs, (0.5, 0.95))
    io_probe = io_probe or {}
    results: Dict[str, Any] = {
        "p50_ms": p50,
        "p95_ms": p95,
        "p99_ms": p99,
        "throughput_rps": throughput,
        "tokens_per_sec": tokens_per_sec,
        "error_rate": error_rate,
        "cold_start_count": len(cold_start_times),
        "time_to_first_token_ms": stats.mean(ttfs) if ttfs else None,
        "ttft_p50_ms": t50,
        "ttft_p95_ms": t95,
        **util_full,
        **energy,
        **compute_cold_warm_metrics(rows),
        **compute_token_timing_analysis(rows),
        "window": {"start": start, "end": end, "seconds": duration},
        "requests": {"total": total, "success": success},
        "network_rtt_p50_ms": (io_probe.get("rtt") or {}).get("p50_ms"),
        "network_rtt_p95_ms": (io_probe.get("rtt") or {}).get("p95_ms"),
        "s3_avg_MBps": (io_probe.get("s3") or {}).get("avg_MBps"),
        "cache_hit_ratio": cache_ratio,
    }
    if ttft_tok:
        c50, c95 = percentiles(ttft_tok, (0.5, 0.95))
        results.update(
            {
                "first_content_token_ms": stats.mean(ttft_tok),
                "first_content_token_p50_ms": c50,
                "first_content_token_p95_ms": c95,
            }
        )
    return results


def analyze_run(
    run_dir: str,
    namespace: str = "default",
    service: str = "demo-llm",
    prom_url: Optional[str] = None,
    backend: Optional[ClusterBackend] = None,
    quantile_engine: str = "auto",
    quiet: bool = False,
) -> Dict[str, Any]:
    req_csv = os.path.join(run_dir, "requests.csv")
    rows = read_requests_csv(req_csv)
    if not rows:
        raise SystemExit("No request rows found")
    backend = backend or get_backend()
    start, end = window_bounds(rows)

    try:
        cold_times = backend.cold_start_times(namespace, service, start, end)
    except Exception:
        cold_times = []

    util: Dict[str, Optional[float]] = {}
    if prom_url:
        try:
            from kvmini.telemetry.prometheus import utilization_from_prom

            util = utilization_from_prom(prom_url, namespace, service, start, end)
        except Exception as e:  # Prometheus is optional
            print(f"Prometheus query failed: {e}", file=sys.stderr)
    local_util = utilization_from_power_json(run_dir, start, end)
    for k, v in local_util.items():
        if util.get(k) is None:
            util[k] = v

    io_probe = load_json(os.path.join(run_dir, "io_probe.json"), default={}) if os.path.exists(os.path.join(run_dir, "io_probe.json")) else {}
    try:
        chr_val = cache_hit_ratio(prom_url, backend, namespace, service, start, end)
    except Exception:
        chr_val = None

    results = analyze_rows(rows, cold_times, util, io_probe, chr_val)
    results.update(inter_token_latency_metrics(run_dir, engine=quantile_engine))
    # structured-output runs: how many answers satisfy the schema they were asked for (written by the load generator)
    structured_path = os.path.join(run_dir, "structured.json")
    if os.path.exists(structured_path):
        results["structured_output"] = load_json(structured_path, default={})

    # classified CSV (adds is_cold_start)
    fieldnames = list(rows[0].keys())
    with open(os.path.join(run_dir, "requests_classified.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=fieldnames, extrasaction="ignore")
        w.writeheader()
        w.writerows(rows)

    merged = merge_json_file(os.path.join(run_dir, "results.json"), results)
    merged = merge_json_file(os.path.join(run_dir, "results.json"), add_alias_keys(merged))
    if not quiet:
        print(json.dumps(load_json(os.path.join(run_dir, "results.json")), indent=2))
    return merged


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(prog="kvmini analyze", description="Reduce a run directory into results.json")
    ap.add_argument("--run-dir", required=True)
    ap.add_argument("--namespace", default="default")
    ap.add_argument("--service", default="demo-llm")
    ap.add_argument("--prom-url", default=None)
    ap.add_argument("--cluster-backend", default=None, choices=["auto", "kubectl", "local", "fake"])
    ap.add_argument("--quantile-engine", default="auto", choices=["auto", "python", "numpy", "hip"])
    ap.add_argument("--quiet", action="store_true")
    args = ap.parse_args(argv)
    if not os.path.exists(os.path.join(args.run_dir, "requests.csv")):
        print(f"requests.csv not found in {args.run_dir}", file=sys.stderr)
        return 1
    analyze_run(
        args.run_dir,
        args.namespace,
        args.service,
        args.prom_url,
        backend=get_backend(args.cluster_backend),
        quantile_engine=args.quantile_engine,
        quiet=args.quiet,
    )
    return 0


if __name__ == "__main__":
    sys.exit(main())

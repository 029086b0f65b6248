"""Grafana dashboards as code (``python -m kvmini.report.dashboards --out dashboards/``).

Four dashboards, same topics as the reference's ``dashboards/*.json`` (request
latency, utilisation, cold-start impact, Knative/KServe golden signals) but keyed
to what an AMD cluster exports: the ROCm device-metrics exporter
(``gpu_gfx_activity``, ``gpu_used_vram``, ``gpu_package_power`` …) instead of
DCGM, plus the vLLM-style engine metrics that both vLLM-ROCm and the native
server publish (``vllm:time_to_first_token_seconds``, ``vllm:num_requests_running``,
``vllm:gpu_prefix_cache_*``).  Generating them from one table keeps the metric
names in sync with ``kvmini.telemetry.prometheus``.
"""

from __future__ import annotations

import argparse
import json
import os
from typing import Any, Dict, List, Optional, Sequence, Tuple

POD = 'namespace="$namespace",pod=~"$isvc-predictor-.*"'
ISTIO = 'reporter="destination", destination_workload_namespace="$namespace", destination_workload=~"$isvc-predictor-.*"'
CTR = POD + ',container!="",container!="POD"'
VLLM = 'namespace="$namespace",pod=~"$isvc-predictor-.*"'


def _templating() -> Dict[str, Any]:
    def var(name: str, query: str, current: str) -> Dict[str, Any]:
        return {"name": name, "type": "query", "datasource": {"type": "prometheus", "uid": "${DS_PROMETHEUS}"}, "query": query, "refresh": 2, "current": {"text": current, "value": current}, "includeAll": False}

    return {"list": [var("namespace", "label_values(kube_pod_info, namespace)", "ml-prod"), var("isvc", 'label_values(kube_pod_info{namespace="$namespace", pod=~".*-predictor-.*"}, pod)', "demo-llm")]}


def panel(pid: int, title: str, exprs: Sequence[Tuple[str, str]], unit: str = "short", x: int = 0, y: int = 0, w: int = 12, h: int = 8, kind: str = "timeseries", description: str = "") -> Dict[str, Any]:
    return {
        "id": pid,
        "type": kind,
        "title": title,
        "description": description,
        "datasource": {"type": "prometheus", "uid": "${DS_PROMETHEUS}"},
        "gridPos": {"x": x, "y": y, "w": w, "h": h},
        "fieldConfig": {"defaults": {"unit": unit}, "overrides": []},
        "targets": [{"refId": chr(65 + i), "expr": e, "legendFormat": leg} for i, (e, leg) in enumerate(exprs)],
    }


def dashboard(uid: str, title: str, panels: List[Dict[str, Any]], tags: Sequence[str]) -> Dict[str, Any]:
    return {
        "uid": uid,
        "title": title,
        "tags": list(tags),
        "schemaVersion": 39,
        "version": 1,
        "editable": True,
        "refresh": "10s",
        "time": {"from": "now-30m", "to": "now"},
        "__inputs": [{"name": "DS_PROMETHEUS", "label": "Prometheus", "type": "datasource", "pluginId": "prometheus"}],
        "templating": _templating(),
        "panels": panels,
    }


def _hq(q: float, metric: str, sel: str, window: str = "2m") -> str:
    return f"histogram_quantile({q}, sum(rate({metric}{{{sel}}}[{window}])) by (le))"


def latency() -> Dict[str, Any]:
    p = [
        panel(1, "Istio p50/p95 latency (ms)", [(_hq(0.50, "istio_request_duration_milliseconds_bucket", ISTIO), "p50"), (_hq(0.95, "istio_request_duration_milliseconds_bucket", ISTIO), "p95")], "ms", 0, 0),
        panel(2, "Requests per second", [(f'sum(rate(istio_requests_total{{{ISTIO}, response_code="200"}}[1m]))', "200"), (f"sum(rate(istio_requests_total{{{ISTIO}}}[1m]))", "all")], "reqps", 12, 0),
        panel(3, "Time to first token (s)", [(_hq(0.50, "vllm:time_to_first_token_seconds_bucket", VLLM), "p50"), (_hq(0.95, "vllm:time_to_first_token_seconds_bucket", VLLM), "p95")], "s", 0, 8),
        panel(4, "Time per output token (s)", [(_hq(0.50, "vllm:time_per_output_token_seconds_bucket", VLLM), "p50"), (_hq(0.95, "vllm:time_per_output_token_seconds_bucket", VLLM), "p95")], "s", 12, 8),
        panel(5, "End-to-end request latency (s)", [(_hq(0.95, "vllm:e2e_request_latency_seconds_bucket", VLLM), "p95"), (_hq(0.99, "vllm:e2e_request_latency_seconds_bucket", VLLM), "p99")], "s", 0, 16),
        panel(6, "Running / waiting requests", [(f"sum(vllm:num_requests_running{{{VLLM}}})", "running"), (f"sum(vllm:num_requests_waiting{{{VLLM}}})", "waiting")], "short", 12, 16),
    ]
    return dashboard("kserve-llm-latency", "KServe LLM on ROCm - Request Latency", p, ["kserve", "llm", "rocm", "latency"])


def utilization() -> Dict[str, Any]:
    p = [
        panel(1, "GPU utilisation (%)", [(f"avg(gpu_gfx_activity{{{POD}}})", "gfx"), (f"avg(gpu_umc_activity{{{POD}}})", "memory controller")], "percent", 0, 0, description="ROCm device-metrics exporter: gpu_gfx_activity / gpu_umc_activity"),
        panel(2, "GPU memory (GiB)", [(f"avg(gpu_used_vram{{{POD}}}) / 1024", "used"), (f"avg(gpu_total_vram{{{POD}}}) / 1024", "total")], "gbytes", 12, 0, description="exporter reports VRAM in MiB"),
        panel(3, "GPU package power (W)", [(f"sum(gpu_package_power{{{POD}}})", "package"), (f"sum(gpu_average_package_power{{{POD}}})", "average")], "watt", 0, 8),
        panel(4, "GPU temperature / clock", [(f"max(gpu_junction_temperature{{{POD}}})", "junction °C"), (f"avg(gpu_clock{{{POD}}})", "gfx MHz")], "short", 12, 8),
        panel(5, "CPU (cores)", [(f"avg(rate(container_cpu_usage_seconds_total{{{CTR}}}[2m]))", "cores")], "short", 0, 16),
        panel(6, "Memory (GiB)", [(f"avg(container_memory_working_set_bytes{{{CTR}}}) / 1073741824", "working set")], "gbytes", 12, 16),
        panel(7, "KV-cache usage / prefix-cache hit rate", [(f"avg(vllm:gpu_cache_usage_perc{{{VLLM}}})", "kv usage"), (f"sum(rate(vllm:gpu_prefix_cache_hits_total{{{VLLM}}}[2m])) / sum(rate(vllm:gpu_prefix_cache_queries_total{{{VLLM}}}[2m]))", "prefix hit rate")], "percentunit", 0, 24),
        panel(8, "Energy (Wh, integral of package power)", [(f"sum(sum_over_time(gpu_package_power{{{POD}}}[$__range])) * $__interval_ms / 3600000", "Wh")], "watth", 12, 24, kind="stat"),
    ]
    return dashboard("kserve-llm-utilization", "KServe LLM on ROCm - Utilization", p, ["kserve", "llm", "rocm", "utilization"])


def cold_start() -> Dict[str, Any]:
    p = [
        panel(1, "Ready predictor pods", [(f'sum(kube_pod_status_ready{{namespace="$namespace",pod=~"$isvc-predictor-.*",condition="true"}})', "ready")], "short", 0, 0),
        panel(2, "Pod starts (scale-from-zero events)", [(f'sum(changes(kube_pod_start_time{{namespace="$namespace",pod=~"$isvc-predictor-.*"}}[5m]))', "starts / 5m")], "short", 12, 0),
        panel(3, "p95 latency around cold starts (ms)", [(_hq(0.95, "istio_request_duration_milliseconds_bucket", ISTIO, "1m"), "p95 1m"), (_hq(0.95, "istio_request_duration_milliseconds_bucket", ISTIO, "10m"), "p95 10m")], "ms", 0, 8, 24),
        panel(4, "Activator queue depth (requests held while scaling)", [(f'sum(activator_request_concurrency{{namespace_name="$namespace", configuration_name=~"$isvc-predictor.*"}})', "held")], "short", 0, 16),
        panel(5, "Container start → ready (s)", [(f'max(kube_pod_status_ready_time{{namespace="$namespace",pod=~"$isvc-predictor-.*"}} - kube_pod_start_time{{namespace="$namespace",pod=~"$isvc-predictor-.*"}})', "startup")], "s", 12, 16),
        panel(6, "GPU memory during model load (GiB)", [(f"max(gpu_used_vram{{{POD}}}) / 1024", "used")], "gbytes", 0, 24, 24),
    ]
    return dashboard("cold-start-impact", "KServe LLM on ROCm - Cold-Start Impact", p, ["kserve", "knative", "cold-start"])


def golden() -> Dict[str, Any]:
    kn = 'namespace_name="$namespace", configuration_name=~"$isvc-predictor.*"'
    p = [
        panel(1, "Traffic (RPS by response code)", [(f"sum(rate(revision_app_request_count{{{kn}}}[1m])) by (response_code_class)", "{{response_code_class}}")], "reqps", 0, 0),
        panel(2, "Errors (5xx ratio)", [(f'sum(rate(revision_app_request_count{{{kn}, response_code_class="5xx"}}[2m])) / sum(rate(revision_app_request_count{{{kn}}}[2m]))', "5xx ratio")], "percentunit", 12, 0),
        panel(3, "Latency (queue-proxy p50/p95/p99, ms)", [(_hq(q, "revision_app_request_latencies_bucket", kn), f"p{int(q * 100)}") for q in (0.5, 0.95, 0.99)], "ms", 0, 8),
        panel(4, "Saturation: concurrency vs target", [(f"sum(autoscaler_stable_request_concurrency{{{kn}}})", "stable concurrency"), (f"sum(autoscaler_target_concurrency_per_pod{{{kn}}})", "target per pod"), (f"sum(autoscaler_panic_request_concurrency{{{kn}}})", "panic concurrency")], "short", 12, 8),
        panel(5, "Pods: desired / actual / not ready", [(f"sum(autoscaler_desired_pods{{{kn}}})", "desired"), (f"sum(autoscaler_actual_pods{{{kn}}})", "actual"), (f"sum(autoscaler_not_ready_pods{{{kn}}})", "not ready")], "short", 0, 16),
        panel(6, "Panic mode", [(f"max(autoscaler_panic_mode{{{kn}}})", "panic")], "short", 12, 16, kind="stat"),
        panel(7, "GPU saturation (gfx activity %, per pod)", [(f"avg(gpu_gfx_activity{{{POD}}}) by (pod)", "{{pod}}")], "percent", 0, 24, 24),
    ]
    return dashboard("knative-kserve-golden", "Knative / KServe Golden Signals (ROCm)", p, ["kserve", "knative", "golden-signals"])


ALL = {"kserve-llm-latency.json": latency, "kserve-llm-utilization.json": utilization, "cold-start-impact.json": cold_start, "knative-kserve-golden.json": golden}


def write_all(out_dir: str) -> List[str]:
    os.makedirs(out_dir, exist_ok=True)
    paths = []
    for name, fn in ALL.items():
        path = os.path.join(out_dir, name)
        with open(path, "w") as f:
            json.dump(fn(), f, indent=2)
            f.write("\n")
        paths.append(path)
    return paths


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(description="Generate the Grafana dashboards")
    ap.add_argument("--out", default="dashboards")
    a = ap.parse_args(argv)
    for p in write_all(a.out):
        print(p)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())

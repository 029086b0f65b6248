"""Backend comparator (``kvmini compare-backends``).

Benchmarks one model on several serving runtimes with the same profile and
reports one ``BackendResult`` per runtime — latency quantiles, TTFT, RPS,
tokens/s, $/1K tokens, Wh/1K tokens, GPU utilisation, deploy time, cold-start
count — into ``backend_comparison.{json,html,csv}`` with per-metric winners
(reference ``scripts/compare_backends.py:20-92,370-395,518-531``).

Runtimes here are the ROCm ones: ``vllm`` (vLLM-ROCm ServingRuntime), ``tgi``
(TGI-ROCm) and ``native`` (the in-tree MI355X engine) — the reference's third
backend is TensorRT-LLM, which has no AMD build.  ``--endpoints name=url``
compares endpoints that are already up.
"""

from __future__ import annotations

import argparse
import csv
import html
import os
import sys
import time
from dataclasses import asdict, dataclass, fields
from typing import Any, Dict, List, Optional

from kvmini.cluster import get_backend
from kvmini.config.profiles import load_profile, resolve_load_settings
from kvmini.deploy import manifests as mf
from kvmini.report.generator import CSS, format_number, svg_bar_chart
from kvmini.utils import load_json, write_json

LOWER_BETTER = {"avg_latency_ms", "p95_latency_ms", "avg_ttft_ms", "cost_per_1k_tokens", "deployment_time_s", "energy_per_1k_tokens_wh"}
HIGHER_BETTER = {"throughput_rps", "tokens_per_sec", "avg_gpu_util_pct"}
RUNTIMES = {"vllm": "kserve-vllm-rocm", "tgi": "kserve-tgi-rocm", "native": "kvmini-native"}
PROTOCOL = {"vllm": "openai", "native": "openai", "tgi": "tgi"}


@dataclass
class BackendResult:
    backend: str
    runtime: str
    model: str
    profile: str
    success: bool = False
    error: Optional[str] = None
    total_requests: int = 0
    successful_requests: int = 0
    failed_requests: int = 0
    avg_latency_ms: float = 0.0
    p50_latency_ms: float = 0.0
    p95_latency_ms: float = 0.0
    p99_latency_ms: float = 0.0
    avg_ttft_ms: float = 0.0
    throughput_rps: float = 0.0
    tokens_per_sec: float = 0.0
    cost_per_1k_tokens: float = 0.0
    cost_per_request: float = 0.0
    energy_per_1k_tokens_wh: float = 0.0
    avg_gpu_util_pct: float = 0.0
    avg_gpu_memory_util_pct: float = 0.0
    peak_memory_gb: float = 0.0
    deployment_time_s: float = 0.0
    cold_start_count: int = 0

    def absorb(self, res: Dict[str, Any]) -> None:
        """Fill the metric fields from a results.json document."""

        def g(*keys: str, default: float = 0.0) -> float:
            for k in keys:
                v = res.get(k)
                if isinstance(v, (int, float)) and v == v:
                    return float(v)
            return default

        req = res.get("requests") if isinstance(res.get("requests"), dict) else {}
        self.total_requests = int(req.get("total") or g("total_requests"))
        self.successful_requests = int(req.get("success") or g("successful_requests", default=round(self.total_requests * (1.0 - g("error_rate")))))
        self.failed_requests = self.total_requests - self.successful_requests
        self.avg_latency_ms = g("mean_latency_ms", "avg_latency_ms", "p50_ms")
        self.p50_latency_ms = g("p50_ms")
        self.p95_latency_ms = g("p95_ms")
        self.p99_latency_ms = g("p99_ms")
        self.avg_ttft_ms = g("first_content_token_ms", "time_to_first_token_ms", "avg_ttft_ms", "ttft_p50_ms")
        self.throughput_rps = g("throughput_rps")
        self.tokens_per_sec = g("tokens_per_sec")
        self.cost_per_1k_tokens = g("cost_per_1k_tokens")
        self.cost_per_request = g("cost_per_request")
        self.energy_per_1k_tokens_wh = g("energy_wh_per_1k_tokens", "energy_per_1k_tokens_wh")
        self.avg_gpu_util_pct = g("gpu_util_avg")
        self.avg_gpu_memory_util_pct = g("gpu_mem_util_avg")
        self.peak_memory_gb = g("gpu_mem_used_max_mb", "gpu_mem_used_avg") / 1024.0
        self.cold_start_count = int(g("cold_start_count", "cold_requests"))


@dataclass
class ComparisonSummary:
    model: str
    profile: str
    timestamp: str
    results: List[BackendResult]

    def get_winner(self, metric: str) -> Optional[str]:
        ok = [r for r in self.results if r.success]
        if not ok:
            return None
        if metric in LOWER_BETTER:
            return min(ok, key=lambda r: getattr(r, metric, float("inf"))).backend
        if metric in HIGHER_BETTER:
            return max(ok, key=lambda r: getattr(r, metric, 0)).backend
        return None

    def to_dict(self) -> Dict[str, Any]:
        d = asdict(self)
        d["winners"] = {m: self.get_winner(m) for m in sorted(LOWER_BETTER | HIGHER_BETTER)}
        return d


class BackendComparator:
    def __init__(self, namespace: str = "ml-prod", cost_file: str = "cost.yaml", cluster_backend: Optional[str] = None, endpoints: Optional[Dict[str, str]] = None, ready_timeout: float = 600.0, keep: bool = False):
        self.namespace = namespace
        self.cost_file = cost_file
        self.cluster_backend = cluster_backend
        self.cluster = get_backend(cluster_backend)
        self.endpoints = endpoints or {}
        self.ready_timeout = ready_timeout
        self.keep = keep
        self.backends = dict(RUNTIMES)

    def manifest(self, backend: str, model: str, service: str) -> Dict[str, Any]:
        is_uri = "://" in model
        if backend == "tgi":
            return mf.tgi_rocm_isvc(service, self.namespace, model.split("://", 1)[-1] if is_uri else model)
        if backend == "native":
            return mf.native_isvc(service, self.namespace, model)
        return mf.vllm_rocm_isvc(service, self.namespace, model_uri=model if is_uri else None, model_name=None if is_uri else model)

    def run_single_backend(self, backend: str, runtime: str, model: str, profile_name: str, service: str, output_dir: str) -> BackendResult:
        from kvmini.harness import bench

        prof_stem = os.path.splitext(os.path.basename(profile_name))[0]
        r = BackendResult(backend=backend, runtime=runtime, model=model, profile=prof_stem)
        deployed = False
        try:
            prof = load_profile(profile_name)
            s = resolve_load_settings({"requests": None, "concurrency": None, "max_tokens": None, "pattern": None, "stream": None}, prof)
            t0 = time.time()
            if backend in self.endpoints:
                url = self.endpoints[backend]
            else:
                self.cluster.apply(self.manifest(backend, model, service))
                deployed = True
                self.cluster.wait_ready(self.namespace, service, self.ready_timeout)
                url = self.cluster.get_isvc_url(self.namespace, service)
                if not url:
                    raise RuntimeError("service has no URL")
            r.deployment_time_s = round(time.time() - t0, 3)
            run_dir = os.path.join(output_dir, f"run_{backend}")
            argv = ["--namespace", self.namespace, "--service", service, "--url", url, "--model", model.rsplit("/", 1)[-1], "--profile", profile_name, "--run-dir", run_dir, "--cost-file", self.cost_file, "--protocol", PROTOCOL.get(backend, "openai")]
            if self.cluster_backend:
                argv += ["--cluster-backend", self.cluster_backend]
            rc = bench.main(argv)
            res_path = os.path.join(run_dir, "results.json")
            if rc != 0 or not os.path.exists(res_path):
                raise RuntimeError(f"benchmark failed (exit {rc})")
            r.absorb(load_json(res_path))
            if r.successful_requests == 0:
                raise RuntimeError("no successful requests")
            r.success = True
        except (Exception, SystemExit) as e:
            r.error = str(e) or type(e).__name__
        finally:
            if deployed and not self.keep:
                try:
                    self.cluster.delete_isvc(self.namespace, service)
                except Exception:
                    pass
        return r

    # -- reports ------------------------------------------------------------------------
    def generate_comparison_report(self, summary: ComparisonSummary, output_dir: str) -> Dict[str, str]:
        os.makedirs(output_dir, exist_ok=True)
        paths = {k: os.path.join(output_dir, f"backend_comparison.{k}") for k in ("json", "html", "csv")}
        write_json(paths["json"], summary.to_dict())
        with open(paths["html"], "w") as f:
            f.write(self.render_html(summary))
        with open(paths["csv"], "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=[x.name for x in fields(BackendResult)])
            w.writeheader()
            for r in summary.results:
                w.writerow(asdict(r))
        return paths

    @staticmethod
    def render_html(summary: ComparisonSummary) -> str:
        ok = [r for r in summary.results if r.success]
        names = [r.backend for r in ok]
        out = [f"<!DOCTYPE html><html><head><meta charset='utf-8'><title>Backend comparison</title><style>{CSS} tr.failed td{{color:#c92a2a}} td.win{{font-weight:700;color:#2b8a3e}}</style></head><body>"]
        out.append(f"<h1>Backend comparison</h1><p>Model <b>{html.escape(summary.model)}</b> · profile <b>{html.escape(summary.profile)}</b> · {html.escape(summary.timestamp)}</p>")
        if ok:
            out.append(svg_bar_chart(names, {"p50 ms": [r.p50_latency_ms for r in ok], "p95 ms": [r.p95_latency_ms for r in ok], "TTFT ms": [r.avg_ttft_ms for r in ok]}, "Latency"))
            out.append(svg_bar_chart(names, {"RPS": [r.throughput_rps for r in ok]}, "Throughput"))
            out.append(svg_bar_chart(names, {"$ / 1K tokens": [r.cost_per_1k_tokens for r in ok]}, "Cost"))
        metrics = [("p95_latency_ms", "P95 ms"), ("avg_ttft_ms", "TTFT ms"), ("throughput_rps", "RPS"), ("tokens_per_sec", "tok/s"), ("cost_per_1k_tokens", "$/1K tok"), ("energy_per_1k_tokens_wh", "Wh/1K tok"), ("avg_gpu_util_pct", "GPU util %"), ("deployment_time_s", "deploy s")]
        out.append("<table><tr><th>backend</th><th>runtime</th>" + "".join(f"<th>{lab}</th>" for _, lab in metrics) + "<th>status</th></tr>")
        for r in summary.results:
            cells = "".join(f"<td class='{'win' if r.success and summary.get_winner(m) == r.backend else ''}'>{format_number(getattr(r, m)) if r.success else '–'}</td>" for m, _ in metrics)
            out.append(f"<tr class='{'' if r.success else 'failed'}'><td>{html.escape(r.backend)}</td><td>{html.escape(r.runtime)}</td>{cells}<td>{'ok' if r.success else html.escape(r.error or 'failed')}</td></tr>")
        out.append("</table></body></html>")
        return "\n".join(out)


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(prog="kvmini compare-backends", description="Compare KServe inference runtime backends on AMD GPUs")
    ap.add_argument("--model", required=True, help="model URI or name to benchmark across backends")
    ap.add_argument("--profile", default="runners/profiles/standard.yaml")
    ap.add_argument("--backends", nargs="+", default=["vllm", "tgi"])
    ap.add_argument("--namespace", default="ml-prod")
    ap.add_argument("--output-dir", default="backend-comparison")
    ap.add_argument("--cost-file", default="cost.yaml")
    ap.add_argument("--endpoints", nargs="*", default=None, metavar="NAME=URL")
    ap.add_argument("--cluster-backend", default=None, choices=["auto", "kubectl", "local", "fake"])
    ap.add_argument("--ready-timeout", type=float, default=600.0)
    ap.add_argument("--keep", action="store_true", help="do not delete the deployed services")
    a = ap.parse_args(argv)
    endpoints = dict(e.split("=", 1) for e in (a.endpoints or []))
    os.makedirs(a.output_dir, exist_ok=True)
    comp = BackendComparator(a.namespace, a.cost_file, a.cluster_backend, endpoints, a.ready_timeout, a.keep)
    results: List[BackendResult] = []
    print(f"Starting backend comparison for model: {a.model}\nProfile: {a.profile}\nBackends: {a.backends}")
    for b in a.backends:
        if b not in comp.backends and b not in endpoints:
            print(f"Warning: unknown backend {b}, skipping")
            continue
        r = comp.run_single_backend(b, comp.backends.get(b, "external"), a.model, a.profile, f"compare-{b}-{int(time.time())}", a.output_dir)
        results.append(r)
        print(f"  {b}: P95={r.p95_latency_ms:.1f}ms, RPS={r.throughput_rps:.1f}, Cost=${r.cost_per_1k_tokens:.4f}/1K" if r.success else f"  {b}: FAILED: {r.error}")
    summary = ComparisonSummary(a.model, os.path.splitext(os.path.basename(a.profile))[0], time.strftime("%Y-%m-%d %H:%M:%S UTC", time.gmtime()), results)
    paths = comp.generate_comparison_report(summary, a.output_dir)
    print("Reports generated:\n" + "\n".join(f"  {k.upper()}: {v}" for k, v in paths.items()))
    ok = [r for r in results if r.success]
    if ok:
        print(f"\nSuccessful benchmarks: {len(ok)}/{len(results)}")
        print(f"Best P95 latency: {summary.get_winner('p95_latency_ms')}\nBest throughput: {summary.get_winner('throughput_rps')}\nBest cost efficiency: {summary.get_winner('cost_per_1k_tokens')}")
    return 0 if ok or not results else 1


if __name__ == "__main__":
    sys.exit(main())

"""Backend A/B comparison (``kvmini ab-compare``).

Deploys the same model on each serving backend (``vllm-rocm``, ``tgi-rocm``,
``native`` — the in-tree MI355X engine — or any already-running endpoint given
with ``--endpoints name=url``), runs the identical traffic profile against each,
optionally once streaming and once non-streaming, and writes
``ab_comparison.csv`` (``backend,streaming,requests,concurrency,mean_ttft_ms,
p95_ttft_ms,mean_tpot_ms,p95_tpot_ms,throughput_req_s,gpu_util_avg,
cost_per_1k_tokens``) plus ``comparison_report.json`` with the fastest-TTFT /
highest-throughput / lowest-cost winners and the streaming deltas (reference
``runners/ab-compare.sh:140-395``).
"""

from __future__ import annotations

import argparse
import csv
import os
import statistics
import sys
import time
from typing import Any, Dict, List, Optional

from kvmini.cluster import get_backend
from kvmini.config.profiles import load_profile, resolve_load_settings
from kvmini.deploy import manifests as mf
from kvmini.utils import load_yaml, load_json, write_json

CSV_COLS = ["backend", "streaming", "requests", "concurrency", "mean_ttft_ms", "p95_ttft_ms", "mean_tpot_ms", "p95_tpot_ms", "throughput_req_s", "gpu_util_avg", "cost_per_1k_tokens"]
PROTOCOL = {"vllm-rocm": "openai", "vllm": "openai", "native": "openai", "mock": "openai", "tgi-rocm": "tgi", "tgi": "tgi"}


def summarize_test(test_dir: str, backend: str, streaming: bool, requests: int, concurrency: int) -> Optional[Dict[str, Any]]:
    req_csv = os.path.join(test_dir, "requests.csv")
    if not os.path.exists(req_csv):
        return None
    ttft: List[float] = []
    tpot: List[float] = []
    lat: List[float] = []

    def num(row: Dict[str, str], key: str) -> Optional[float]:
        try:
            v = float(row.get(key) or "")
        except ValueError:
            return None
        return None if v != v else v

    with open(req_csv) as f:
        for row in csv.DictReader(f):
            if row.get("status") != "200":
                continue
            if num(row, "latency_ms") is not None:
                lat.append(num(row, "latency_ms"))
            if streaming:  # a non-streamed response has no first token: its "first byte" is the whole answer
                v = num(row, "ttft_ms")
                v = num(row, "ttfb_ms") if v is None else v
                if v is not None:
                    ttft.append(v)
            # time per output token: the column when a producer wrote it (the reference's
            # ab-compare.sh:209 expects one nobody writes), else decode time / decoded tokens
            t = num(row, "tpot_ms")
            if t is None:
                first = num(row, "ttft_ms")
                first = num(row, "ttfb_ms") if first is None else first
                last, n = num(row, "tllt_ms"), num(row, "completion_tokens")
                if streaming and first is not None and last is not None and n and n > 1:
                    t = (last - first) / (n - 1)
                elif not streaming and n and num(row, "latency_ms") is not None:
                    t = num(row, "latency_ms") / n  # whole-request time per token (includes prefill)
            if t is not None:
                tpot.append(t)

    def p95(xs: List[float]) -> Optional[float]:
        return sorted(xs)[min(len(xs) - 1, int(0.95 * len(xs)))] if xs else None

    def r(v: Optional[float], nd: int) -> Optional[float]:
        return None if v is None else round(float(v), nd)

    res_path = os.path.join(test_dir, "results.json")
    summary = load_json(res_path) if os.path.exists(res_path) else {}
    # a metric nobody measured stays empty (None): a 0.0 placeholder would win every "lowest" ranking
    return {
        "backend": backend,
        "streaming": "true" if streaming else "false",
        "requests": requests,
        "concurrency": concurrency,
        "mean_ttft_ms": r(statistics.mean(ttft), 1) if ttft else None,
        "p95_ttft_ms": r(p95(ttft), 1),
        "mean_tpot_ms": r(statistics.mean(tpot), 2) if tpot else None,
        "p95_tpot_ms": r(p95(tpot), 2),
        "throughput_req_s": r(summary.get("throughput_rps") or summary.get("throughput_req_per_sec"), 2),
        "gpu_util_avg": r(summary.get("gpu_util_avg") if summary.get("gpu_util_avg") is not None else summary.get("gpu_utilization_avg"), 1),
        "cost_per_1k_tokens": r(summary.get("cost_per_1k_tokens"), 6),
        # JSON / HTML report only; the CSV keeps the reference's columns
        "mean_latency_ms": r(statistics.mean(lat), 1) if lat else None,
        "p95_latency_ms": r(summary.get("p95_ms"), 1),
        "energy_wh_per_1k_tokens": r(summary.get("energy_wh_per_1k_tokens"), 5),
        "gpu_power_watts_avg": r(summary.get("gpu_power_watts_avg"), 1),
    }


def build_report(rows: List[Dict[str, Any]], meta: Dict[str, Any]) -> Dict[str, Any]:
    def f(r: Dict[str, Any], k: str) -> Optional[float]:
        v = r.get(k)
        return None if v in (None, "") else float(v)

    def best(k: str, pick: Any, unit_key: str) -> Optional[Dict[str, Any]]:
        cands = [r for r in rows if f(r, k) is not None]
        if not cands:
            return None
        w = pick(cands, key=lambda r: f(r, k))
        return {"backend": w["backend"], "streaming": w["streaming"], unit_key: f(w, k)}

    report: Dict[str, Any] = {
        "comparison_metadata": meta,
        "results": rows,
        "winners": {
            "fastest_ttft": best("mean_ttft_ms", min, "value_ms"),
            "fastest_tpot": best("mean_tpot_ms", min, "value_ms"),
            "highest_throughput": best("throughput_req_s", max, "value_req_s"),
            "lowest_cost": best("cost_per_1k_tokens", min, "value_per_1k"),
            "lowest_energy": best("energy_wh_per_1k_tokens", min, "value_wh_per_1k"),
        },
    }
    by: Dict[str, Dict[str, Dict[str, Any]]] = {}
    for r in rows:
        by.setdefault(r["backend"], {})[str(r["streaming"])] = r
    deltas = {}
    for b, modes in by.items():
        if "true" in modes and "false" in modes:
            # what streaming buys: the first token arrives this much earlier than the complete
            # non-streamed answer (mean request latency), at this throughput difference
            lat_ns, ttft_s = f(modes["false"], "mean_latency_ms"), f(modes["true"], "mean_ttft_ms")
            tp_s, tp_ns = f(modes["true"], "throughput_req_s"), f(modes["false"], "throughput_req_s")
            deltas[b] = {
                "ttft_advantage_ms": None if lat_ns is None or ttft_s is None else lat_ns - ttft_s,
                "throughput_advantage_req_s": None if tp_s is None or tp_ns is None else tp_s - tp_ns,
            }
    if deltas:
        report["streaming_vs_non_streaming"] = deltas
    return report


def write_html_report(path: str, report: Dict[str, Any]) -> str:
    """Cost + energy + latency report of an A/B run: one bar chart per metric (backends on the
    x axis, streaming / non-streaming as series), the result table and the winners."""
    import html

    from kvmini.report.generator import format_number, svg_bar_chart

    rows = report["results"]
    backends = list(dict.fromkeys(r["backend"] for r in rows))
    modes = list(dict.fromkeys(str(r["streaming"]) for r in rows))

    def series(key: str) -> Dict[str, List[Optional[float]]]:
        by = {(r["backend"], str(r["streaming"])): r.get(key) for r in rows}
        return {("streaming" if m == "true" else "non-streaming"): [None if by.get((b, m)) in (None, "") else float(by[(b, m)]) for b in backends] for m in modes}

    charts = [
        ("mean_ttft_ms", "Time to first token (mean)", " ms"), ("mean_tpot_ms", "Time per output token (mean)", " ms"), ("p95_latency_ms", "Request latency p95", " ms"),
        ("throughput_req_s", "Throughput", " req/s"), ("cost_per_1k_tokens", "Cost per 1K tokens", " $"), ("energy_wh_per_1k_tokens", "Energy per 1K tokens", " Wh"),
    ]
    meta = report["comparison_metadata"]
    out = ["<!doctype html><html><head><meta charset='utf-8'><title>Backend A/B comparison</title><style>body{font-family:system-ui,sans-serif;margin:2em;max-width:84em}"
           "table{border-collapse:collapse;margin:1em 0}td,th{border:1px solid #ced4da;padding:.3em .6em;text-align:right}td:first-child,th:first-child{text-align:left}"
           ".grid{display:grid;grid-template-columns:repeat(auto-fill,minmax(640px,1fr));gap:1em}.win{color:#2b8a3e;font-weight:700}</style></head><body>"]
    out.append(f"<h1>Backend A/B comparison</h1><p>Model <b>{html.escape(str(meta.get('model')))}</b>, profile <b>{html.escape(str(meta.get('profile')))}</b>, {meta.get('requests')} requests at concurrency {meta.get('concurrency')}"
               + (f", partition profile <b>{html.escape(str(meta.get('partition_profile')))}</b>" if meta.get("partition_profile") else "") + f" &middot; {html.escape(str(meta.get('timestamp')))}</p>")
    w = report.get("winners") or {}
    names = {"fastest_ttft": "Fastest first token", "fastest_tpot": "Fastest per token", "highest_throughput": "Highest throughput", "lowest_cost": "Lowest cost", "lowest_energy": "Lowest energy"}
    out.append("<ul>" + "".join(f"<li>{names.get(k, k)}: " + ("not measured" if not v else f"<span class='win'>{html.escape(v['backend'])}</span> (streaming: {v['streaming']}) &mdash; {format_number(next(x for kk, x in v.items() if kk.startswith('value')), 6)}") + "</li>" for k, v in w.items()) + "</ul>")
    out.append("<div class='grid'>" + "".join(svg_bar_chart(backends, series(k), title, unit) for k, title, unit in charts if any(v is not None for s_ in series(k).values() for v in s_)) + "</div>")
    cols = ["backend", "streaming", "mean_ttft_ms", "p95_ttft_ms", "mean_tpot_ms", "p95_tpot_ms", "mean_latency_ms", "p95_latency_ms", "throughput_req_s", "gpu_util_avg", "gpu_power_watts_avg", "cost_per_1k_tokens", "energy_wh_per_1k_tokens"]
    out.append("<table><tr>" + "".join(f"<th>{c}</th>" for c in cols) + "</tr>" + "".join("<tr>" + "".join(f"<td>{'' if r.get(c) is None else html.escape(str(r.get(c)))}</td>" for c in cols) + "</tr>" for r in rows) + "</table>")
    out.append("<p>Empty cells were not measured (for example: no GPU power for an endpoint that runs on another machine; no first token for a non-streamed response).</p></body></html>")
    with open(path, "w") as f:
        f.write("".join(out))
    return path


class ABRunner:
    def __init__(self, a: argparse.Namespace, settings: Dict[str, Any]):
        self.a = a
        self.s = settings
        self.cluster = get_backend(a.cluster_backend)
        self.endpoints = dict(e.split("=", 1) for e in (a.endpoints or []))
        self.partition = load_yaml(a.partition_profile) if getattr(a, "partition_profile", None) else None
        self.rows: List[Dict[str, Any]] = []

    def service(self, backend: str) -> str:
        return f"{self.a.model}-{backend}".replace("_", "-").replace(".", "-").lower()

    def deploy(self, backend: str) -> str:
        if backend in self.endpoints:
            return self.endpoints[backend]
        if backend not in mf.BUILDERS:
            raise RuntimeError(f"no deployer for backend '{backend}' (known: {sorted(mf.BUILDERS)}; or pass --endpoints {backend}=URL)")
        name = self.service(backend)
        if backend in ("tgi-rocm", "tgi"):
            manifest = mf.tgi_rocm_isvc(name, self.a.namespace, self.a.model)
        elif backend == "native":
            manifest = mf.native_isvc(name, self.a.namespace, self.a.model)
        else:
            manifest = mf.vllm_rocm_isvc(name, self.a.namespace, model_name=self.a.model, model_uri=self.a.model_uri)
        if self.partition is not None:
            manifest = mf.apply_partition_profile(manifest, self.partition)
        self.cluster.apply(manifest)
        self.cluster.wait_ready(self.a.namespace, name, self.a.ready_timeout)
        url = self.cluster.get_isvc_url(self.a.namespace, name)
        if not url:
            raise RuntimeError(f"could not get URL for {name}")
        return url

    def cleanup(self, backend: str) -> None:
        if self.a.no_cleanup or backend in self.endpoints:
            return
        try:
            self.cluster.delete_isvc(self.a.namespace, self.service(backend))
        except Exception as e:
            print(f"  cleanup warning: {e}", file=sys.stderr)

    def test(self, backend: str, url: str, streaming: bool) -> None:
        from kvmini.harness import bench

        tag = f"{backend}_{'streaming' if streaming else 'nonstreaming'}"
        test_dir = os.path.join(self.a.run_dir, tag)
        print(f"Testing {backend} (streaming: {str(streaming).lower()}) -> {test_dir}")
        argv = ["--namespace", self.a.namespace, "--service", self.service(backend), "--url", url, "--model", self.a.model, "--requests", str(self.s["requests"]), "--concurrency", str(self.s["concurrency"]), "--max-tokens", str(self.s["max_tokens"]), "--pattern", self.s["pattern"], "--run-dir", test_dir, "--protocol", PROTOCOL.get(backend, "openai"), "--cost-file", self.a.cost_file]
        if backend == "mock" or self.a.no_energy:
            argv.append("--no-energy")
        if not streaming:
            argv.append("--no-stream")
        if self.a.cluster_backend:
            argv += ["--cluster-backend", self.a.cluster_backend]
        try:
            bench.main(argv)
        except SystemExit:
            pass
        row = summarize_test(test_dir, backend, streaming, self.s["requests"], self.s["concurrency"])
        if row is None:
            print(f"WARNING: no results for {backend} test", file=sys.stderr)
            return
        self.rows.append(row)
        self.write_csv()

    def write_csv(self) -> str:
        path = os.path.join(self.a.run_dir, "ab_comparison.csv")
        with open(path, "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=CSV_COLS, extrasaction="ignore")
            w.writeheader()
            w.writerows({k: ("" if v is None else v) for k, v in r.items()} for r in self.rows)
        return path

    def run(self) -> Optional[Dict[str, Any]]:
        backends = [b.strip() for b in self.a.backends.split(",") if b.strip()]
        for i, b in enumerate(backends):
            print(f"=== Testing backend: {b} ===")
            try:
                url = self.deploy(b)
            except Exception as e:
                print(f"ERROR: {b}: {e}", file=sys.stderr)
                continue
            try:
                self.test(b, url, True)
                if self.a.toggle_streaming:
                    if self.a.wait_between > 0:
                        time.sleep(self.a.wait_between)
                    self.test(b, url, False)
            finally:
                self.cleanup(b)
            if i + 1 < len(backends) and self.a.wait_between > 0:
                time.sleep(self.a.wait_between)
        if not self.rows:
            print("No results to analyze", file=sys.stderr)
            return None
        meta = {"timestamp": time.strftime("%Y-%m-%dT%H:%M:%S", time.gmtime()), "profile": self.a.profile, "model": self.a.model, "requests": self.s["requests"], "concurrency": self.s["concurrency"], "streaming_tested": bool(self.a.toggle_streaming),
                "partition_profile": getattr(self.a, "partition_profile", None)}
        report = build_report(self.rows, meta)
        write_json(os.path.join(self.a.run_dir, "comparison_report.json"), report)
        if not getattr(self.a, "no_report", False):
            print("Report:", write_html_report(os.path.join(self.a.run_dir, "ab_report.html"), report))
        return report


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(prog="kvmini ab-compare", description="Backend A/B testing for fair runtime comparisons")
    ap.add_argument("--backends", required=True, help="comma-separated: vllm-rocm,tgi-rocm,native (or names given in --endpoints)")
    ap.add_argument("--model", required=True)
    ap.add_argument("--model-uri", default=None)
    ap.add_argument("--profile", default="standard")
    ap.add_argument("--namespace", default="ml-prod")
    ap.add_argument("--toggle-streaming", action="store_true")
    ap.add_argument("--requests", type=int, default=None)
    ap.add_argument("--concurrency", type=int, default=None)
    ap.add_argument("--run-dir", default=None)
    ap.add_argument("--no-cleanup", action="store_true")
    ap.add_argument("--wait-between", type=float, default=60.0)
    ap.add_argument("--ready-timeout", type=float, default=600.0)
    ap.add_argument("--endpoints", nargs="*", default=None, metavar="NAME=URL", help="already-running endpoints to compare (no deploy / cleanup)")
    ap.add_argument("--cluster-backend", default=None, choices=["auto", "kubectl", "local", "fake"])
    ap.add_argument("--cost-file", default="cost.yaml")
    ap.add_argument("--no-energy", action="store_true", help="do not sample amd-smi power during the tests")
    ap.add_argument("--partition-profile", default=None, help="MI355X partition profile YAML applied to every deployed InferenceService (profiles/partition/*.yaml)")
    ap.add_argument("--no-report", action="store_true", help="skip ab_report.html")
    a = ap.parse_args(argv)
    try:
        profile = load_profile(a.profile)
    except (FileNotFoundError, ValueError) as e:
        print(f"ERROR: {e}", file=sys.stderr)
        return 1
    s = resolve_load_settings({"requests": a.requests, "concurrency": a.concurrency, "max_tokens": None, "pattern": None, "stream": None}, profile)
    a.run_dir = a.run_dir or os.path.join("runs", "ab_compare_" + time.strftime("%Y-%m-%d_%H-%M-%S"))
    os.makedirs(a.run_dir, exist_ok=True)
    print(f"=== Backend A/B comparison ===\nBackends: {a.backends}\nModel: {a.model}\nProfile: {a.profile} (pattern {s['pattern']}, {s['requests']} requests, concurrency {s['concurrency']}, max tokens {s['max_tokens']})")
    report = ABRunner(a, s).run()
    if report is None:
        return 1
    w = report["winners"]
    print("\n=== A/B comparison results ===")

    def line(title: str, key: str, val_key: str, unit: str) -> None:
        x = w.get(key)
        print(f"  {title:<20}" + ("not measured" if not x else f"{x['backend']} (streaming: {x['streaming']}) - {x[val_key]} {unit}"))

    line("Fastest TTFT:", "fastest_ttft", "value_ms", "ms")
    line("Fastest per token:", "fastest_tpot", "value_ms", "ms/token")
    line("Highest throughput:", "highest_throughput", "value_req_s", "req/s")
    line("Lowest cost:", "lowest_cost", "value_per_1k", "$ per 1K tokens")
    for b, d in (report.get("streaming_vs_non_streaming") or {}).items():
        ta, tp = d["ttft_advantage_ms"], d["throughput_advantage_req_s"]
        print(f"  {b}: first token {'n/a' if ta is None else f'{ta:.1f} ms'} earlier than the complete non-streamed answer, throughput {'n/a' if tp is None else f'{tp:+.2f} req/s'}")
    print(f"Results directory: {a.run_dir}")
    return 0


if __name__ == "__main__":
    sys.exit(main())

"""Quantization & decoding sweep engine (``kvmini quant-sweep``).

For each entry of ``sweep_configurations`` (weight quantization, KV-cache dtype,
context length, memory utilization, TP degree, decoding settings) the runner
deploys a dedicated InferenceService ``<service>-<name>`` (vLLM-ROCm manifest on
``amd.com/gpu``; or a local native / mock server with ``--local``), runs the
benchmark pipeline, optionally adds the quality score, tears the service down,
and appends the merged row to ``sweep_results_intermediate.csv`` so an
interrupted sweep keeps what it measured.  Analysis: quantization × KV-dtype
heatmaps per metric, a multi-objective Pareto frontier (p95 ↓, $/1K tok ↓,
quality ↑, RPS ↑) with ``pareto_optimal_configs.csv``, and ``sweep_report.html``
(reference ``sweeps/quantization_sweep.py:40-170,321-550``).

Differences: run ids hash the config with sha1 (Python's ``hash`` is salted per
process, so the reference's ids are not reproducible); a point is dropped from
the frontier when it *is dominated* (the reference drops the dominating point);
plots are inline SVG so the sweep needs no plotting stack.
"""

from __future__ import annotations

import argparse
import csv
import hashlib
import json
import os
import subprocess
import sys
import time
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import yaml

from kvmini.cluster import get_backend
from kvmini.deploy import manifests as mf
from kvmini.harness.quality import AgreementEvaluator, ParetoAnalyzer, QualityEvaluator
from kvmini.report.generator import CSS, format_number, svg_heatmap, svg_scatter
from kvmini.utils import load_json, load_yaml, write_json

OBJECTIVES: List[Tuple[str, str]] = [("p95_ms", "minimize"), ("cost_per_1k_tokens", "minimize"), ("quality_score", "maximize"), ("throughput_rps", "maximize")]
# quantization modes the ROCm vLLM build accepts; fp8 / mxfp4 run natively on gfx950 matrix cores
ROCM_QUANT = {"none", "fp8", "awq", "gptq", "mxfp4", "compressed-tensors", "quark"}


def config_run_id(config: Dict[str, Any]) -> str:
    h = hashlib.sha1(json.dumps(config, sort_keys=True).encode()).hexdigest()[:12]
    return f"quant_sweep_{h}_{time.strftime('%H%M%S')}"


def loadtest_args_for(decoding: Dict[str, Any], output_dir: str, name: str) -> List[str]:
    lt: List[str] = []
    for key, flag in (("temperature", "--temperature"), ("top_p", "--top-p"), ("top_k", "--top-k"), ("num_completions", "--num-completions")):
        if key in decoding:
            lt += [flag, str(decoding[key])]
    if decoding.get("json_mode"):
        lt.append("--json-mode")
    extra = decoding.get("extra_openai")
    if extra:
        path = os.path.join(output_dir, f"extra_openai_{name}.json")
        with open(path, "w") as f:
            json.dump(extra, f)
        lt += ["--extra-openai-json", path]
    return lt


def vllm_args_for(config: Dict[str, Any]) -> List[str]:
    args: List[str] = []
    q = str(config.get("quantization", "none") or "none")
    if q != "none":
        args += ["--quantization", q]
    kv = str(config.get("kv_cache_dtype", "auto") or "auto")
    if kv != "auto":
        args += ["--kv-cache-dtype", kv]
    if config.get("enforce_eager"):
        args.append("--enforce-eager")
    return args


def pareto_mask(rows: Sequence[Dict[str, Any]], objectives: Sequence[Tuple[str, str]] = OBJECTIVES) -> List[bool]:
    """Non-dominated rows under the objectives present in at least one row.
    Missing values are neutral: they neither help nor hurt a comparison."""
    used = [(m, d) for m, d in objectives if any(r.get(m) is not None for r in rows)]
    if not rows or not used:
        return [True] * len(rows)
    n = len(rows)
    mat = np.full((n, len(used)), np.nan)
    for i, r in enumerate(rows):
        for j, (m, d) in enumerate(used):
            try:
                v = float(r.get(m))
            except (TypeError, ValueError):
                continue
            mat[i, j] = v if d == "minimize" else -v
    if n >= 64:
        from kvmini.analysis import native_host

        if native_host.available():
            return [bool(x) for x in native_host.pareto_mask(mat, nan_neutral=True)]
    a, b = mat[:, None, :], mat[None, :, :]  # is i dominated by j ?
    known = ~np.isnan(a) & ~np.isnan(b)
    le = np.where(known, b <= a, True).all(-1)
    lt = np.where(known, b < a, False).any(-1)
    dominated = (le & lt).any(1)
    return [not d for d in dominated]


class QuantizationSweepRunner:
    def __init__(self, config: Dict[str, Any], output_dir: str, local: Optional[str] = None, cluster_backend: Optional[str] = None, runs_root: str = "runs", port: int = 8000, cost_file: str = "cost.yaml",
                 url: Optional[str] = None, resume: bool = False):
        self.config = config
        self.url = url  # benchmark this endpoint for every configuration instead of the deployed service's own
        self.resume = resume
        self.output_dir = output_dir
        self.local = local
        self.backend = get_backend("local" if local else cluster_backend)
        self.cluster_backend = cluster_backend
        self.runs_root = runs_root
        self.port = port
        self.cost_file = cost_file
        self.results: List[Dict[str, Any]] = []
        os.makedirs(output_dir, exist_ok=True)

    # -- deployment ---------------------------------------------------------------
    def service_name(self, config: Dict[str, Any]) -> str:
        return f"{self.config['benchmark']['service']}-{config['name']}"

    def manifest_for(self, config: Dict[str, Any]) -> Dict[str, Any]:
        b = self.config["benchmark"]
        tp = int(config.get("tensor_parallel_size", 1) or 1)
        return mf.vllm_rocm_isvc(
            self.service_name(config), b["namespace"], model_uri=b.get("model_uri"), model_name=b["model"], tensor_parallel_size=tp,
            max_model_len=int(config.get("max_model_len", 2048)), gpu_memory_utilization=float(config.get("gpu_memory_utilization", 0.9)),
            vllm_args=vllm_args_for(config), cpu="4", memory="32Gi", annotations={"serving.kserve.io/deploymentMode": "Serverless"},
        )

    def deploy(self, config: Dict[str, Any]) -> Optional[str]:
        """Bring the service up and return its URL (None: discover through the backend)."""
        b = self.config["benchmark"]
        if self.local:
            from kvmini.deploy.cli import start_local_server

            ns = argparse.Namespace(local_server=self.local, port=self.port, model_name=b["model"] if self.local == "native" else None, service=self.service_name(config), namespace=b["namespace"], profile=None, timeout=900.0)
            if self.local == "native":
                ns.serve_args = ["--max-model-len", str(config.get("max_model_len", 2048))]
                q = str(config.get("quantization", "none") or "none")
                if q in ("fp8", "mxfp4", "int4", "awq", "gptq"):  # the native engine's weight formats (4-bit names all select MXFP4)
                    ns.serve_args += ["--quantization", q]
                if config.get("max_num_seqs"):
                    ns.serve_args += ["--max-num-seqs", str(config["max_num_seqs"])]
                if str(config.get("kv_cache_dtype", "auto")).startswith("fp8"):
                    ns.serve_args += ["--kv-cache-dtype", "fp8"]
                ns.tensor_parallel_size = int(config.get("tensor_parallel_size", 1) or 1)  # the server starts its own ranks
            info = start_local_server(ns)
            self._local_pid = info["pid"]
            return info["url"]
        self.backend.apply(self.manifest_for(config))
        self.backend.wait_ready(b["namespace"], self.service_name(config), float(self.config.get("ready_timeout_s", 300)))
        return None

    def cleanup(self, config: Dict[str, Any]) -> None:
        try:
            if self.local:
                import signal

                pid = getattr(self, "_local_pid", None)
                if pid:
                    try:
                        os.kill(pid, signal.SIGTERM)
                    except OSError:
                        pass
                    from kvmini.cluster import LocalBackend

                    LocalBackend().mark_finished(pid)
                    self._local_pid = None
                    time.sleep(0.3)
            else:
                self.backend.delete_isvc(self.config["benchmark"]["namespace"], self.service_name(config))
        except Exception as e:
            print(f"  cleanup warning: {e}", file=sys.stderr)

    # -- one configuration --------------------------------------------------------
    def error_result(self, config: Dict[str, Any], error: str) -> Dict[str, Any]:
        return {"config_name": config["name"], "quantization": config.get("quantization", "none"), "kv_cache_dtype": config.get("kv_cache_dtype", "auto"), "error": error, "p95_ms": float("inf"), "throughput_rps": 0, "cost_per_1k_tokens": float("inf"), "quality_score": 0}

    def run_benchmark(self, config: Dict[str, Any]) -> Dict[str, Any]:
        b = self.config["benchmark"]
        run_id = config_run_id(config)
        run_dir = os.path.join(self.runs_root, run_id)
        q = str(config.get("quantization", "none") or "none")
        if q not in ROCM_QUANT:
            return self.error_result(config, f"unsupported_quantization:{q}")
        cmd = [sys.executable, "-m", "kvmini", "bench", "--namespace", b["namespace"], "--service", self.service_name(config), "--model", b["model"], "--requests", str(b["requests"]), "--concurrency", str(b["concurrency"]), "--max-tokens", str(config.get("max_tokens", 64)), "--run-id", run_id, "--run-dir", run_dir, "--cost-file", self.cost_file]
        if self.local:
            cmd += ["--cluster-backend", "local"]
            if self.local == "mock":
                cmd += ["--no-energy"]  # a mock burns no GPU power worth integrating
        lt = loadtest_args_for(config.get("decoding") or {}, self.output_dir, config["name"])
        if lt:
            cmd += ["--loadtest-args", " ".join(lt)]
        try:
            url = self.deploy(config)
            url = self.url or url
            if url:
                cmd += ["--url", url]
            if self.cluster_backend and not self.local:
                cmd += ["--cluster-backend", self.cluster_backend]
            # the child must find this package whatever the working directory is
            pkg_parent = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            env = {**os.environ, "PYTHONPATH": pkg_parent + (os.pathsep + os.environ["PYTHONPATH"] if os.environ.get("PYTHONPATH") else "")}
            proc = subprocess.run(cmd, capture_output=True, text=True, timeout=float(self.config.get("bench_timeout_s", 600)), env=env)
            if proc.returncode != 0:
                sys.stderr.write(proc.stderr[-2000:])
                return self.error_result(config, "benchmark_failed")
            res_path = os.path.join(run_dir, "results.json")
            if not os.path.exists(res_path):
                return self.error_result(config, "results_not_found")
            res = load_json(res_path)
            res.update({
                "config_name": config["name"], "quantization": q, "kv_cache_dtype": config.get("kv_cache_dtype", "auto"), "max_model_len": config.get("max_model_len", 2048),
                "gpu_memory_utilization": config.get("gpu_memory_utilization", 0.9), "tensor_parallel_size": config.get("tensor_parallel_size", 1), "run_id": run_id, "error": "",
            })
            qe = self.config.get("quality_eval") or {}
            if qe.get("enabled"):
                endpoint = url or self.backend.get_isvc_url(b["namespace"], self.service_name(config))
                try:
                    if qe.get("mode") == "agreement":
                        res.update(self.agreement_quality(config, endpoint, qe))
                    else:
                        res.update(QualityEvaluator(endpoint, b["model"], ["hellaswag", "boolq", "math"], 10).evaluate())
                except Exception as e:
                    print(f"  quality evaluation failed: {e}", file=sys.stderr)
                    res["quality_score"] = None
            return res
        except subprocess.TimeoutExpired:
            return self.error_result(config, "timeout")
        except (Exception, SystemExit) as e:
            return self.error_result(config, str(e) or type(e).__name__)
        finally:
            self.cleanup(config)

    def reference_config_name(self) -> str:
        """The configuration the others are compared with in ``quality_eval.mode: agreement``: the
        one marked ``reference: true``, else the first."""
        configs = self.config["sweep_configurations"]
        return next((c["name"] for c in configs if c.get("reference")), configs[0]["name"])

    def agreement_quality(self, config: Dict[str, Any], endpoint: str, qe: Dict[str, Any]) -> Dict[str, Any]:
        """Greedy continuations of fixed prompts against the reference configuration's (kept in
        ``agreement_reference.json`` so that a resumed sweep still has them)."""
        ev = AgreementEvaluator(endpoint, self.config["benchmark"]["model"], int(qe.get("prompts", 96)), int(qe.get("max_tokens", 8)))
        ref_path = os.path.join(self.output_dir, "agreement_reference.json")
        if config["name"] == self.reference_config_name():
            out = ev.evaluate(None)
            write_json(ref_path, {"config_name": config["name"], "prompts": ev.prompts, "continuations": out["continuations"]})
        else:
            if not os.path.exists(ref_path):
                raise RuntimeError(f"no reference continuations yet ({self.reference_config_name()} must run first and succeed)")
            out = ev.evaluate(load_json(ref_path)["continuations"])
        out.pop("continuations")
        return out

    # -- sweep + analysis ------------------------------------------------------------
    FLAT_KEYS = ["config_name", "quantization", "kv_cache_dtype", "max_model_len", "gpu_memory_utilization", "tensor_parallel_size", "run_id", "error", "p50_ms", "p95_ms", "p99_ms", "ttft_p50_ms", "ttft_p95_ms", "throughput_rps", "tokens_per_sec", "error_rate", "cost_per_1k_tokens", "cost_per_request", "energy_wh_per_1k_tokens", "quality_score", "prefix_agreement", "avg_quality_latency_ms"]

    def save_intermediate(self) -> str:
        path = os.path.join(self.output_dir, "sweep_results_intermediate.csv")
        with open(path, "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=self.FLAT_KEYS, extrasaction="ignore")
            w.writeheader()
            for r in self.results:
                w.writerow({k: r.get(k, "") for k in self.FLAT_KEYS})
        return path

    def load_rows(self, path: str) -> List[Dict[str, Any]]:
        """Rows of an intermediate CSV, numbers as numbers, empty cells as None."""
        rows: List[Dict[str, Any]] = []
        with open(path) as f:
            for row in csv.DictReader(f):
                r: Dict[str, Any] = {}
                for k, v in row.items():
                    if v == "":
                        r[k] = None if k != "error" else ""
                    else:
                        try:
                            r[k] = float(v) if k not in ("config_name", "quantization", "kv_cache_dtype", "run_id", "error") else v
                        except ValueError:
                            r[k] = v
                rows.append(r)
        return rows

    def run_sweep(self) -> List[Dict[str, Any]]:
        configs = self.config["sweep_configurations"]
        done: Dict[str, Dict[str, Any]] = {}
        prev = os.path.join(self.output_dir, "sweep_results_intermediate.csv")
        if self.resume and os.path.exists(prev):
            # the reference writes this file after every configuration (sweeps/quantization_sweep.py:343-349)
            # but cannot pick a sweep up from it; here finished (error-free) configurations are kept
            done = {r["config_name"]: r for r in self.load_rows(prev) if not r.get("error")}
        for i, c in enumerate(configs, 1):
            if c["name"] in done:
                print(f"[{i}/{len(configs)}] {c['name']}: resumed from {prev}")
                self.results.append(done[c["name"]])
            else:
                print(f"[{i}/{len(configs)}] Testing: {c['name']}")
                self.results.append(self.run_benchmark(c))
            self.save_intermediate()
        self.generate_analysis()
        return self.results

    def generate_analysis(self) -> Optional[str]:
        clean = [r for r in self.results if not r.get("error")]
        if not clean:
            print("No successful results for analysis", file=sys.stderr)
            return None
        mask = pareto_mask(clean)
        for r, m in zip(clean, mask):
            r["pareto_optimal"] = m
        bucketed = ParetoAnalyzer.classify_pareto_bucket(clean)
        for r, bkt in zip(clean, bucketed):
            r["pareto_bucket"] = bkt["pareto_bucket"]
        frontier = sorted((r for r in clean if r["pareto_optimal"]), key=lambda r: (r.get("cost_per_1k_tokens") is None, r.get("cost_per_1k_tokens") or 0))
        with open(os.path.join(self.output_dir, "pareto_optimal_configs.csv"), "w", newline="") as f:
            keys = self.FLAT_KEYS + ["pareto_optimal", "pareto_bucket"]
            w = csv.DictWriter(f, fieldnames=keys, extrasaction="ignore")
            w.writeheader()
            for r in frontier:
                w.writerow({k: r.get(k, "") for k in keys})
        return self.write_report(clean, frontier)

    def write_report(self, clean: List[Dict[str, Any]], frontier: List[Dict[str, Any]]) -> str:
        quants = sorted({str(r["quantization"]) for r in clean})
        kvs = sorted({str(r["kv_cache_dtype"]) for r in clean})
        parts = [f"<!DOCTYPE html><html><head><meta charset='utf-8'><title>Quantization sweep</title><style>{CSS}</style></head><body><h1>Quantization &amp; decoding sweep</h1>"]
        b = self.config["benchmark"]
        parts.append(f"<p>Model <b>{b['model']}</b> · {b['requests']} requests @ concurrency {b['concurrency']} · {len(clean)}/{len(self.results)} configurations succeeded · {len(frontier)} on the Pareto frontier</p>")
        best = {m: (min if d == "minimize" else max)((r for r in clean if r.get(m) is not None), key=lambda r: r[m], default=None) for m, d in OBJECTIVES}
        parts.append("<h2>Best per objective</h2><ul>" + "".join(f"<li>{m}: <b>{v['config_name']}</b> ({format_number(v[m])})</li>" for m, v in best.items() if v) + "</ul>")
        parts.append("<h2>Heatmaps (quantization × KV-cache dtype)</h2>")
        for m, d in OBJECTIVES:
            cells: Dict[Tuple[Any, Any], Optional[float]] = {}
            for qn in quants:
                for kv in kvs:
                    vals = [float(r[m]) for r in clean if str(r["quantization"]) == qn and str(r["kv_cache_dtype"]) == kv and r.get(m) is not None]
                    cells[(qn, kv)] = sum(vals) / len(vals) if vals else None
            if any(v is not None for v in cells.values()):
                svg = svg_heatmap(quants, kvs, cells, f"{m.replace('_', ' ')} by configuration", lower_is_better=(d == "minimize"), col_label="kv_cache_dtype")
                parts.append(svg)
                with open(os.path.join(self.output_dir, f"heatmap_{m}.svg"), "w") as f:
                    f.write(svg)
        parts.append("<h2>Pareto analysis</h2>")
        scat = []
        for x, y, title in (("cost_per_1k_tokens", "p95_ms", "Cost vs latency"), ("cost_per_1k_tokens", "quality_score", "Cost vs quality"), ("p95_ms", "quality_score", "Latency vs quality"), ("throughput_rps", "quality_score", "Throughput vs quality")):
            if any(r.get(x) is not None and r.get(y) is not None for r in clean):
                scat.append(svg_scatter([(r.get(x), r.get(y), r["config_name"], bool(r.get("pareto_optimal"))) for r in clean], title, x, y))
        parts.append("".join(scat))
        with open(os.path.join(self.output_dir, "pareto_analysis.svg.html"), "w") as f:
            f.write("<html><body>" + "".join(scat) + "</body></html>")
        cols = ["config_name", "quantization", "kv_cache_dtype", "p95_ms", "throughput_rps", "cost_per_1k_tokens", "quality_score", "pareto_optimal"]
        parts.append("<h2>All configurations</h2><table><tr>" + "".join(f"<th>{c}</th>" for c in cols) + "</tr>")
        for r in self.results:
            parts.append("<tr>" + "".join(f"<td>{format_number(r.get(c)) if isinstance(r.get(c), (int, float)) and not isinstance(r.get(c), bool) else (r.get(c) if r.get(c) is not None else '')}</td>" for c in cols) + (f"<td class='bad'>{r['error']}</td>" if r.get("error") else "") + "</tr>")
        parts.append("</table><p class='muted'>Files: sweep_results_intermediate.csv, pareto_optimal_configs.csv, heatmap_*.svg</p></body></html>")
        path = os.path.join(self.output_dir, "sweep_report.html")
        with open(path, "w") as f:
            f.write("\n".join(parts))
        return path


def create_default_config() -> Dict[str, Any]:
    """The default sweep: every (weight format, KV-cache dtype) point an MI355X deployment really has —
    bf16 / FP8 (e4m3) / 4-bit weights x bf16 / FP8 KV — with one decoding variation per weight format, so
    no two rows measure the same configuration.  The six-entry sweep of the reference
    (``sweeps/quantization_sweep.yaml:11-73``) names two 4-bit checkpoint formats (AWQ, GPTQ); on this
    stack they are one kernel format, hence one 4-bit column with two KV dtypes instead."""
    def row(name: str, q: str, kv: str, ctx: int = 2048, util: float = 0.9, **decoding: Any) -> Dict[str, Any]:
        return {"name": name, "quantization": q, "kv_cache_dtype": kv, "max_model_len": ctx, "gpu_memory_utilization": util, "max_tokens": 64,
                "decoding": {"temperature": 0.0, "top_p": 1.0, **decoding}}

    return {
        "benchmark": {"namespace": "benchmark", "service": "llama3-8b-sweep", "model": "llama-3-8b", "requests": 100, "concurrency": 10},
        "quality_eval": {"enabled": True},
        "sweep_configurations": [
            row("bf16-reference", "none", "auto"),
            row("bf16-fp8kv-long-context", "none", "fp8", ctx=8192),
            row("fp8-weights", "fp8", "auto", temperature=0.2, top_p=0.9),
            row("fp8-weights-fp8kv-best-of-4", "fp8", "fp8", ctx=4096, util=0.95, extra_openai={"use_beam_search": True, "best_of": 4}),
            row("w4-weights-sampled", "mxfp4", "auto", util=0.95, temperature=0.7, top_p=0.9),
            row("w4-weights-fp8kv", "mxfp4", "fp8", ctx=4096),
        ],
    }


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(prog="kvmini quant-sweep", description="Quantization & decoding sweep engine")
    ap.add_argument("--config", default="quantization_sweep.yaml")
    ap.add_argument("--output-dir", default="sweeps/quantization")
    ap.add_argument("--create-config", action="store_true")
    ap.add_argument("--local", default=None, choices=["native", "mock"], help="run each configuration against a local server instead of a cluster")
    ap.add_argument("--port", type=int, default=8000)
    ap.add_argument("--cluster-backend", default=None, choices=["auto", "kubectl", "local", "fake"])
    ap.add_argument("--runs-root", default="runs")
    ap.add_argument("--cost-file", default="cost.yaml")
    ap.add_argument("--analyze-only", default=None, metavar="CSV", help="rebuild the analysis from an intermediate CSV")
    ap.add_argument("--url", default=None, help="benchmark this endpoint for every configuration (already deployed / shared service)")
    ap.add_argument("--resume", action="store_true", help="keep the configurations that <output-dir>/sweep_results_intermediate.csv already holds without an error")
    ap.add_argument("--no-quality", action="store_true", help="skip the quality mini-eval even if the config enables it")
    ap.add_argument("--model", default=None, help="override benchmark.model of the config (e.g. llama-3-70b on an 8-GPU node)")
    ap.add_argument("--tensor-parallel-size", type=int, default=None, help="override tensor_parallel_size of every configuration")
    a = ap.parse_args(argv)
    if a.create_config:
        with open(a.config, "w") as f:
            yaml.safe_dump(create_default_config(), f, default_flow_style=False, indent=2, sort_keys=False)
        print(f"Created default config: {a.config}")
        return 0
    if not os.path.exists(a.config):
        print(f"Config file not found: {a.config}\nRun with --create-config to generate default configuration", file=sys.stderr)
        return 1
    cfg = load_yaml(a.config)
    if a.no_quality:
        cfg["quality_eval"] = {"enabled": False}
    if a.model:
        cfg.setdefault("benchmark", {})["model"] = a.model
    if a.tensor_parallel_size:
        for c in cfg.get("sweep_configurations") or []:
            c["tensor_parallel_size"] = a.tensor_parallel_size
    runner = QuantizationSweepRunner(cfg, a.output_dir, a.local, a.cluster_backend, a.runs_root, a.port, a.cost_file, url=a.url, resume=a.resume)
    if a.analyze_only:
        runner.results += runner.load_rows(a.analyze_only)
        runner.generate_analysis()
    else:
        runner.run_sweep()
    print(f"\nQuantization sweep completed.\nResults: {a.output_dir}/sweep_report.html\nPareto configs: {a.output_dir}/pareto_optimal_configs.csv")
    clean = [r for r in runner.results if not r.get("error") and r.get("cost_per_1k_tokens") is not None]
    if clean:
        best = min(clean, key=lambda r: r["cost_per_1k_tokens"])
        print(f"\nTop recommendation: {best['config_name']}\n  cost ${best['cost_per_1k_tokens']:.4f}/1K tokens, p95 {best.get('p95_ms') or 0:.1f} ms, {best.get('quantization')} weights + {best.get('kv_cache_dtype')} KV")
    return 0


if __name__ == "__main__":
    sys.exit(main())

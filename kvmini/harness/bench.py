"""Benchmark orchestrator (``kvmini bench`` / ``bench.sh``).

Stages, as in the reference (``bench.sh:201-289``): 0 validate config, 1 load
test, optional network/storage probe, 2 analyze, 3 cost, 4 bundle.  Same flags
(``--namespace --service --url --requests --concurrency --model --max-tokens
--pattern --profile --prom-url --api-key --run-dir --insecure --cost-file --bundle
--loadtest-args --dry-run --list-profiles``) plus the ones its own callers pass
(``--run-id``: ``tools/chaos_harness.sh:117``, ``sweeps/quantization_sweep.py:85``).

Deliberate differences: runs in one process (no bash → python → bash hops);
CLI flags beat profile values; ``vllm_features`` are not forwarded to the load
generator; streaming is on by default so TTFT / tokens / cost-per-1k-tokens are
populated (``--no-stream`` restores the reference default); an amd-smi sampler
runs around stage 1 and its ``power.json`` is integrated into ``energy.json``;
``runs/latest`` points at the newest run; ``--serve`` starts a local server
(native engine or mock) for the duration of the run, and ``--cold-restart-at``
restarts it mid-run to exercise the cold/warm split without Knative.
"""

from __future__ import annotations

import argparse
import asyncio
import json
import os
import shlex
import sys
import time
from typing import Any, Dict, List, Optional

from kvmini.cluster import get_backend
from kvmini.config.profiles import list_profiles, load_profile, resolve_load_settings
from kvmini.utils import load_json, write_json


def build_arg_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(prog="kvmini bench", description="Run the full benchmark pipeline: validate -> load test -> analyze -> cost [-> bundle]")
    ap.add_argument("--namespace", default="ml-prod")
    ap.add_argument("--service", default="demo-llm")
    ap.add_argument("--url", default=os.environ.get("URL") or None, help="endpoint base URL (default: discovered from the InferenceService)")
    ap.add_argument("--requests", type=int, default=None)
    ap.add_argument("--concurrency", type=int, default=None)
    ap.add_argument("--model", default=os.environ.get("MODEL", "placeholder"))
    ap.add_argument("--max-tokens", type=int, default=None)
    ap.add_argument("--pattern", default=None, choices=["steady", "poisson", "bursty", "heavy"])
    ap.add_argument("--profile", default=None, help="profile name or YAML path (see --list-profiles)")
    ap.add_argument("--prom-url", default=None)
    ap.add_argument("--api-key", default=os.environ.get("API_KEY") or None)
    ap.add_argument("--run-dir", default=None)
    ap.add_argument("--run-id", default=None, help="run directory name under runs/ (default: timestamp)")
    ap.add_argument("--insecure", action="store_true", default=bool(os.environ.get("INSECURE")))
    ap.add_argument("--cost-file", default="cost.yaml")
    ap.add_argument("--bundle", action="store_true")
    ap.add_argument("--loadtest-args", default="", help="extra load generator flags, e.g. '--temperature 0.7 --no-stream'")
    ap.add_argument("--dry-run", action="store_true", help="validate only")
    ap.add_argument("--list-profiles", action="store_true")
    ap.add_argument("--cluster-backend", default=None, choices=["auto", "kubectl", "local", "fake"])
    ap.add_argument("--protocol", default="openai", choices=["openai", "openai-completions", "tgi", "kserve-v2"])
    ap.add_argument("--no-stream", action="store_true")
    ap.add_argument("--no-energy", action="store_true", help="do not sample amd-smi power during the load test")
    ap.add_argument("--report", action="store_true", help="also write report.html")
    ap.add_argument("--slo", default=None, help="run the SLO gate against this file after the pipeline")
    ap.add_argument("--serve", default=None, choices=["native", "mock"], help="start a local server for this run")
    ap.add_argument("--serve-model", default="llama-3-8b")
    ap.add_argument("--serve-port", type=int, default=8000)
    ap.add_argument("--cold-restart-at", type=float, default=None, help="with --serve: restart the server after this fraction of the requests")
    return ap


def _latest_symlink(run_dir: str) -> None:
    parent = os.path.dirname(os.path.abspath(run_dir))
    link = os.path.join(parent, "latest")
    try:
        if os.path.islink(link) or os.path.exists(link):
            os.remove(link)
        os.symlink(os.path.basename(os.path.abspath(run_dir)), link)
    except OSError:
        pass


EXIT_NO_SUCCESS = 4  # every request of the load test failed


def main(argv: Optional[List[str]] = None) -> int:
    a = build_arg_parser().parse_args(argv)
    if a.list_profiles:
        groups = list_profiles()
        print("Available profiles (runners/profiles):")
        for p in groups["runners/profiles"]:
            print(f"  - {p}")
        print("\nOther profiles:")
        print("  - profiles/partition/*.yaml (MI355X compute-partition / SR-IOV scheduling profiles)")
        print("  - profiles/engine/*.yaml (native-engine build profiles)")
        return 0

    try:
        profile = load_profile(a.profile)
    except (FileNotFoundError, ValueError) as e:
        print(f"ERROR: {e}", file=sys.stderr)
        return 1
    s = resolve_load_settings({"requests": a.requests, "concurrency": a.concurrency, "max_tokens": a.max_tokens, "pattern": a.pattern, "stream": False if a.no_stream else None}, profile)
    backend = get_backend(a.cluster_backend)

    print("=== 0/4 Configuration validation ===")
    from kvmini.gates.validate_config import ConfigValidator

    cfg: Dict[str, Any] = dict(profile.raw)
    cfg.update({"requests": s["requests"], "concurrency": s["concurrency"], "max_tokens": s["max_tokens"]})
    v = ConfigValidator()
    ok = v.validate_profile(cfg)
    for w in v.warnings:
        print(f"  WARNING: {w}", file=sys.stderr)
    if not ok:
        for e in v.errors:
            print(f"  ERROR: {e}", file=sys.stderr)
        print("FATAL: Configuration validation failed. Review errors above and fix configuration.", file=sys.stderr)
        return 1
    print("Configuration validation PASSED.")
    if a.dry_run:
        print("=== 1/4 Load test ===\n--dry-run enabled: validation only; skipping load test, analyze, and cost.")
        return 0

    run_dir = a.run_dir or os.path.join("runs", a.run_id or time.strftime("%Y-%m-%d_%H-%M-%S"))
    os.makedirs(run_dir, exist_ok=True)

    server_proc = None
    url = a.url
    if a.serve:
        from kvmini.deploy.cli import start_local_server

        ns = argparse.Namespace(local_server=a.serve, port=a.serve_port, model_name=a.serve_model if a.serve == "native" else None, service=a.service, namespace=a.namespace, profile=a.profile, timeout=900.0)
        info = start_local_server(ns)
        url = info["url"]
        server_proc = info["pid"]
        from kvmini.cluster import LocalBackend

        backend = LocalBackend()
        if a.model == "placeholder" and a.serve == "native":
            a.model = a.serve_model
    if not url:
        print(f"Discovering InferenceService URL for {a.service} in {a.namespace} ...", file=sys.stderr)
        try:
            url = backend.get_isvc_url(a.namespace, a.service)
        except Exception:
            url = None
        if not url:
            print("ERROR: Could not determine URL. Provide --url explicitly or ensure the InferenceService is READY.", file=sys.stderr)
            return 1

    # ---- 1/4 load test ------------------------------------------------------
    print("=== 1/4 Load test ===")
    from kvmini.loadgen import runner as lg

    lt_argv = [
        "--url", url, "--model", a.model, "--requests", str(s["requests"]), "--concurrency", str(s["concurrency"]), "--max-tokens", str(s["max_tokens"]),
        "--pattern", s["pattern"], "--run-dir", run_dir, "--protocol", a.protocol,
    ]
    if a.api_key:
        lt_argv += ["--api-key", a.api_key]
    if a.insecure:
        lt_argv.append("--insecure")
    if s.get("target_rps") is not None:
        lt_argv += ["--target-rps", str(s["target_rps"] if s["target_rps"] > 0 else 1e6)]
    if s.get("prompt_tokens"):
        lt_argv += ["--prompt-tokens", str(s["prompt_tokens"])]
    if s.get("stream") is False:
        lt_argv.append("--no-stream")
    extra = shlex.split(a.loadtest_args) if a.loadtest_args else []
    lt_args = lg.build_arg_parser().parse_args(lt_argv + extra)
    cfg_l = lg.config_from_args(lt_args)
    if profile.test_schema and not cfg_l.spec.response_schema:
        cfg_l.spec.response_schema = profile.test_schema
    if profile.tools and not cfg_l.spec.tools:
        cfg_l.spec.tools = profile.openai_tools()

    sampler = None
    if not a.no_energy:
        from kvmini.telemetry.native_rt import make_sampler

        sampler = make_sampler(0.1)
        if sampler.working():
            sampler.start()
        else:
            sampler = None

    restart_task = None
    if a.serve and a.cold_restart_at:
        done_count = {"n": 0}
        threshold = int(s["requests"] * a.cold_restart_at)

        def on_result(_r: Any) -> None:
            done_count["n"] += 1
            if done_count["n"] == threshold:
                import signal
                import threading

                def _restart() -> None:
                    from kvmini.cluster import LocalBackend
                    from kvmini.deploy.cli import start_local_server

                    try:
                        os.kill(server_proc, signal.SIGTERM)
                    except OSError:
                        pass
                    LocalBackend().mark_finished(server_proc)
                    time.sleep(0.5)
                    ns2 = argparse.Namespace(local_server=a.serve, port=a.serve_port, model_name=a.serve_model if a.serve == "native" else None, service=a.service, namespace=a.namespace, profile=a.profile, timeout=900.0)
                    start_local_server(ns2)

                threading.Thread(target=_restart, daemon=True).start()

        cfg_l.on_result = on_result
        # requests that arrive while the server restarts wait for it (as behind Knative's
        # activator) instead of failing fast: their latency carries the cold start
        cfg_l.connect_retry_s = max(cfg_l.connect_retry_s, 300.0)

    def usage() -> Any:
        try:
            return backend.usage_snapshot(a.namespace, a.service)
        except Exception:
            return None

    t0 = time.time()
    use0 = usage()
    try:
        results = lg.run_load_sync(cfg_l)
    except KeyboardInterrupt:
        print("Aborted")
        return 130
    finally:
        if sampler is not None:
            sampler.stop()
            write_json(os.path.join(run_dir, "power.json"), sampler.to_power_json())
    use1 = usage()
    if use0 and use1 and use1["cpu_seconds"] >= use0["cpu_seconds"]:
        # what the service's processes really used during the load test (the cost record prices this
        # instead of the manifest's nominal requests)
        write_json(os.path.join(run_dir, "resource_usage.json"), {
            "source": f"/proc of {use1['processes']} process(es) via cluster backend '{backend.name}'",
            "window_s": use1["t"] - use0["t"], "cpu_core_seconds": use1["cpu_seconds"] - use0["cpu_seconds"],
            "mean_cpu_cores": (use1["cpu_seconds"] - use0["cpu_seconds"]) / max(use1["t"] - use0["t"], 1e-9),
            "rss_bytes_start": use0["rss_bytes"], "rss_bytes_end": use1["rss_bytes"], "peak_rss_bytes": max(use0["peak_rss_bytes"], use1["peak_rss_bytes"])})
    n_ok = sum(1 for r in results if r.status == 200)
    print(f"Load test: {len(results)} requests, {n_ok} ok, {time.time() - t0:.1f}s")

    # ---- optional probe ----------------------------------------------------------
    if not os.environ.get("DISABLE_IO_PROBE"):
        print("\n=== (optional) Network/Storage Probe ===")
        try:
            from kvmini.harness.net_storage_probe import probe

            probe(url, os.path.join(run_dir, "io_probe.json"), os.environ.get("S3_OBJECT_URL"), insecure=a.insecure)
        except Exception as e:
            print(f"probe skipped: {e}", file=sys.stderr)

    # ---- 2/4 analyze ------------------------------------------------------------------
    print("\n=== 2/4 Analyze ===")
    from kvmini.analysis.reduce import analyze_run

    analyze_run(run_dir, a.namespace, a.service, a.prom_url, backend=backend, quiet=True)
    if os.path.exists(os.path.join(run_dir, "power.json")):
        from kvmini.telemetry import energy as en

        en.integrate_energy(argparse.Namespace(run_dir=run_dir, power=None, include_warmup=False, idle_tax="none", merge_results=True))

    # ---- 3/4 cost -----------------------------------------------------------------------
    print("\n=== 3/4 Cost ===")
    from kvmini.cost.estimator import estimate_cost

    try:
        estimate_cost(run_dir, a.namespace, a.service, a.cost_file, backend)
    except (FileNotFoundError, ValueError) as e:
        print(f"cost estimation failed: {e}", file=sys.stderr)

    from kvmini.analysis.reduce import add_alias_keys
    from kvmini.utils import merge_json_file

    merged = load_json(os.path.join(run_dir, "results.json"))
    merge_json_file(os.path.join(run_dir, "results.json"), add_alias_keys(merged))
    _latest_symlink(run_dir)

    if a.report:
        from kvmini.report.generator import generate_single_report

        generate_single_report(os.path.join(run_dir, "results.json"), os.path.join(run_dir, "report.html"))

    print("\n=== DONE ===" if n_ok else "\n=== FAILED: no request succeeded ===")
    print(f"Results: {run_dir}/results.json")
    res = load_json(os.path.join(run_dir, "results.json"))
    brief = {k: res.get(k) for k in ("p50_ms", "p95_ms", "p99_ms", "throughput_rps", "tokens_per_sec", "error_rate", "time_to_first_token_ms", "cold_start_count", "cost_per_1k_tokens", "energy_wh_per_1k_tokens")}
    print(json.dumps(brief, indent=2))

    rc = 0
    if a.bundle:
        print("\n=== 4/4 Bundle Artifacts ===")
        from kvmini.harness.bundle import bundle_run

        bundle_run(run_dir, a.namespace, a.service, backend=backend)
    if a.slo:
        from kvmini.gates import slo as gate

        rc = gate.main(["--results", os.path.join(run_dir, "results.json"), "--slo", a.slo] + (["--energy", os.path.join(run_dir, "energy.json")] if os.path.exists(os.path.join(run_dir, "energy.json")) else []))
    if a.serve and server_proc:
        backend.delete_isvc(a.namespace, a.service)
    if n_ok == 0:
        # every request failed: the artifacts are written (they show the errors) but the run is
        # not a measurement; callers (sweeps, CI) must not read it as one
        errs = sorted({str(r.error) for r in results if r.error})[:3]
        print(f"ERROR: 0 of {len(results)} requests succeeded" + (f" ({'; '.join(errs)})" if errs else ""), file=sys.stderr)
        return rc or EXIT_NO_SUCCESS
    return rc


if __name__ == "__main__":
    sys.exit(main())

"""Parameter sweeps: grid (``grid-sweep.sh``), Knative autoscaling
(``sweeps/autoscale-sweep.sh``) and GPU partition profiles (``sweeps/mig-sweep.sh``
→ MI355X SPX/DPX/QPX/CPX + SR-IOV VF).

All three call the in-process bench pipeline once per cell into its own run
directory, append one CSV row per cell (same headers as the reference:
``grid-sweep.sh:98``, ``autoscale-sweep.sh:113-115``, ``mig-sweep.sh:70``), write
sentinel rows (``ERROR`` / ``NO_RESULTS`` / ``deploy_failed``) instead of aborting,
and print a ranking at the end.  ``--resume`` skips cells whose run directory
already holds a ``results.json`` (the reference cannot resume).
"""

from __future__ import annotations

import argparse
import csv
import itertools
import json
import os
import statistics
import sys
import time
from typing import Any, Dict, Iterable, List, Optional, Sequence

from kvmini.cluster import get_backend
from kvmini.utils import dns1123_label, load_json, load_yaml, repo_root

GRID_HEADER = ["concurrency", "max_tokens", "pattern", "p50_ms", "p95_ms", "p99_ms", "throughput_rps", "tokens_per_sec", "error_rate", "ttft_p50_ms", "ttft_p95_ms", "cost_per_request", "cost_per_1k_tokens", "gpu_util_avg", "run_dir"]
AUTOSCALE_HEADER = [
    "container_concurrency", "initial_scale", "scale_to_zero_grace_period", "stable_window", "panic_window", "p50_ms", "p95_ms", "p99_ms", "p50_warm_ms", "p95_warm_ms",
    "p50_cold_ms", "p95_cold_ms", "cold_start_count", "cold_multiplier", "throughput_rps", "tokens_per_sec", "error_rate", "cost_per_request", "cost_per_1k_tokens",
    "cold_cost_per_request", "warm_cost_per_request", "cold_cost_per_1k_tokens", "warm_cost_per_1k_tokens", "energy_wh_per_1k_tokens", "gpu_util_avg", "run_dir", "deploy_time_sec", "notes",
]
PARTITION_HEADER = ["profile", "p50_ms", "p95_ms", "throughput_rps", "Wh_per_1k_tokens", "cost_per_1k_tokens", "error_rate"]


def _ints(s: str) -> List[int]:
    return [int(x) for x in s.split(",") if x.strip()]


def _strs(s: str) -> List[str]:
    return [x.strip() for x in s.split(",") if x.strip()]


def _bench(argv: List[str]) -> int:
    from kvmini.harness import bench

    try:
        return bench.main(argv)
    except SystemExit as e:  # argparse / fatal errors inside one cell must not end the sweep
        return int(e.code or 1) if not isinstance(e.code, str) else 1
    except Exception as e:
        print(f"  bench failed: {type(e).__name__}: {e}", file=sys.stderr)
        return 1


def _common_bench_args(a: argparse.Namespace) -> List[str]:
    out = ["--namespace", a.namespace, "--service", a.service, "--model", a.model, "--cost-file", a.cost_file]
    if a.url:
        out += ["--url", a.url]
    if a.prom_url:
        out += ["--prom-url", a.prom_url]
    if a.api_key:
        out += ["--api-key", a.api_key]
    if a.cluster_backend:
        out += ["--cluster-backend", a.cluster_backend]
    if getattr(a, "loadtest_args", ""):
        out += ["--loadtest-args", a.loadtest_args]
    return out


def _add_common(ap: argparse.ArgumentParser) -> None:
    ap.add_argument("--namespace", default="ml-prod")
    ap.add_argument("--service", default="demo-llm")
    ap.add_argument("--url", default=None)
    ap.add_argument("--model", default="placeholder")
    ap.add_argument("--prom-url", default=None)
    ap.add_argument("--api-key", default=None)
    ap.add_argument("--cost-file", default="cost.yaml")
    ap.add_argument("--cluster-backend", default=None, choices=["auto", "kubectl", "local", "fake"])
    ap.add_argument("--native-model", default="llama-3-8b", help="--backend native: the architecture preset the engine serves")
    ap.add_argument("--serve-arg", action="append", default=[], dest="serve_args", help="--backend native: extra `kvmini serve` flag (repeatable)")
    ap.add_argument("--loadtest-args", default="")
    ap.add_argument("--resume", action="store_true", help="skip cells whose results.json already exists")
    ap.add_argument("--runs-root", default="runs")


# ---------------------------------------------------------------------------------------------
def _build_isvc(a: argparse.Namespace, backend: Any, svc: str) -> Dict[str, Any]:
    """The cell's InferenceService.  On a bare node (local cluster backend) only this repository's
    engine can be started, so the native manifest is used whatever ``--backend`` says."""
    from kvmini.cluster import LocalBackend
    from kvmini.deploy import manifests as mf

    kind = a.backend
    if isinstance(backend, LocalBackend) and kind != "native":
        print(f"  local cluster backend: deploying the native engine instead of {kind}")
        kind = "native"
    if kind == "native":
        return mf.native_isvc(svc, a.namespace, a.native_model, serve_args=[str(x) for x in a.serve_args])
    builder = mf.BUILDERS[kind]
    return builder(svc, a.namespace, a.model_uri) if kind.startswith("vllm") else builder(svc, a.namespace)


def main_grid(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(prog="kvmini grid-sweep", description="concurrency x max_tokens x pattern sweep")
    _add_common(ap)
    ap.add_argument("--requests", type=int, default=100)
    ap.add_argument("--concurrency", default="5,10,20", help="comma list (e.g. 1,2,4,8,16,32,64)")
    ap.add_argument("--max-tokens", default="32,64,128")
    ap.add_argument("--patterns", default="steady,poisson,bursty")
    ap.add_argument("--prompt-tokens", default="", help="optional comma list: adds a long-context axis (synthetic prompts)")
    ap.add_argument("--output", default=None)
    ap.add_argument("--report", action="store_true")
    a = ap.parse_args(argv)
    ts = time.strftime("%Y%m%d_%H%M%S")
    out = a.output or f"sweep_results_{ts}.csv"
    concs, toks, pats = _ints(a.concurrency), _ints(a.max_tokens), _strs(a.patterns)
    ptoks = _ints(a.prompt_tokens) or [0]
    cells = list(itertools.product(concs, toks, pats, ptoks))
    print(f"=== Grid sweep: {len(cells)} runs -> {out} ===")
    skipped = 0
    with open(out, "w", newline="") as f:  # rows of resumed cells are rebuilt from their results.json
        w = csv.writer(f)
        w.writerow(GRID_HEADER)
        for i, (c, t, p, pt) in enumerate(cells, 1):
            tag = f"c{c}_t{t}_{p}" + (f"_p{pt}" if pt else "")
            run_dir = os.path.join(a.runs_root, f"grid_{ts if not a.resume else 'resume'}_{tag}")
            print(f"[{i}/{len(cells)}] concurrency={c} max_tokens={t} pattern={p}" + (f" prompt_tokens={pt}" if pt else ""))
            res_path = os.path.join(run_dir, "results.json")
            if a.resume and os.path.exists(res_path):
                skipped += 1
                print("  resume: results.json exists, not re-run")
            else:
                cell = argparse.Namespace(**vars(a))
                if pt:
                    cell.loadtest_args = (a.loadtest_args + f" --prompt-tokens {pt}").strip()
                args = _common_bench_args(cell) + ["--requests", str(a.requests), "--concurrency", str(c), "--max-tokens", str(t), "--pattern", p, "--run-dir", run_dir]
                rc = _bench(args)
                if rc != 0:
                    if os.path.exists(res_path):
                        os.replace(res_path, res_path + ".failed")  # a failed cell must not look finished to --resume
                    w.writerow([c, t, p] + ["ERROR"] * 11 + [run_dir])
                    f.flush()
                    continue
            if not os.path.exists(res_path):
                w.writerow([c, t, p] + ["NO_RESULTS"] * 11 + [run_dir])
                f.flush()
                continue
            r = load_json(res_path)
            w.writerow([c, t, p] + [r.get(k, "N/A") if r.get(k) is not None else "N/A" for k in GRID_HEADER[3:-1]] + [run_dir])
            f.flush()
    rows = [r for r in csv.DictReader(open(out)) if r["p95_ms"] not in ("ERROR", "NO_RESULTS", "N/A")]
    if rows:
        print("\nTop 5 by p95 latency:")
        for r in sorted(rows, key=lambda r: float(r["p95_ms"]))[:5]:
            print(f"  c={r['concurrency']} t={r['max_tokens']} {r['pattern']}: p95={float(r['p95_ms']):.1f}ms rps={r['throughput_rps']}")
        print("Top 5 by throughput:")
        for r in sorted(rows, key=lambda r: -float(r["throughput_rps"]) if r["throughput_rps"] not in ("N/A", "") else 0)[:5]:
            print(f"  c={r['concurrency']} t={r['max_tokens']} {r['pattern']}: rps={r['throughput_rps']} p95={float(r['p95_ms']):.1f}ms")
    if a.report:
        from kvmini.report.generator import generate_grid_report

        print("Report:", generate_grid_report(out, os.path.splitext(out)[0] + "_report.html"))
    print(f"\nSweep results: {out}" + (f" ({skipped} cells resumed)" if skipped else ""))
    return 0


# ---------------------------------------------------------------------------------------------
def autoscale_summary(csv_path: str) -> Dict[str, Any]:
    """Best configs + the scale-to-zero vs pre-warmed trade-off (reference computes
    this with pandas, ``autoscale-sweep.sh:345-415``)."""
    rows = [r for r in csv.DictReader(open(csv_path)) if r.get("notes") == "success" and r.get("p95_ms") not in ("", None)]
    if not rows:
        return {"successful_runs": 0}

    def fl(r: Dict[str, str], k: str) -> Optional[float]:
        try:
            return float(r[k])
        except (KeyError, ValueError):
            return None

    best_lat = min(rows, key=lambda r: fl(r, "p95_ms") or 1e18)
    cost_rows = [r for r in rows if fl(r, "cost_per_1k_tokens")]
    s0 = [r for r in rows if r["initial_scale"] == "0"]
    warm = [r for r in rows if r["initial_scale"] != "0"]

    def med(rs: List[Dict[str, str]], k: str) -> Optional[float]:
        xs = [fl(r, k) for r in rs]
        xs = [x for x in xs if x is not None]
        return statistics.median(xs) if xs else None

    return {
        "successful_runs": len(rows),
        "best_p95": {k: best_lat[k] for k in AUTOSCALE_HEADER[:5] + ["p95_ms"]},
        "best_cost": ({k: min(cost_rows, key=lambda r: fl(r, "cost_per_1k_tokens"))[k] for k in AUTOSCALE_HEADER[:5] + ["cost_per_1k_tokens"]} if cost_rows else None),
        "scale_to_zero": {"runs": len(s0), "median_p95_ms": med(s0, "p95_ms"), "median_cold_multiplier": med(s0, "cold_multiplier"), "median_cost_per_1k_tokens": med(s0, "cost_per_1k_tokens")},
        "pre_warmed": {"runs": len(warm), "median_p95_ms": med(warm, "p95_ms"), "median_cold_multiplier": med(warm, "cold_multiplier"), "median_cost_per_1k_tokens": med(warm, "cost_per_1k_tokens")},
    }


def _autoscale_row(cc: int, init: int, grace: str, stable: str, panic: str, r: Dict[str, Any], run_dir: str, deploy_s: Any) -> List[Any]:
    c95, w95 = r.get("cold_p95_ms"), r.get("warm_p95_ms")
    mult = c95 / w95 if c95 and w95 else ""
    g = lambda k: "" if r.get(k) is None else r.get(k)  # noqa: E731
    return [cc, init, grace, stable, panic, g("p50_ms"), g("p95_ms"), g("p99_ms"), g("warm_p50_ms"), g("warm_p95_ms"), g("cold_p50_ms"), g("cold_p95_ms"), g("cold_start_count"), mult,
            g("throughput_rps"), g("tokens_per_sec"), g("error_rate"), g("cost_per_request"), g("cost_per_1k_tokens"), g("cold_cost_per_request"), g("warm_cost_per_request"),
            g("cold_cost_per_1k_tokens"), g("warm_cost_per_1k_tokens"), g("energy_wh_per_1k_tokens"), g("gpu_util_avg"), run_dir, deploy_s, "success"]


def main_autoscale(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(prog="kvmini autoscale-sweep", description="Knative autoscaling parameter sweep")
    _add_common(ap)
    ap.add_argument("--requests", type=int, default=200)
    ap.add_argument("--container-concurrency", default="1,4,8,16")
    ap.add_argument("--initial-scale", default="0,1,2")
    ap.add_argument("--scale-to-zero-grace", default="30s,60s,300s")
    ap.add_argument("--stable-window", default="60s,120s")
    ap.add_argument("--panic-window", default="6s,10s")
    ap.add_argument("--backend", default="vllm-rocm", choices=["vllm-rocm", "tgi-rocm", "native"])
    ap.add_argument("--model-uri", default=None)
    ap.add_argument("--settle-seconds", type=float, default=45.0)
    ap.add_argument("--output", default=None)
    ap.add_argument("--max-cells", type=int, default=0, help="stop after N cells (0 = all)")
    ap.add_argument("--no-cleanup", action="store_true")
    a = ap.parse_args(argv)
    from kvmini.deploy import manifests as mf

    backend = get_backend(a.cluster_backend)
    ts = time.strftime("%Y%m%d_%H%M%S")
    out = a.output or f"autoscale_sweep_{ts}.csv"
    grid = list(itertools.product(_ints(a.container_concurrency), _ints(a.initial_scale), _strs(a.scale_to_zero_grace), _strs(a.stable_window), _strs(a.panic_window)))
    if a.max_cells:
        grid = grid[: a.max_cells]
    print(f"Total parameter combinations: {len(grid)}")
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(AUTOSCALE_HEADER)
        for i, (cc, init, grace, stable, panic) in enumerate(grid, 1):
            svc = dns1123_label(f"{a.service}-as{i}")
            run_dir = os.path.join(a.runs_root, f"autoscale_{ts if not a.resume else 'resume'}_cc{cc}_is{init}_g{grace}_s{stable}_p{panic}")
            print(f"[{i}/{len(grid)}] target={cc} initialScale={init} grace={grace} stable={stable} panic={panic}")
            res_path = os.path.join(run_dir, "results.json")
            if a.resume and os.path.exists(res_path):
                print("  resume: results.json exists, not re-deployed")
                w.writerow(_autoscale_row(cc, init, grace, stable, panic, load_json(res_path), run_dir, load_json(res_path).get("deploy_time_sec", "")))
                f.flush()
                continue
            isvc = _build_isvc(a, backend, svc)
            # panic window is a percentage of the stable window in Knative
            try:
                pct = f"{100.0 * float(panic.rstrip('s')) / float(stable.rstrip('s')):.1f}"
            except ValueError:
                pct = None
            isvc = mf.apply_autoscaling(isvc, target=cc, initial_scale=init, min_scale=0 if init == 0 else init, max_scale=max(4, init), scale_to_zero_grace=grace, stable_window=stable, panic_window_pct=pct)
            t0 = time.time()
            try:
                backend.apply(isvc)
                backend.wait_ready(a.namespace, svc, 600)
                deploy_s = time.time() - t0
            except Exception as e:
                w.writerow([cc, init, grace, stable, panic] + [""] * 20 + [run_dir, "", f"deploy_failed: {e}"])
                f.flush()
                continue
            if a.settle_seconds:
                time.sleep(a.settle_seconds)
            url = a.url or backend.get_isvc_url(a.namespace, svc)
            args = ["--namespace", a.namespace, "--service", svc, "--requests", str(a.requests), "--concurrency", "10", "--max-tokens", "64", "--pattern", "poisson", "--run-dir", run_dir, "--model", "autoscale-test", "--cost-file", a.cost_file]
            if url:
                args += ["--url", url]
            if a.prom_url:
                args += ["--prom-url", a.prom_url]
            if a.api_key:
                args += ["--api-key", a.api_key]
            if a.cluster_backend:
                args += ["--cluster-backend", a.cluster_backend]
            rc = _bench(args)
            if rc == 0 and os.path.exists(res_path):
                from kvmini.utils import merge_json_file

                merge_json_file(res_path, {"deploy_time_sec": round(deploy_s, 1)})  # a resumed sweep reads it back
                w.writerow(_autoscale_row(cc, init, grace, stable, panic, load_json(res_path), run_dir, f"{deploy_s:.1f}"))
            else:
                if os.path.exists(res_path):
                    os.replace(res_path, res_path + ".failed")  # a failed cell must not look finished to --resume
                w.writerow([cc, init, grace, stable, panic] + [""] * 20 + [run_dir, f"{deploy_s:.1f}", "benchmark_failed"])
            f.flush()
            if not a.no_cleanup:
                try:
                    backend.delete_isvc(a.namespace, svc)
                except Exception:
                    pass
    summary = autoscale_summary(out)
    with open(os.path.splitext(out)[0] + "_summary.json", "w") as sf:
        json.dump(summary, sf, indent=2)
    print(json.dumps(summary, indent=2))
    return 0


# ---------------------------------------------------------------------------------------------
def main_partition(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(prog="kvmini partition-sweep", description="Benchmark one service across MI355X partition profiles")
    _add_common(ap)
    ap.add_argument("--profiles", default="cpx,qpx,dpx,spx", help="names under runners/profiles/partition or YAML paths")
    ap.add_argument("--requests", type=int, default=200)
    ap.add_argument("--concurrency", type=int, default=10)
    ap.add_argument("--max-tokens", type=int, default=64)
    ap.add_argument("--backend", default="vllm-rocm", choices=["vllm-rocm", "tgi-rocm", "native"])
    ap.add_argument("--model-uri", default=None)
    ap.add_argument("--output", default="partition_matrix.csv")
    ap.add_argument("--report", action="store_true")
    a = ap.parse_args(argv)
    from kvmini.config.profiles import resolve_profile_path
    from kvmini.deploy import manifests as mf

    backend = get_backend(a.cluster_backend)
    ts = time.strftime("%Y%m%d_%H%M%S")
    with open(a.output, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(PARTITION_HEADER)
        for name in _strs(a.profiles):
            prof = load_yaml(resolve_profile_path(name))
            pname = str(prof.get("profile", name))
            svc = dns1123_label(f"{a.service}-{pname}")
            isvc = _build_isvc(a, backend, svc)
            isvc = mf.apply_partition_profile(isvc, prof)
            run_dir = os.path.join(a.runs_root, f"partition_{ts if not a.resume else 'resume'}_{pname}")
            print(f"=== partition profile {pname} ===")
            res_path = os.path.join(run_dir, "results.json")
            if a.resume and os.path.exists(res_path):
                print("  resume: results.json exists, not re-deployed")
                r = load_json(res_path)
                w.writerow([pname, r.get("p50_ms"), r.get("p95_ms"), r.get("throughput_rps"), r.get("energy_wh_per_1k_tokens"), r.get("cost_per_1k_tokens"), r.get("error_rate")])
                f.flush()
                continue
            try:
                backend.apply(isvc)
                backend.wait_ready(a.namespace, svc, 600)
            except Exception as e:
                print(f"  deploy failed: {e}", file=sys.stderr)
                w.writerow([pname] + ["ERROR"] * 6)
                continue
            url = a.url or backend.get_isvc_url(a.namespace, svc)
            args = ["--namespace", a.namespace, "--service", svc, "--requests", str(a.requests), "--concurrency", str(a.concurrency), "--max-tokens", str(a.max_tokens), "--run-dir", run_dir, "--model", a.model, "--cost-file", a.cost_file]
            if url:
                args += ["--url", url]
            if a.cluster_backend:
                args += ["--cluster-backend", a.cluster_backend]
            rc = _bench(args)
            if rc != 0 and os.path.exists(res_path):
                os.replace(res_path, res_path + ".failed")
            if rc == 0 and os.path.exists(res_path):
                r = load_json(res_path)
                w.writerow([pname, r.get("p50_ms"), r.get("p95_ms"), r.get("throughput_rps"), r.get("energy_wh_per_1k_tokens"), r.get("cost_per_1k_tokens"), r.get("error_rate")])
            else:
                w.writerow([pname] + ["ERROR" if rc != 0 else "NO_RESULTS"] * 6)
            f.flush()
            try:
                backend.delete_isvc(a.namespace, svc)
            except Exception:
                pass
    if a.report:
        from kvmini.report.generator import generate_partition_report

        print("Report:", generate_partition_report(a.output, os.path.splitext(a.output)[0] + "_report.html"))
    print(f"Partition matrix: {a.output}")
    return 0

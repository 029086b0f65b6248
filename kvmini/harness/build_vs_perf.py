"""Engine build-time vs inference-performance tradeoff (``kvmini build-vs-perf``).

Consumes an engine profile (``profiles/engine/*.yaml``: dtype, quantization,
KV dtype, TP/PP, batch / length limits, ``builder_flags``), times the engine
build, serves it, runs the benchmark pipeline and appends one CSV row with the
build time next to p95 / RPS / mean TTFB (columns of the reference's
``scripts/trtllm_build_vs_perf.py:245-262``).

The reference times a TensorRT-LLM engine compile.  The engine built here has no
offline compile; its "build" is what the native server does before it reports
healthy — loading or initialising weights, re-ordering them into MFMA fragment
order with the RMSNorm gains folded in, sizing the paged KV cache and capturing
one HIP graph per token bucket — so ``build_time_s`` is start → ``/health`` of
``kvmini serve`` launched with the profile's flags, plus the wall time of an
optional ``--builder-cmd`` template (``{placeholders}`` are profile keys and
``builder_flags`` keys; ``{quantization_arg}`` expands to ``--quantization X`` or
nothing) for pipelines that do have an offline step such as weight conversion.
"""

from __future__ import annotations

import argparse
import csv
import os
import shlex
import signal
import subprocess
import sys
import time
from typing import Any, Dict, List, Optional, Tuple

from kvmini.cluster import LocalBackend, get_backend
from kvmini.deploy import manifests as mf
from kvmini.utils import dns1123_label, load_json, load_yaml, redact_cmd

COLUMNS = ["timestamp", "model_family", "model_name", "dtype", "quantization", "kv_cache_dtype", "max_batch_size", "max_input_len", "max_output_len", "tp_size", "pp_size", "build_time_s", "p95_total_ms", "throughput_req_per_sec", "mean_ttfb_ms"]


def run_timed(cmd: Any, env: Optional[Dict[str, str]] = None, timeout: Optional[float] = None) -> Tuple[subprocess.CompletedProcess, float]:
    print(f"$ {redact_cmd(cmd)}")
    t0 = time.time()
    proc = subprocess.run(cmd if isinstance(cmd, list) else shlex.split(cmd), env=env, capture_output=True, text=True, timeout=timeout)
    dt = time.time() - t0
    if proc.returncode != 0:
        sys.stderr.write(proc.stdout[-2000:] + proc.stderr[-2000:])
        raise RuntimeError(f"command failed ({proc.returncode}): {redact_cmd(cmd)}")
    return proc, dt


def builder_time(profile: Dict[str, Any], template: Optional[str]) -> float:
    if not template:
        return 0.0
    subs = dict(profile)
    subs.update(profile.get("builder_flags") or {})
    q = profile.get("quantization", "none")
    subs["quantization_arg"] = "" if q in (None, "", "none") else f"--quantization {q}"
    try:
        cmd = template.format(**subs)
    except KeyError as e:
        raise KeyError(f"missing placeholder {e} in --builder-cmd template; available keys: {sorted(subs)}") from e
    env = None
    if isinstance(profile.get("runtime_env"), dict):
        env = {**os.environ, **{k: str(v) for k, v in profile["runtime_env"].items()}}
    _, dt = run_timed(cmd, env=env)
    return dt


def serve_args_from_profile(profile: Dict[str, Any]) -> List[str]:
    bf = profile.get("builder_flags") or {}
    args: List[str] = []
    for key, flag in (("max_num_seqs", "--max-num-seqs"), ("max_batch_size", "--max-num-seqs"), ("max_batched_tokens", "--max-batched-tokens"), ("max_model_len", "--max-model-len")):
        if profile.get(key) is not None and flag not in args:
            args += [flag, str(profile[key])]
    if profile.get("dtype") in ("bf16", "fp16", "fp32"):
        args += ["--dtype", profile["dtype"]]
    if profile.get("quantization") == "fp8":
        args += ["--quantization", "fp8"]
    if bf.get("hip_graphs") is False:
        args.append("--no-graphs")
    if bf.get("prefix_caching") is False:
        args.append("--no-prefix-caching")
    return args


def write_row(path: str, row: Dict[str, Any]) -> None:
    new = not os.path.exists(path)
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    with open(path, "a", newline="") as f:
        w = csv.DictWriter(f, fieldnames=COLUMNS)
        if new:
            w.writeheader()
        w.writerow(row)


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(prog="kvmini build-vs-perf", description="Engine build time vs serving performance")
    ap.add_argument("--profile", required=True, help="profiles/engine/*.yaml")
    ap.add_argument("--namespace", default="ml-prod")
    ap.add_argument("--requests", type=int, default=200)
    ap.add_argument("--concurrency", type=int, default=10)
    ap.add_argument("--max-tokens", type=int, default=64)
    ap.add_argument("--streaming", action="store_true")
    ap.add_argument("--builder-cmd", default=None, help="offline build command template with {placeholders} from the profile")
    ap.add_argument("--output", default="engine_tradeoffs.csv")
    ap.add_argument("--mode", default="local", choices=["local", "cluster", "mock"], help="serve with a local native server, a cluster InferenceService, or the mock (CI)")
    ap.add_argument("--port", type=int, default=8000)
    ap.add_argument("--cluster-backend", default=None, choices=["auto", "kubectl", "local", "fake"])
    ap.add_argument("--cost-file", default="cost.yaml")
    a = ap.parse_args(argv)
    if not os.path.exists(a.profile):
        raise SystemExit(f"Profile not found: {a.profile}")
    profile = load_yaml(a.profile)
    model = profile.get("model_name", "llama-3-8b")
    build_s = builder_time(profile, a.builder_cmd)

    from kvmini.harness import bench

    service = dns1123_label(f"{model}-native-{int(time.time())}")
    run_dir = os.path.join("runs", f"engine-{int(time.time())}")
    results: Dict[str, Any] = {}
    pid = None
    cluster = None
    try:
        if a.mode in ("local", "mock"):
            from kvmini.deploy.cli import start_local_server

            ns = argparse.Namespace(local_server="native" if a.mode == "local" else "mock", port=a.port, model_name=model if a.mode == "local" else None, service=service, namespace=a.namespace, profile=None, timeout=1800.0, serve_args=serve_args_from_profile(profile) if a.mode == "local" else [])
            info = start_local_server(ns)
            url, pid = info["url"], info["pid"]
            build_s += info["ready_s"]
            backend_name = "local"
        else:
            cluster = get_backend(a.cluster_backend)
            tp = int(profile.get("tensor_parallel_size", 1) or 1) * int(profile.get("pipeline_parallel_size", 1) or 1)
            t0 = time.time()
            cluster.apply(mf.native_isvc(service, a.namespace, model, gpus=tp, serve_args=serve_args_from_profile(profile)))
            cluster.wait_ready(a.namespace, service, 1800.0)
            build_s += time.time() - t0
            url = cluster.get_isvc_url(a.namespace, service)
            if not url:
                raise RuntimeError("failed to get InferenceService URL")
            backend_name = a.cluster_backend
        argv2 = ["--namespace", a.namespace, "--service", service, "--url", url, "--model", model, "--requests", str(a.requests), "--concurrency", str(a.concurrency), "--max-tokens", str(a.max_tokens), "--pattern", "steady", "--run-dir", run_dir, "--cost-file", a.cost_file]
        if not a.streaming:
            argv2.append("--no-stream")
        if backend_name:
            argv2 += ["--cluster-backend", backend_name]
        bench.main(argv2)
        rp = os.path.join(run_dir, "results.json")
        if os.path.exists(rp):
            results = load_json(rp)
        else:
            print(f"Warning: {rp} not found; using empty metrics.", file=sys.stderr)
    finally:
        if pid:
            try:
                os.kill(pid, signal.SIGTERM)
            except OSError:
                pass
            LocalBackend().mark_finished(pid)
        elif cluster is not None:
            try:
                cluster.delete_isvc(a.namespace, service)
            except Exception:
                pass

    row = {
        "timestamp": time.strftime("%Y-%m-%d %H:%M:%S"),
        "model_family": profile.get("model_family"),
        "model_name": model,
        "dtype": profile.get("dtype"),
        "quantization": profile.get("quantization"),
        "kv_cache_dtype": profile.get("kv_cache_dtype"),
        "max_batch_size": profile.get("max_batch_size", profile.get("max_num_seqs")),
        "max_input_len": profile.get("max_input_len", profile.get("max_model_len")),
        "max_output_len": profile.get("max_output_len", a.max_tokens),
        "tp_size": profile.get("tensor_parallel_size"),
        "pp_size": profile.get("pipeline_parallel_size"),
        "build_time_s": round(build_s, 2),
        "p95_total_ms": results.get("p95_ms", results.get("p95_total_ms", 0.0)),
        "throughput_req_per_sec": results.get("throughput_rps", results.get("throughput_req_per_sec", 0.0)),
        "mean_ttfb_ms": results.get("time_to_first_token_ms", results.get("mean_ttfb_ms", 0.0)),
    }
    write_row(a.output, row)
    print("\n=== Tradeoff result ===")
    for k, v in row.items():
        print(f"{k}: {v}")
    print(f"\nSaved: {a.output}")
    return 0


if __name__ == "__main__":
    sys.exit(main())

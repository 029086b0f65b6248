"""``kvmini deploy`` / ``deploy.sh``: create an InferenceService (or a local server).

Flags of the reference (``deploy.sh:28-74``): ``--namespace --service --model-uri
--runtime --dry-run``; plus the ones its own callers already pass
(``scripts/reference_runner.py:201``: ``--gpu-limit``) and ROCm additions:
``--backend vllm-rocm|tgi-rocm|native``, ``--profile`` (its ``vllm_features`` become
server args), ``--partition-profile`` (MI355X compute partition / SR-IOV VF),
``--tensor-parallel-size``.  With ``--cluster-backend local`` nothing touches
Kubernetes: a ``kvmini serve`` (or ``kvmini mock``) process is started on this host,
registered in the local cluster registry and waited for — the path used on a
single MI355X box.
"""

from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time
from typing import Any, Dict, List, Optional

from kvmini.cluster import FakeBackend, KubectlBackend, LocalBackend, get_backend
from kvmini.config.profiles import load_profile
from kvmini.utils import load_yaml

from . import manifests as mf
from .preflight import run_preflight


def build_manifest(a: argparse.Namespace) -> Dict[str, Any]:
    prof = load_profile(a.profile) if a.profile else None
    vf = dict(prof.vllm_features) if prof else {}
    tp = a.tensor_parallel_size or int(vf.pop("tensor_parallel_size", 1) or 1)
    backend = a.backend
    if backend in ("vllm-rocm", "vllm"):
        device = "cpu" if vf.get("device") == "cpu" else "rocm"
        max_len = int(vf.pop("max_model_len", 8192) or 8192)
        util = vf.pop("gpu_memory_utilization", 0.90) or 0.90
        if prof:
            prof.vllm_features = vf
        m = mf.vllm_rocm_isvc(a.service, a.namespace, a.model_uri, a.model_name, a.runtime, tp, 1, max_len, float(util), prof.vllm_args() if prof else None, device=device)
    elif backend in ("tgi-rocm", "tgi"):
        m = mf.tgi_rocm_isvc(a.service, a.namespace, a.model_name or "meta-llama/Llama-3.1-8B-Instruct", num_shard=tp)
    elif backend == "native":
        m = mf.native_isvc(a.service, a.namespace, a.model_name or "llama-3-8b", gpus=tp)
    else:
        raise SystemExit(f"unknown backend {backend}")
    if a.gpu_limit is not None:
        m = mf.set_gpu_limit(m, a.gpu_limit)
    if a.partition_profile:
        m = mf.apply_partition_profile(m, load_yaml(a.partition_profile))
    return m


def local_manifest(a: argparse.Namespace) -> Dict[str, Any]:
    """The InferenceService a local deployment stands for: the native engine's manifest (profile-driven
    server flags and ``serve_args`` in the container args) or, for ``--local-server mock``, a
    container whose command is ``kvmini mock``.  ``LocalBackend.apply`` turns either into processes."""
    port = getattr(a, "port", None)
    ann = {"kvmini/local-port": str(port)} if port else {}
    if getattr(a, "local_server", "native") == "mock":
        container = {"name": "kserve-container", "image": "kvmini/mock:latest", "command": ["kvmini", "mock"],
                     "args": ["--model", a.model_name or "mock-llm"] + [str(x) for x in (getattr(a, "serve_args", None) or [])]}
        return {"apiVersion": "serving.kserve.io/v1beta1", "kind": "InferenceService", "metadata": mf._meta(a.service, a.namespace, "mock", ann),
                "spec": {"predictor": {"containers": [container]}}}
    serve_args: List[str] = []
    prof = load_profile(a.profile) if getattr(a, "profile", None) else None
    if prof and prof.native_engine.get("draft_model"):
        serve_args += ["--speculative-model", str(prof.native_engine["draft_model"]), "--num-speculative-tokens", str(prof.native_engine.get("num_speculative_tokens", 4))]
    if prof and prof.vllm_features.get("quantization") == "fp8":
        serve_args += ["--quantization", "fp8"]
    serve_args += [str(x) for x in (getattr(a, "serve_args", None) or [])]
    tp = int(getattr(a, "tensor_parallel_size", None) or 1)
    if "--tensor-parallel-size" in serve_args:
        tp = int(serve_args[serve_args.index("--tensor-parallel-size") + 1])
    m = mf.native_isvc(a.service, a.namespace, a.model_name or "llama-3-8b", gpus=tp, serve_args=serve_args, annotations=ann)
    replicas = int(getattr(a, "replicas", None) or 1)
    if replicas > 1:
        m = mf.apply_autoscaling(m, initial_scale=replicas, min_scale=replicas)
    if getattr(a, "partition_profile", None):
        m = mf.apply_partition_profile(m, load_yaml(a.partition_profile))
    return m


def start_local_server(a: argparse.Namespace) -> Dict[str, Any]:
    """Deploy on this host through the local cluster backend's write side (``LocalBackend.apply`` +
    ``wait_ready``): the one way every harness starts, resizes and replaces a local server."""
    lb = LocalBackend()
    manifest = local_manifest(a)
    ns, name = manifest["metadata"]["namespace"], manifest["metadata"]["name"]
    t0 = time.time()
    lb.apply(manifest)
    try:
        lb.wait_ready(ns, name, float(getattr(a, "timeout", 600.0)))
    except (RuntimeError, TimeoutError) as e:
        lb.delete_isvc(ns, name)
        raise SystemExit(f"local server failed to start: {e}")
    dep = lb.get_isvc(ns, name)
    reg = (lb._load().get("isvc") or {}).get(f"{ns}/{name}", {})
    return {"url": dep["status"]["url"], "pid": reg.get("pid"), "ready_s": time.time() - t0}


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(prog="kvmini deploy", description="Deploy an InferenceService (vLLM-ROCm / TGI-ROCm / native) or a local server")
    ap.add_argument("--namespace", default="ml-prod")
    ap.add_argument("--service", default="demo-llm")
    ap.add_argument("--model-uri", default=None, help="storage URI of the model weights (s3://, pvc://, hf://)")
    ap.add_argument("--model-name", default=None)
    ap.add_argument("--runtime", default="kserve-vllm-rocm", help="KServe ServingRuntime name")
    ap.add_argument("--backend", default="vllm-rocm", choices=["vllm-rocm", "vllm", "tgi-rocm", "tgi", "native"])
    ap.add_argument("--profile", default=None, help="benchmark profile whose vllm_features configure the server")
    ap.add_argument("--partition-profile", default=None, help="MI355X partition profile YAML (profiles/partition/*.yaml)")
    ap.add_argument("--gpu-limit", type=int, default=None, help="override the amd.com/gpu limit")
    ap.add_argument("--tensor-parallel-size", type=int, default=None)
    ap.add_argument("--dry-run", action="store_true", help="print the manifest instead of applying it")
    ap.add_argument("--cluster-backend", default=None, choices=["auto", "kubectl", "local", "fake"])
    ap.add_argument("--local-server", default="native", choices=["native", "mock"])
    ap.add_argument("--port", type=int, default=8000)
    ap.add_argument("--replicas", type=int, default=None, help="local backend: engine replicas behind the data-parallel router (clusters: initialScale / minScale)")
    ap.add_argument("--device", default=None, choices=["cpu", "cuda"], help="local native server: run the engine on this device (cpu: tests)")
    ap.add_argument("--delete", action="store_true", help="remove the service instead of creating it")
    ap.add_argument("--serve-arg", action="append", default=[], dest="serve_args", help="extra flag for the local server (repeatable, e.g. --serve-arg=--max-model-len --serve-arg=4096)")
    ap.add_argument("--timeout", type=float, default=600.0)
    ap.add_argument("--skip-preflight", action="store_true")
    a = ap.parse_args(argv)

    print("=== kvmini deployment ===")
    if a.dry_run:
        print("Mode: Dry Run")
    print(f"Namespace: {a.namespace}\nService name: {a.service}\nBackend: {a.backend}\nRuntime: {a.runtime}")
    if a.model_uri:
        print(f"Model URI: {a.model_uri}")
    print("")
    if a.device:
        a.serve_args = list(a.serve_args) + ["--device", a.device]
    local = a.cluster_backend == "local"
    if a.delete:
        backend = get_backend(a.cluster_backend)
        backend.delete_isvc(a.namespace, mf.dns1123_label(a.service))
        print(f"Deleted {a.namespace}/{a.service}")
        return 0
    # a bare node runs this repository's servers (the other runtimes' images need a cluster): the manifest is the native engine's
    manifest = local_manifest(a) if local else build_manifest(a)
    if a.dry_run:
        print("--- Generated InferenceService manifest ---")
        print(mf.to_yaml(manifest))
        return 0
    backend = get_backend(a.cluster_backend)
    if isinstance(backend, LocalBackend) or local:
        info = start_local_server(a)
        print(f"Local server ready in {info['ready_s']:.1f}s: {info['url']} (pid {info['pid']})")
        return 0
    if isinstance(backend, KubectlBackend) and not a.skip_preflight:
        rc = run_preflight(backend, quiet=False)
        if rc != 0:
            return rc
    if isinstance(backend, FakeBackend) and a.cluster_backend != "fake":
        print("kubectl not found and no local/fake backend requested", file=sys.stderr)
        return 2
    print("Deploying InferenceService...")
    backend.apply(manifest)
    print("Waiting for InferenceService to be READY...")
    waited = backend.wait_ready(a.namespace, manifest["metadata"]["name"], a.timeout)
    url = backend.get_isvc_url(a.namespace, manifest["metadata"]["name"])
    print(f"\n=== DEPLOYMENT STATUS ===\nREADY after {waited:.1f}s\nURL: {url}")
    print(json.dumps({"service": manifest["metadata"]["name"], "namespace": a.namespace, "url": url, "ready_s": waited}))
    return 0


if __name__ == "__main__":
    sys.exit(main())

"""Pre-flight checks before touching a cluster (reference
``scripts/preflight-checks.sh:25-51``): kubectl present and pointing at a
context, KServe CRD installed, GPU nodes present (label
``amd.com/gpu.device-id`` / capacity ``amd.com/gpu`` from the AMD GPU operator;
warning only), object-store credentials (warning only)."""

from __future__ import annotations

import os
import subprocess
import sys
from typing import Any


def run_preflight(backend: Any, quiet: bool = False) -> int:
    def say(msg: str) -> None:
        if not quiet:
            print(msg)

    if not backend.available():
        print("kubectl not found", file=sys.stderr)
        return 2
    if not hasattr(backend, "kubectl"):
        # local / fake backends: no kubectl, no CRDs — report what the backend knows about its GPUs
        try:
            facts = backend.cluster_facts()
        except Exception as e:
            print(f"ERROR: preflight failed: {e}", file=sys.stderr)
            return 2
        say(f"cluster backend: {getattr(backend, 'name', type(backend).__name__)}")
        gpu_nodes = [n for n in (facts.get("gpu_nodes") or facts.get("nodes") or []) if n]
        if not gpu_nodes:
            say("WARNING: the backend reports no GPU node")
        return 0
    try:
        ctx = subprocess.run([backend.kubectl, "config", "current-context"], capture_output=True, text=True, timeout=10)
        if ctx.returncode != 0:
            print("ERROR: kubectl has no current context", file=sys.stderr)
            return 2
        say(f"kubectl context: {ctx.stdout.strip()}")
        crd = subprocess.run([backend.kubectl, "get", "crd", "inferenceservices.serving.kserve.io"], capture_output=True, text=True, timeout=15)
        if crd.returncode != 0:
            print("ERROR: KServe CRD inferenceservices.serving.kserve.io not found (install KServe first)", file=sys.stderr)
            return 2
        nodes = subprocess.run([backend.kubectl, "get", "nodes", "-o", "jsonpath={range .items[*]}{.status.capacity.amd\\.com/gpu}{\"\\n\"}{end}"], capture_output=True, text=True, timeout=15)
        if not any(line.strip() not in ("", "0") for line in nodes.stdout.splitlines()):
            say("WARNING: no node advertises amd.com/gpu (is the AMD GPU operator / device plugin installed?)")
    except (OSError, subprocess.SubprocessError) as e:
        print(f"ERROR: preflight failed: {e}", file=sys.stderr)
        return 2
    if not (os.environ.get("AWS_ACCESS_KEY_ID") or os.environ.get("AWS_PROFILE")):
        say("WARNING: no AWS/S3 credentials in the environment (needed only for s3:// model URIs)")
    return 0


def main(argv=None) -> int:
    import argparse

    from kvmini.cluster import KubectlBackend

    ap = argparse.ArgumentParser(prog="kvmini preflight", description="Check kubectl / KServe / GPU nodes before deploying")
    ap.add_argument("--quiet", action="store_true")
    a = ap.parse_args(argv)
    rc = run_preflight(KubectlBackend(), a.quiet)
    if rc == 0 and not a.quiet:
        print("Preflight checks passed")
    return rc

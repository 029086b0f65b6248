"""Artifact bundling and provenance (``kvmini bundle`` / ``tools/bundle_run.sh``),
cluster facts (``tools/collect_cluster_facts.sh``), SBOM (``tools/sbom.sh``) and
image signing (``tools/sign.sh``).

Bundle layout (reference ``tools/bundle_run.sh:110-333``): run artifacts
(results / requests / classified / meta / fairness / energy / power / io probe),
``provenance.json`` (bundle + run meta + cluster facts + reproducibility notes),
``cluster_facts.json``, ``configs/`` (repo config files, deployed ISVC and pod
specs), ``grafana/`` (dashboard JSON + export instructions), ``traces/``,
``SUMMARY.md``, ``supply-chain/`` (image list, SBOMs when ``syft`` exists); the
archive is reproducible (sorted names, fixed mtime, zero uid/gid).  Everything is
done in-process — no ``jq``, no shelling out except optional ``git`` / ``syft`` /
``cosign``.
"""

from __future__ import annotations

import argparse
import datetime as dt
import gzip
import io
import json
import os
import shutil
import subprocess
import sys
import tarfile
from pathlib import Path
from typing import Any, Dict, List, Optional

from kvmini.cluster import ClusterBackend, get_backend
from kvmini.utils import load_json, repo_root, write_json

FIXED_MTIME = int(dt.datetime(2025, 1, 1, tzinfo=dt.timezone.utc).timestamp())
RUN_FILES = ["results.json", "requests.csv", "requests_classified.csv", "meta.json", "fairness_summary.json", "fairness_report.html", "energy.json", "power.json", "io_probe.json", "report.html", "token_events.npy", "cache_analysis.json", "capability_matrix.json"]


def _git(*args: str) -> Optional[str]:
    try:
        r = subprocess.run(["git", *args], capture_output=True, text=True, timeout=10, cwd=str(repo_root()))
        return r.stdout.strip() if r.returncode == 0 else None
    except (OSError, subprocess.SubprocessError):
        return None


def collect_cluster_facts(namespace: str, service: str, backend: Optional[ClusterBackend] = None, at: Optional[float] = None) -> Dict[str, Any]:
    """Kubernetes / GPU-node / serving-stack / git facts for provenance (reference
    ``tools/collect_cluster_facts.sh:52-147``, re-keyed to ``amd.com/gpu`` labels)."""
    backend = backend or get_backend()
    when = dt.datetime.now(dt.timezone.utc) if at is None else dt.datetime.fromtimestamp(at, dt.timezone.utc)
    facts: Dict[str, Any] = {"collected_at": when.isoformat(), "namespace": namespace, "service": service}
    try:
        facts.update(backend.cluster_facts())
    except Exception as e:
        facts["cluster_error"] = str(e)
    try:
        pods = backend.get_pods(namespace, service)
    except Exception:
        pods = []
    facts["pod_images"] = sorted({cs.get("imageID") or cs.get("image") for p in pods for cs in (p.get("status") or {}).get("containerStatuses", []) if cs.get("image")})
    facts["pod_count"] = len(pods)
    isvc = {}
    try:
        isvc = backend.get_isvc(namespace, service)
    except Exception:
        pass
    facts["inferenceservice"] = {"url": (isvc.get("status") or {}).get("url"), "annotations": (isvc.get("metadata") or {}).get("annotations")}
    dirty = _git("status", "--porcelain")
    facts["git"] = {"sha": _git("rev-parse", "HEAD"), "branch": _git("rev-parse", "--abbrev-ref", "HEAD"), "dirty": bool(dirty) if dirty is not None else None}
    try:
        import torch

        facts["software"] = {"torch": torch.__version__, "hip": getattr(torch.version, "hip", None)}
    except Exception:
        pass
    return facts


def generate_sboms(images: List[str], out_dir: str) -> List[str]:
    """SPDX-JSON SBOM per image via ``syft`` when installed; otherwise writes a
    manifest listing the images so the step can be repeated where syft exists."""
    os.makedirs(out_dir, exist_ok=True)
    written: List[str] = []
    syft = shutil.which("syft")
    for img in images:
        safe = img.replace("/", "_").replace(":", "_").replace("@", "_")
        out = os.path.join(out_dir, f"{safe}.spdx.json")
        if syft:
            try:
                r = subprocess.run([syft, "packages", img, "-o", "spdx-json"], capture_output=True, text=True, timeout=600)
                if r.returncode == 0:
                    Path(out).write_text(r.stdout)
                    written.append(out)
                    continue
            except (OSError, subprocess.SubprocessError):
                pass
    if not written:
        todo = os.path.join(out_dir, "PENDING.json")
        write_json(todo, {"note": "syft not available; run `syft packages <image> -o spdx-json` for each image", "images": images})
        written.append(todo)
    return written


def sign_images(images: List[str], key: str, dry_run: bool = False) -> int:
    """``cosign sign --key <key> <image>`` for each image (exit 2 when cosign is missing)."""
    cosign = shutil.which("cosign")
    if not cosign and not dry_run:
        print("cosign not found", file=sys.stderr)
        return 2
    rc = 0
    for img in images:
        cmd = [cosign or "cosign", "sign", "--key", key, img]
        print("+", " ".join(cmd))
        if not dry_run:
            rc |= subprocess.run(cmd).returncode
    return rc


def _summary_md(run_id: str, namespace: str, service: str, results: Dict[str, Any], has_fairness: bool, at: Optional[float] = None) -> str:
    when = dt.datetime.now(dt.timezone.utc) if at is None else dt.datetime.fromtimestamp(at, dt.timezone.utc)

    def g(k: str, fmt: str = "{}") -> str:
        v = results.get(k)
        return "N/A" if v is None else fmt.format(v)

    err = results.get("error_rate")
    lines = [
        "# Run Summary",
        "",
        f"**Run ID**: {run_id}",
        f"**Service**: {service}",
        f"**Namespace**: {namespace}",
        f"**Bundled**: {when.strftime('%Y-%m-%d %H:%M:%S UTC')}",
        "",
        "## Key Results",
        "",
        f"**P95 Latency**: {g('p95_ms', '{:.1f}')}ms",
        f"**TTFT (mean)**: {g('time_to_first_token_ms', '{:.1f}')}ms",
        f"**Throughput**: {g('throughput_rps', '{:.2f}')} RPS",
        f"**Error Rate**: {'N/A' if err is None else f'{err * 100:.2f}'}%",
        f"**Cost/1K Tokens**: ${g('cost_per_1k_tokens', '{:.6f}')}",
        f"**Cold Starts**: {g('cold_start_count')}",
        f"**Energy**: {g('energy_wh_per_1k_tokens', '{:.4f}')} Wh/1K tokens",
        "",
        "## Contents",
        "",
        "- `provenance.json` - reproducibility metadata",
        "- `cluster_facts.json` - cluster / node / GPU state and versions",
        "- `results.json` - benchmark results",
        "- `requests.csv` - per-request timing data",
        "- `configs/` - configuration files and resolved manifests",
        "- `grafana/` - dashboard definitions and export instructions",
        "- `traces/` - OTLP-JSON client traces",
        "- `supply-chain/` - image list and SBOMs",
    ]
    if has_fairness:
        lines.append("- `fairness_summary.json`, `fairness_report.html` - multi-tenant fairness results")
    lines += [
        "",
        "## Reproduction",
        "",
        "1. Use the Kubernetes / KServe / ROCm versions from `cluster_facts.json`",
        "2. Deploy with the pinned image digests from `configs/deployed_isvc.yaml`",
        "3. Run the benchmark parameters from `meta.json`",
        "4. Expect ±10% variance in p95 latency",
        "",
        f"Generated by kvmini {_git('rev-parse', '--short', 'HEAD') or 'unknown'}",
        "",
    ]
    return "\n".join(lines)


def _add_bytes(tar: tarfile.TarFile, name: str, data: bytes) -> None:
    ti = tarfile.TarInfo(name)
    ti.size = len(data)
    ti.mtime = FIXED_MTIME
    ti.mode = 0o644
    ti.uid = ti.gid = 0
    ti.uname = ti.gname = ""
    tar.addfile(ti, io.BytesIO(data))


def bundle_run(run_dir: str, namespace: str = "default", service: str = "demo-llm", output: Optional[str] = None, backend: Optional[ClusterBackend] = None, include_traces: bool = True, include_grafana: bool = True, include_sbom: bool = False, quiet: bool = False) -> str:
    run = Path(run_dir)
    if not run.is_dir():
        raise FileNotFoundError(run_dir)
    run_id = run.name
    output = output or str(Path("artifacts") / f"{run_id}.tar.gz")
    name = Path(output).name[: -len(".tar.gz")] if output.endswith(".tar.gz") else Path(output).stem
    backend = backend or get_backend()
    files: Dict[str, bytes] = {}

    for f in RUN_FILES:
        p = run / f
        if p.exists():
            files[f] = p.read_bytes()
    if "results.json" not in files and not quiet:
        print(f"WARNING: No results.json found in {run_dir}", file=sys.stderr)
    results = json.loads(files["results.json"]) if "results.json" in files else {}
    meta = json.loads(files["meta.json"]) if "meta.json" in files else {}

    # Every time stamp inside the archive is the RUN's (end of its measurement window, else the
    # newest artifact), not the moment of bundling: bundling the same run twice gives the same
    # bytes (the reference fixes the tar mtimes, tools/bundle_run.sh:333, but stamps `now` into
    # provenance.json, so its archives never compare equal).
    w_end = (results.get("window") or {}).get("end")
    stamp = float(w_end) if isinstance(w_end, (int, float)) else max((os.path.getmtime(run / f) for f in RUN_FILES if (run / f).exists()), default=float(FIXED_MTIME))
    stamp = float(int(stamp))
    facts = collect_cluster_facts(namespace, service, backend, at=stamp)
    files["cluster_facts.json"] = json.dumps(facts, indent=2, default=str).encode()
    prov = {
        "bundle": {"run_id": run_id, "created": dt.datetime.fromtimestamp(stamp, dt.timezone.utc).strftime("%Y-%m-%dT%H:%M:%SZ"), "tool_version": (facts.get("git") or {}).get("sha") or "unknown", "schema_version": "1.0"},
        "run": meta,
        "cluster": facts,
        "reproducibility": {
            "instructions": "Use pinned digests from cluster_facts.json pod_images and identical Kubernetes/KServe/ROCm versions",
            "verification": "Compare results.json metrics against baseline with ±10% tolerance",
        },
    }
    files["provenance.json"] = json.dumps(prov, indent=2, default=str).encode()

    root = repo_root()
    for cfg in ("isvc.yaml", "cost.yaml", "slo.json", "example-config.yaml"):
        p = root / cfg
        if p.exists():
            files[f"configs/{cfg}"] = p.read_bytes()
    try:
        isvc = backend.get_isvc(namespace, service)
        if isvc:
            files["configs/deployed_isvc.yaml"] = _yaml(isvc)
        pods = backend.get_pods(namespace, service)
        if pods:
            files["configs/actual_pods.yaml"] = _yaml({"apiVersion": "v1", "kind": "List", "items": pods})
    except Exception:
        pods = []

    if include_grafana:
        dash = root / "dashboards"
        names = []
        if dash.is_dir():
            for p in sorted(dash.glob("*.json")):
                files[f"grafana/{p.name}"] = p.read_bytes()
                names.append(p.name)
        w = results.get("window") or {}
        files["grafana/README.md"] = (
            "# Grafana Export Instructions\n\n1. Import the dashboards from the JSON files in this directory\n"
            f"2. Set the time range to: {w.get('start')} - {w.get('end')} (epoch seconds)\n3. Export panels as PNG and add them here for a complete audit trail\n\nDashboards:\n"
            + "".join(f"- {n}\n" for n in names)
        ).encode()
    if include_traces and (run / "traces").is_dir():
        for p in sorted((run / "traces").rglob("*")):
            if p.is_file():
                files[f"traces/{p.relative_to(run / 'traces')}"] = p.read_bytes()

    images = sorted({c.get("image") for p in pods for c in (p.get("spec") or {}).get("containers", []) if c.get("image")})
    files["supply-chain/images.txt"] = ("\n".join(images) + ("\n" if images else "")).encode()
    files["supply-chain/README.md"] = (
        "# Supply Chain Artifacts\n\nCommands:\n"
        f"- Generate SBOMs: kvmini sbom --namespace {namespace} --service {service} --out-dir sboms/\n"
        "- Sign images:    kvmini sign --images-file images.txt --key cosign.key\n- Verify:         cosign verify --key cosign.pub <image>\n"
    ).encode()
    if include_sbom and images:
        tmp = Path(output).parent / f".sbom_{run_id}"
        for w_ in generate_sboms(images, str(tmp)):
            files[f"supply-chain/sboms/{Path(w_).name}"] = Path(w_).read_bytes()
        shutil.rmtree(tmp, ignore_errors=True)
    files["SUMMARY.md"] = _summary_md(run_id, namespace, service, results, "fairness_summary.json" in files, at=stamp).encode()

    os.makedirs(os.path.dirname(output) or ".", exist_ok=True)
    raw = io.BytesIO()
    with tarfile.open(fileobj=raw, mode="w", format=tarfile.GNU_FORMAT) as tar:
        for fname in sorted(files):
            _add_bytes(tar, f"{name}/{fname}", files[fname])
    with open(output, "wb") as fo:
        with gzip.GzipFile(filename="", mode="wb", fileobj=fo, mtime=FIXED_MTIME) as gz:
            gz.write(raw.getvalue())
    import hashlib

    with open(output, "rb") as fi, open(output + ".sha256", "w") as fs:
        fs.write(f"{hashlib.sha256(fi.read()).hexdigest()}  {os.path.basename(output)}\n")
    if not quiet:
        print(f"Bundle created: {output} ({os.path.getsize(output) / 1024:.1f} KiB, {len(files)} files)")
    return output


def _yaml(obj: Any) -> bytes:
    import yaml

    return yaml.safe_dump(obj, sort_keys=False).encode()


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(prog="kvmini bundle", description="Bundle a run directory with provenance into a reproducible tar.gz")
    ap.add_argument("--run-dir", required=True)
    ap.add_argument("--namespace", default="default")
    ap.add_argument("--service", default="demo-llm")
    ap.add_argument("--output", default=None)
    ap.add_argument("--no-traces", action="store_true")
    ap.add_argument("--no-grafana", action="store_true")
    ap.add_argument("--sbom", action="store_true")
    ap.add_argument("--cluster-backend", default=None, choices=["auto", "kubectl", "local", "fake"])
    a = ap.parse_args(argv)
    try:
        bundle_run(a.run_dir, a.namespace, a.service, a.output, get_backend(a.cluster_backend), not a.no_traces, not a.no_grafana, a.sbom)
    except FileNotFoundError as e:
        print(f"ERROR: run dir not found: {e}", file=sys.stderr)
        return 1
    return 0


def main_facts(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(prog="kvmini cluster-facts")
    ap.add_argument("--namespace", default="default")
    ap.add_argument("--service", default="demo-llm")
    ap.add_argument("--output", default=None)
    ap.add_argument("--cluster-backend", default=None, choices=["auto", "kubectl", "local", "fake"])
    a = ap.parse_args(argv)
    facts = collect_cluster_facts(a.namespace, a.service, get_backend(a.cluster_backend))
    if a.output:
        write_json(a.output, facts)
    else:
        print(json.dumps(facts, indent=2, default=str))
    return 0


def main_sbom(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(prog="kvmini sbom")
    ap.add_argument("--namespace", default="default")
    ap.add_argument("--service", default="demo-llm")
    ap.add_argument("--out-dir", required=True)
    ap.add_argument("--cluster-backend", default=None, choices=["auto", "kubectl", "local", "fake"])
    a = ap.parse_args(argv)
    pods = get_backend(a.cluster_backend).get_pods(a.namespace, a.service)
    images = sorted({c.get("image") for p in pods for c in (p.get("spec") or {}).get("containers", []) if c.get("image")})
    for w in generate_sboms(images, a.out_dir):
        print(w)
    return 0


def main_sign(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(prog="kvmini sign")
    ap.add_argument("--images-file", required=True)
    ap.add_argument("--key", required=True)
    ap.add_argument("--dry-run", action="store_true")
    a = ap.parse_args(argv)
    images = [ln.strip() for ln in Path(a.images_file).read_text().splitlines() if ln.strip()]
    return sign_images(images, a.key, a.dry_run)


if __name__ == "__main__":
    sys.exit(main())

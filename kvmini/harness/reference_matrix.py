"""Reference-run matrix (``kvmini reference-matrix``).

Walks ``matrix.gpus × matrix.models × matrix.traffic_patterns`` from
``reference-matrix.yaml``: deploy with the GPU's node selector and GPU limit,
run the benchmark pipeline with the traffic pattern, validate against the
thresholds (error rate, minimum RPS, cold/warm p95 multiplier), move the run to
``artifacts.output_dir``, write a ``BOM.md`` bill of materials with a sha256 over
the run's files, bundle (and cosign-sign when available), then summarise into
``matrix_summary.json`` and a top-level ``results.json`` (reference
``scripts/reference_runner.py:60-112,177-391``).

``--local native|mock`` runs every cell against a local server so the matrix can
be exercised on a single MI355X box (or in CPU CI) without a cluster.
"""

from __future__ import annotations

import argparse
import hashlib
import os
import shutil
import subprocess
import sys
import time
from typing import Any, Dict, List, Optional

from kvmini.cluster import get_backend
from kvmini.deploy import manifests as mf
from kvmini.utils import load_json, load_yaml, write_json


def directory_checksum(run_dir: str) -> str:
    """sha256 over (relative path, content) of every file, in sorted order."""
    h = hashlib.sha256()
    for root, dirs, files in os.walk(run_dir):
        dirs.sort()
        for name in sorted(files):
            if name == "BOM.md":
                continue
            p = os.path.join(root, name)
            h.update(os.path.relpath(p, run_dir).encode())
            with open(p, "rb") as f:
                for chunk in iter(lambda: f.read(1 << 20), b""):
                    h.update(chunk)
    return h.hexdigest()


def _tool_version(cmd: List[str]) -> str:
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=10)
        return out.stdout.strip().splitlines()[0] if out.returncode == 0 and out.stdout.strip() else "unknown"
    except (OSError, subprocess.SubprocessError, IndexError):
        return "unknown"


class ReferenceRunner:
    def __init__(self, config: Dict[str, Any], output_dir: Optional[str] = None, local: Optional[str] = None, cluster_backend: Optional[str] = None, port: int = 8000, cost_file: str = "cost.yaml", namespace: str = "benchmark"):
        self.config = config
        self.output_dir = output_dir or config.get("artifacts", {}).get("output_dir", "artifacts/reference")
        self.local = local
        self.cluster_backend = "local" if local else cluster_backend
        self.cluster = get_backend(self.cluster_backend)
        self.port = port
        self.cost_file = cost_file
        self.namespace = namespace
        self.failures: List[Dict[str, Any]] = []
        os.makedirs(self.output_dir, exist_ok=True)

    @staticmethod
    def generate_run_id(gpu: str, model: str, traffic: str) -> str:
        return f"ref_{gpu}_{model}_{traffic}_{time.strftime('%Y%m%d_%H%M%S')}"

    # -- validation ---------------------------------------------------------------------
    def validate_results(self, results: Dict[str, Any]) -> List[str]:
        t = self.config["thresholds"]
        problems = []
        err = results.get("error_rate", 1.0)
        if err is None or err > t["error_rate_max"]:
            problems.append(f"error rate {err} exceeds {t['error_rate_max']}")
        rps = results.get("throughput_rps") or 0
        if rps < t["min_throughput_rps"]:
            problems.append(f"throughput {rps} below {t['min_throughput_rps']} RPS")
        cold, warm = results.get("cold_p95_ms") or 0, results.get("warm_p95_ms") or 0
        if warm > 0 and cold / warm > t["cold_multiplier_max"]:
            problems.append(f"cold multiplier {cold / warm:.2f} exceeds {t['cold_multiplier_max']}")
        return problems

    # -- BOM ------------------------------------------------------------------------------
    def create_bom(self, run_dir: str, gpu: Dict[str, Any], model: Dict[str, Any], traffic: Dict[str, Any], signed: str) -> str:
        t = self.config["thresholds"]
        facts = {}
        try:
            facts = self.cluster.cluster_facts()
        except Exception:
            pass
        k8s = _tool_version(["kubectl", "version", "--client", "-o", "yaml"]) if not self.local else "n/a (local server)"
        bom = f"""# Bill of Materials - Reference Run

**Generated**: {time.strftime("%Y-%m-%dT%H:%M:%S")}
**Run ID**: {os.path.basename(run_dir)}

## Configuration
- **GPU**: {gpu["name"]} (${gpu["cost_per_hour"]}/hour, limit {gpu.get("gpu_limit", 1)})
- **Model**: {model["name"]} ({model.get("uri", "random-init")})
- **Traffic**: {traffic["name"]} ({traffic["requests"]} req, {traffic["concurrency"]} concurrent, {traffic.get("max_tokens", 64)} max tokens)

## Environment
- **Cluster backend**: {self.cluster.name}
- **Kubernetes client**: {k8s}
- **Node selector**: {gpu.get("node_selector") or {}}
- **GPUs seen**: {[g.get("product") for g in facts.get("gpus", [])] if isinstance(facts.get("gpus"), list) else "unknown"}
- **ROCm**: {_tool_version(["cat", "/opt/rocm/.info/version"])}

## Thresholds
- P95 variance: +-{t["p95_variance_pct"]}%
- Error rate max: {t["error_rate_max"]}
- Cold multiplier max: {t["cold_multiplier_max"]}
- Min throughput: {t["min_throughput_rps"]} RPS

## Files Included
- `results.json` - consolidated metrics
- `requests.csv` - per-request data
- `meta.json` - run metadata
- `energy.json` / `power.json` - power and energy data (when sampled)
- `report.html` - visual report
- `BOM.md` - this bill of materials

## Reproducibility
```bash
kvmini bench \\
  --namespace {self.namespace} \\
  --service {model["name"]}-ref \\
  --model {model["name"]} \\
  --requests {traffic["requests"]} \\
  --concurrency {traffic["concurrency"]} \\
  --max-tokens {traffic.get("max_tokens", 64)}
```

## Verification
- Bundle signature: {signed}
- Checksum (sha256 over run files): {directory_checksum(run_dir)}
"""
        path = os.path.join(run_dir, "BOM.md")
        with open(path, "w") as f:
            f.write(bom)
        return path

    def sign_bundle(self, bundle_path: str) -> str:
        """cosign sign-blob when cosign and a key are available; otherwise record why not."""
        key = os.environ.get("COSIGN_KEY")
        if not shutil.which("cosign"):
            return "unsigned (cosign not installed)"
        if not key:
            return "unsigned (COSIGN_KEY not set)"
        sig = bundle_path + ".sig"
        try:
            subprocess.run(["cosign", "sign-blob", "--yes", "--key", key, "--output-signature", sig, bundle_path], check=True, capture_output=True, timeout=120)
            return f"cosign signature {os.path.basename(sig)}"
        except (OSError, subprocess.SubprocessError) as e:
            return f"unsigned (cosign failed: {e})"

    # -- one cell -------------------------------------------------------------------------
    def run_single_benchmark(self, gpu: Dict[str, Any], model: Dict[str, Any], traffic: Dict[str, Any]) -> Optional[str]:
        from kvmini.harness import bench
        from kvmini.harness.bundle import bundle_run

        run_id = self.generate_run_id(gpu["name"], model["name"], traffic["name"])
        service = f"{model['name']}-ref".replace(".", "-")
        tmp_dir = os.path.join("runs", run_id)
        final_dir = os.path.join(self.output_dir, run_id)
        print(f"Starting benchmark: {gpu['name']} + {model['name']} + {traffic['name']}")
        local_pid = None
        deployed = False
        try:
            url = None
            if self.local:
                from kvmini.deploy.cli import start_local_server

                ns = argparse.Namespace(local_server=self.local, port=self.port, model_name=model["name"] if self.local == "native" else None, service=service, namespace=self.namespace, profile=None, timeout=900.0)
                info = start_local_server(ns)
                url, local_pid = info["url"], info["pid"]
            else:
                manifest = mf.vllm_rocm_isvc(service, self.namespace, model_uri=model.get("uri"), model_name=model["name"])
                manifest = mf.set_gpu_limit(manifest, int(gpu.get("gpu_limit", 1)))
                if gpu.get("node_selector"):
                    manifest["spec"]["predictor"]["nodeSelector"] = dict(gpu["node_selector"])
                self.cluster.apply(manifest)
                deployed = True
                self.cluster.wait_ready(self.namespace, service, 900.0)
            argv = ["--namespace", self.namespace, "--service", service, "--model", model["name"], "--requests", str(traffic["requests"]), "--concurrency", str(traffic["concurrency"]), "--max-tokens", str(traffic.get("max_tokens", 64)), "--run-id", run_id, "--run-dir", tmp_dir, "--cost-file", self.cost_file, "--report"]
            if traffic.get("pattern"):
                argv += ["--pattern", traffic["pattern"]]
            if url:
                argv += ["--url", url]
            if self.cluster_backend:
                argv += ["--cluster-backend", self.cluster_backend]
            rc = bench.main(argv)
            res_path = os.path.join(tmp_dir, "results.json")
            if rc != 0 or not os.path.exists(res_path):
                raise RuntimeError(f"benchmark failed (exit {rc}); results not found: {res_path}")
            results = load_json(res_path)
            problems = self.validate_results(results)
            if problems:
                for p in problems:
                    print(f"  threshold violated: {p}", file=sys.stderr)
                self.failures.append({"run_id": run_id, "problems": problems})
                return None
            exp = gpu.get("expected_throughput_rps")
            if exp and (results.get("throughput_rps") or 0) < exp:
                print(f"  note: throughput {results.get('throughput_rps')} below the expected {exp} RPS for {gpu['name']}")
            if os.path.exists(final_dir):
                shutil.rmtree(final_dir)
            shutil.move(tmp_dir, final_dir)
            signed = "unsigned (signing disabled)"
            bundle_path = bundle_run(final_dir, self.namespace, service, output=final_dir.rstrip("/") + ".tar.gz", backend=self.cluster, quiet=True)
            if self.config.get("artifacts", {}).get("sign_bundles"):
                signed = self.sign_bundle(bundle_path)
            self.create_bom(final_dir, gpu, model, traffic, signed)
            print(f"Benchmark completed: {final_dir}")
            return final_dir
        except (Exception, SystemExit) as e:
            print(f"Benchmark failed: {e}", file=sys.stderr)
            self.failures.append({"run_id": run_id, "problems": [str(e)]})
            return None
        finally:
            if local_pid:
                import signal

                try:
                    os.kill(local_pid, signal.SIGTERM)
                except OSError:
                    pass
                from kvmini.cluster import LocalBackend

                LocalBackend().mark_finished(local_pid)
                time.sleep(0.3)
            elif deployed:
                try:
                    self.cluster.delete_isvc(self.namespace, service)
                except Exception:
                    pass

    # -- matrix -----------------------------------------------------------------------------
    def cells(self, gpu_filter: Optional[str] = None, model_filter: Optional[str] = None, traffic_filter: Optional[str] = None) -> List[tuple]:
        m = self.config["matrix"]
        return [
            (g, mo, t)
            for g in m["gpus"] if not gpu_filter or g["name"] == gpu_filter
            for mo in m["models"] if not model_filter or mo["name"] == model_filter
            for t in m["traffic_patterns"] if not traffic_filter or t["name"] == traffic_filter
        ]

    def run_matrix(self, gpu_filter: Optional[str] = None, model_filter: Optional[str] = None, traffic_filter: Optional[str] = None) -> List[str]:
        ok: List[str] = []
        cells = self.cells(gpu_filter, model_filter, traffic_filter)
        for g, mo, t in cells:
            r = self.run_single_benchmark(g, mo, t)
            if r:
                ok.append(r)
        print(f"Reference matrix completed: {len(ok)}/{len(cells)} successful")
        self.generate_summary(ok, len(cells))
        return ok

    def generate_summary(self, runs: List[str], total: int) -> None:
        summary: Dict[str, Any] = {"generated_at": time.strftime("%Y-%m-%dT%H:%M:%S"), "total_runs": len(runs), "attempted": total, "failed": self.failures, "runs": []}
        for d in runs:
            p = os.path.join(d, "results.json")
            if os.path.exists(p):
                r = load_json(p)
                summary["runs"].append({"run_id": os.path.basename(d), "p95_ms": r.get("p95_ms"), "throughput_rps": r.get("throughput_rps"), "error_rate": r.get("error_rate"), "cost_per_1k_tokens": r.get("cost_per_1k_tokens"), "energy_wh_per_1k_tokens": r.get("energy_wh_per_1k_tokens")})
        write_json(os.path.join(self.output_dir, "matrix_summary.json"), summary)
        write_json(os.path.join(self.output_dir, "results.json"), {"generated_at": summary["generated_at"], "successful_runs": len(runs), "has_results": len(runs) > 0})
        print(f"Matrix summary written to {os.path.join(self.output_dir, 'matrix_summary.json')} and results.json")


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(prog="kvmini reference-matrix", description="Run the reference benchmark matrix")
    ap.add_argument("--config", default="reference-matrix.yaml")
    ap.add_argument("--output-dir", default=None)
    ap.add_argument("--gpu", default=None, help="filter to one GPU type")
    ap.add_argument("--model", default=None, help="filter to one model")
    ap.add_argument("--traffic", default=None, help="filter to one traffic pattern")
    ap.add_argument("--dry-run", action="store_true", help="show what would be run")
    ap.add_argument("--local", default=None, choices=["native", "mock"])
    ap.add_argument("--port", type=int, default=8000)
    ap.add_argument("--cluster-backend", default=None, choices=["auto", "kubectl", "local", "fake"])
    ap.add_argument("--cost-file", default="cost.yaml")
    ap.add_argument("--namespace", default="benchmark")
    a = ap.parse_args(argv)
    if not os.path.exists(a.config):
        print(f"Config not found: {a.config}", file=sys.stderr)
        return 1
    runner = ReferenceRunner(load_yaml(a.config), a.output_dir, a.local, a.cluster_backend, a.port, a.cost_file, a.namespace)
    if a.dry_run:
        cells = runner.cells(a.gpu, a.model, a.traffic)
        print(f"Would run {len(cells)} configurations:")
        for g, mo, t in cells:
            print(f"  - {g['name']} + {mo['name']} + {t['name']} ({t['requests']} req @ c{t['concurrency']})")
        return 0
    ok = runner.run_matrix(a.gpu, a.model, a.traffic)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())

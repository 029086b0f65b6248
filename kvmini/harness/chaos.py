"""Chaos & resilience harness (``kvmini chaos``).

Runs a fault-free baseline and then five controlled faults — GPU device-plugin
restart, predictor pod delete, container kill, network degrade (10 % loss +
50 ms delay while the bench runs), node drain — and records for each the MTTR
(seconds until the service is Ready again), the p95 of a benchmark run under /
right after the fault and the SLO-gate verdict into ``resilience_table.json``
(rows ``{fault, mttr_s, p95_ms, slo_gate, run_dir}``; reference
``tools/chaos_harness.sh:85-240``).

Faults go through the cluster backend, so the same harness drives a KServe
cluster (kubectl), a local native server (process kill / stall) and — with
``--admin-url`` — the mock server's ``/admin/chaos`` endpoint (latency + error
injection), which is what CPU-only CI exercises.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
import urllib.request
from typing import Any, Dict, List, Optional

from kvmini.cluster import get_backend
from kvmini.gates import slo as gate
from kvmini.utils import load_json, write_json

FAULTS = ["device-plugin-restart", "pod-delete", "container-kill", "network-degrade", "node-drain"]
# faults that stay in force while the benchmark runs and are removed afterwards
HELD_FAULTS = {"network-degrade"}

# mock-server analogue of each fault (POST /admin/chaos payload while active)
ADMIN_FAULTS: Dict[str, Dict[str, Any]] = {
    "device-plugin-restart": {"extra_latency_ms": 20.0},
    "pod-delete": {"restart": True},
    "container-kill": {"restart": True},
    "network-degrade": {"extra_latency_ms": 50.0, "error_rate": 0.1},
    "node-drain": {"restart": True, "extra_latency_ms": 10.0},
}
ADMIN_CLEAR = {"extra_latency_ms": 0.0, "error_rate": 0.0}


def _admin_post(admin_url: str, payload: Dict[str, Any]) -> None:
    req = urllib.request.Request(admin_url.rstrip("/") + "/admin/chaos", data=json.dumps(payload).encode(), headers={"Content-Type": "application/json"})
    with urllib.request.urlopen(req, timeout=5) as r:
        r.read()


class ChaosRunner:
    def __init__(self, a: argparse.Namespace):
        self.a = a
        self.backend = get_backend(a.cluster_backend)
        self.ts = time.strftime("%Y%m%d_%H%M%S")
        self.rows: List[Dict[str, Any]] = []
        self.slo = load_json(a.slo) if a.slo and os.path.exists(a.slo) else None

    # -- pieces -----------------------------------------------------------------
    def inject(self, fault: str) -> Optional[str]:
        """Returns an error string when the fault could not be injected."""
        try:
            if self.a.admin_url:
                payload = dict(ADMIN_FAULTS[fault])
                if payload.get("restart") and self.a.admin_down_s > 0:
                    payload["down_s"] = self.a.admin_down_s  # the mock stays away this long: what MTTR then measures
                _admin_post(self.a.admin_url, payload)
            else:
                self.backend.inject_fault(fault, self.a.namespace, self.a.service)
            return None
        except Exception as e:  # injection is best-effort, like `| true` in a shell harness
            return f"{type(e).__name__}: {e}"

    def clear(self, fault: str) -> None:
        try:
            if self.a.admin_url:
                _admin_post(self.a.admin_url, ADMIN_CLEAR)
            else:
                self.backend.clear_fault(fault, self.a.namespace, self.a.service)
        except Exception:
            pass

    def wait_ready(self) -> Optional[float]:
        """Seconds until the service answers again (None: not within ``--ready-timeout``).  With an
        admin URL the server's ``/health`` is polled until it returns 200: a server that is down
        refuses the connection or answers 503 at once, so one request would measure nothing."""
        t0 = time.time()
        try:
            if self.a.admin_url:
                deadline = t0 + float(self.a.ready_timeout)
                url = self.a.admin_url.rstrip("/") + "/health"
                while True:
                    try:
                        with urllib.request.urlopen(url, timeout=max(0.2, min(2.0, deadline - time.time()))) as r:
                            r.read()
                            if r.status == 200:
                                break
                    except Exception:
                        pass
                    if time.time() >= deadline:
                        return None
                    time.sleep(0.05)
            else:
                self.backend.wait_ready(self.a.namespace, self.a.service, self.a.ready_timeout)
        except Exception:
            return None
        return round(time.time() - t0, 3)

    def run_bench(self, tag: str) -> str:
        from kvmini.harness import bench

        run_id = f"chaos_{self.a.service}_{tag}_{self.ts}"
        argv = ["--namespace", self.a.namespace, "--service", self.a.service, "--requests", str(self.a.requests), "--concurrency", str(self.a.concurrency), "--max-tokens", str(self.a.max_tokens), "--run-id", run_id, "--cost-file", self.a.cost_file, "--no-energy"]
        if self.a.url:
            argv += ["--url", self.a.url]
        if self.a.model:
            argv += ["--model", self.a.model]
        if self.a.cluster_backend:
            argv += ["--cluster-backend", self.a.cluster_backend]
        if self.a.profile:
            argv += ["--profile", self.a.profile]
        if self.a.api_key:
            argv += ["--api-key", self.a.api_key]
        if self.a.runs_root:
            argv += ["--run-dir", os.path.join(self.a.runs_root, run_id)]
        try:
            bench.main(argv)
        except SystemExit:
            pass
        return os.path.join(self.a.runs_root or "runs", run_id)

    def record(self, fault: str, mttr: Optional[float], run_dir: str, note: Optional[str] = None) -> Dict[str, Any]:
        res_path = os.path.join(run_dir, "results.json")
        res = load_json(res_path) if os.path.exists(res_path) else {}
        verdict = "unknown"
        if self.slo is not None and res:
            _s, failures = gate.evaluate(self.slo, res, None, None)
            verdict = "fail" if failures else "pass"
        row: Dict[str, Any] = {"fault": fault, "mttr_s": mttr, "p95_ms": res.get("p95_ms"), "error_rate": res.get("error_rate"), "slo_gate": verdict, "run_dir": run_dir}
        if note:
            row["note"] = note
        self.rows.append(row)
        print(f"  {fault}: mttr={mttr} p95={row['p95_ms']} gate={verdict}" + (f" ({note})" if note else ""))
        return row

    # -- scenarios --------------------------------------------------------------
    def baseline(self) -> None:
        print("=== baseline ===")
        self.record("baseline", 0.0, self.run_bench("baseline"))

    def fault(self, fault: str) -> None:
        print(f"=== {fault} ===")
        tag = fault.replace("-", "_")
        err = self.inject(fault)
        if err:
            self.record(fault, None, "", note=f"not injected: {err}")
            return
        if fault in HELD_FAULTS:
            try:
                run_dir = self.run_bench(tag)
            finally:
                self.clear(fault)
            self.record(fault, self.wait_ready(), run_dir)
            return
        mttr = self.wait_ready()
        run_dir = self.run_bench(tag)
        self.clear(fault)
        self.record(fault, mttr, run_dir)

    def run(self) -> Dict[str, Any]:
        if not self.a.skip_baseline:
            self.baseline()
        for f in self.a.faults:
            self.fault(f)
        out = self.a.out or os.path.join(self.a.runs_root or "runs", f"chaos_{self.a.service}_{self.ts}", "resilience_table.json")
        doc = {"service": self.a.service, "namespace": self.a.namespace, "generated_at": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()), "backend": "admin" if self.a.admin_url else self.backend.name, "results": self.rows}
        write_json(out, doc)
        print(f"Resilience table: {out}")
        return doc


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(prog="kvmini chaos", description="Fault injection: MTTR, p95 under fault, SLO verdict per fault")
    ap.add_argument("--namespace", default="ml-prod")
    ap.add_argument("--service", default="demo-llm")
    ap.add_argument("--url", default=None)
    ap.add_argument("--model", default=None)
    ap.add_argument("--requests", type=int, default=200)
    ap.add_argument("--concurrency", type=int, default=10)
    ap.add_argument("--max-tokens", type=int, default=64)
    ap.add_argument("--profile", default=None)
    ap.add_argument("--slo", default="slo.json")
    ap.add_argument("--cost-file", default="cost.yaml")
    ap.add_argument("--api-key", default=None)
    ap.add_argument("--faults", nargs="+", default=FAULTS, choices=FAULTS)
    ap.add_argument("--skip-baseline", action="store_true")
    ap.add_argument("--ready-timeout", type=float, default=600.0)
    ap.add_argument("--cluster-backend", default=None, choices=["auto", "kubectl", "local", "fake"])
    ap.add_argument("--admin-url", default=None, help="inject faults through a mock server's /admin/chaos instead of the cluster")
    ap.add_argument("--admin-down-s", type=float, default=0.0, help="with --admin-url: seconds a restart fault keeps the mock server answering 503")
    ap.add_argument("--runs-root", default=None, help="directory for the per-fault run dirs (default runs/)")
    ap.add_argument("--out", default=None, help="resilience_table.json path")
    ap.add_argument("--fail-on-gate", action="store_true", help="exit 3 when any fault run fails the SLO gate")
    a = ap.parse_args(argv)
    doc = ChaosRunner(a).run()
    if a.fail_on_gate and any(r["slo_gate"] == "fail" for r in doc["results"]):
        return 3
    return 0


if __name__ == "__main__":
    sys.exit(main())

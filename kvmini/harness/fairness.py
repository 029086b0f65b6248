"""Dual-tenant fairness + backpressure harness (``kvmini fairness``).

Two tenants (A, B) drive the same endpoint concurrently through the shared load
engine (``X-Tenant`` header on every request).  A guard watches the rolling p95
(window 50, floor-index quantile — reference ``scripts/fairness_dual_tenant.py:46-65``)
of all successful latencies; while it is above ``--p95-budget-ms`` tenant B is
throttled for ``--guard-cooldown`` seconds: its requests are recorded as HTTP 429
``throttled`` without touching the network (``:110-122``).  Outputs:
``requests.csv`` (``id,tenant,start_ms,latency_ms,status,error,guard_action``),
``fairness_summary.json`` (``summary{A,B,throughput_share}``, ``guard_triggers``,
plus ``guard_activations`` — edge-triggered count; the reference counts every
over-budget observation), ``fairness_report.html``; ``--slo`` runs the fairness
section of the SLO gate.
"""

from __future__ import annotations

import argparse
import asyncio
import csv
import html
import json
import os
import sys
import time
from typing import Any, Dict, List, Optional

from kvmini.analysis.quantiles import RollingQuantile, percentiles
from kvmini.loadgen.httpclient import HttpClient
from kvmini.loadgen.protocols import RequestSpec
from kvmini.loadgen.runner import LoadConfig, ReqResult, run_load
from kvmini.report.generator import CSS, svg_bar_chart
from kvmini.utils import write_json


class Guard:
    def __init__(self, p95_budget_ms: Optional[float], window: int = 50, cooldown_sec: float = 2.0):
        self.p95_budget_ms = p95_budget_ms
        self.rolling = RollingQuantile(window, 0.95, "floor")
        self.cooldown_sec = cooldown_sec
        self._throttle_until = 0.0
        self.trigger_count = 0  # every over-budget observation
        self.activations = 0  # off -> on edges

    def observe(self, latency_ms: float) -> None:
        self.rolling.add(latency_ms)
        if self.p95_budget_ms:
            p = self.rolling.value()
            if p and p > self.p95_budget_ms:
                self.trigger_count += 1
                now = time.time()
                if now >= self._throttle_until:
                    self.activations += 1
                self._throttle_until = now + self.cooldown_sec

    def should_throttle_b(self) -> bool:
        return time.time() < self._throttle_until


def summarize(results: List[ReqResult]) -> Dict[str, Any]:
    def per(name: str) -> Dict[str, Any]:
        rs = [r for r in results if r.tenant == name]
        lats = [r.latency_ms for r in rs if r.status == 200]
        p50 = p95 = None
        if lats:
            p50, p95 = percentiles(lats, (0.5, 0.95))
        return {"requests": len(rs), "success": len(lats), "throttled": sum(1 for r in rs if r.status == 429), "p50_ms": p50, "p95_ms": p95}

    a, b = per("A"), per("B")
    tot = a["success"] + b["success"]
    return {"A": a, "B": b, "throughput_share": {"A": a["success"] / tot if tot else None, "B": b["success"] / tot if tot else None}}


def write_csv(results: List[ReqResult], path: str) -> None:
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["id", "tenant", "start_ms", "latency_ms", "status", "error", "guard_action"])
        for r in results:
            w.writerow([r.id, r.tenant, f"{r.start_ms:.3f}", f"{r.latency_ms:.3f}" if r.status == 200 else "", r.status or "", r.error or "", "throttled" if r.status == 429 and r.error == "throttled" else ""])


def write_report(summary: Dict[str, Any], guard: Guard, path: str, gate_status: Optional[str]) -> None:
    a, b = summary["A"], summary["B"]
    sh = summary["throughput_share"]
    doc = [
        f"<!DOCTYPE html><html><head><meta charset='utf-8'><title>Fairness report</title><style>{CSS}</style></head><body><h1>Dual-tenant fairness report</h1>",
        f"<p>Guard triggers: <b>{guard.trigger_count}</b> (activations: {guard.activations}) · SLO gate: <b>{html.escape(gate_status or 'not run')}</b></p>",
        svg_bar_chart(["A", "B"], {"p50 (ms)": [a["p50_ms"], b["p50_ms"]], "p95 (ms)": [a["p95_ms"], b["p95_ms"]]}, "Latency by tenant"),
        svg_bar_chart(["A", "B"], {"throughput share": [sh["A"], sh["B"]]}, "Throughput share"),
        "<table><tr><th>tenant</th><th>requests</th><th>success</th><th>throttled</th><th>p50 ms</th><th>p95 ms</th></tr>"
        + "".join(f"<tr><td>{t}</td><td>{d['requests']}</td><td>{d['success']}</td><td>{d['throttled']}</td><td>{d['p50_ms'] and round(d['p50_ms'], 1)}</td><td>{d['p95_ms'] and round(d['p95_ms'], 1)}</td></tr>" for t, d in (("A", a), ("B", b)))
        + "</table></body></html>",
    ]
    with open(path, "w") as f:
        f.write("\n".join(doc))


async def run_fairness(a: argparse.Namespace) -> Dict[str, Any]:
    guard = Guard(a.p95_budget_ms, a.guard_window, a.guard_cooldown)
    client = HttpClient(pooled=True, verify_tls=not a.insecure)

    def on_result(r: ReqResult) -> None:
        if r.status == 200:
            guard.observe(r.latency_ms)

    def cfg(tenant: str, n: int, conc: int) -> LoadConfig:
        return LoadConfig(
            url=a.url, requests=n, concurrency=conc, pattern="steady", target_rps=1e6, spec=RequestSpec(model=a.model, prompt=a.prompt, max_tokens=a.max_tokens, stream=False),
            api_key=a.api_key, insecure=a.insecure, extra_headers={"X-Tenant": tenant}, trace=False, token_events=False, quiet=True, on_result=on_result,
            admit=(lambda _rid: not guard.should_throttle_b()) if tenant == "B" else None,
        )

    try:
        ra, rb = await asyncio.gather(run_load(cfg("A", a.tenant_a_requests, a.tenant_a_concurrency), client=client), run_load(cfg("B", a.tenant_b_requests, a.tenant_b_concurrency), client=client))
    finally:
        await client.aclose()
    results = ra + rb
    os.makedirs(a.run_dir, exist_ok=True)
    write_csv(results, os.path.join(a.run_dir, "requests.csv"))
    summary = summarize(results)
    payload: Dict[str, Any] = {"summary": summary, "guard_triggers": guard.trigger_count, "guard_activations": guard.activations}
    path = os.path.join(a.run_dir, "fairness_summary.json")
    write_json(path, payload)
    gate_status = None
    if a.slo:
        from kvmini.gates import slo as gate

        sl = json.load(open(a.slo))
        _s, failures = gate.evaluate(sl, {}, None, payload)
        gate_status = "fail" if failures else "pass"
        payload["slo_gate"] = gate_status
        payload["budgets"] = sl.get("fairness", {})
        write_json(path, payload)
    write_report(summary, guard, os.path.join(a.run_dir, "fairness_report.html"), gate_status)
    return payload


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(prog="kvmini fairness", description="Dual-tenant fairness + backpressure harness")
    ap.add_argument("--url", required=True)
    ap.add_argument("--model", required=True)
    ap.add_argument("--prompt", default="Respond briefly to: Hello")
    ap.add_argument("--max-tokens", type=int, default=32)
    ap.add_argument("--tenant-a-requests", type=int, default=200)
    ap.add_argument("--tenant-b-requests", type=int, default=200)
    ap.add_argument("--tenant-a-concurrency", type=int, default=10)
    ap.add_argument("--tenant-b-concurrency", type=int, default=10)
    ap.add_argument("--p95-budget-ms", type=float, default=None)
    ap.add_argument("--guard-window", type=int, default=50)
    ap.add_argument("--guard-cooldown", type=float, default=2.0)
    ap.add_argument("--run-dir", required=True)
    ap.add_argument("--api-key", default=None)
    ap.add_argument("--insecure", action="store_true")
    ap.add_argument("--slo", default=None, help="SLO JSON (its fairness section is gated)")
    a = ap.parse_args(argv)
    payload = asyncio.run(run_fairness(a))
    print(json.dumps(payload, indent=2))
    return 3 if payload.get("slo_gate") == "fail" else 0


if __name__ == "__main__":
    sys.exit(main())

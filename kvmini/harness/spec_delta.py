"""Speculative-decoding delta report (``kvmini spec-delta``).

The reference ships speculation only as a profile whose description promises a
"20-40 % TTFT improvement" (``runners/profiles/speculative-decoding.yaml:27``,
``README.md:118``) — nothing in it measures that claim.  This harness does: it serves
the same model twice on a local native server, once plain and once with the profile's
draft model, drives the profile's request mix through the load engine against each, and
writes ``spec_delta.json`` + ``spec_delta.html``: time to first byte / first content
token, inter-token latency, end-to-end latency, RPS and tokens/s side by side with
relative deltas, plus the draft acceptance rate scraped from the server's ``/metrics``.

Draft choices (``--draft``): a preset name (independent small model), or ``self:<K>``
(early exit after K layers of the served model).  With random-init weights no draft can
agree with the target, so ``--tail-damping`` builds synthetic weights whose tail layers
perturb the residual stream only slightly; the achieved acceptance rate is reported
next to every number, and a run against real checkpoints needs no such knob.
"""

from __future__ import annotations

import argparse
import html
import os
import re
import signal
import sys
import time
import urllib.request
from typing import Any, Dict, List, Optional

from kvmini.analysis.reduce import analyze_run
from kvmini.cluster import LocalBackend
from kvmini.config.profiles import load_profile
from kvmini.loadgen.protocols import RequestSpec
from kvmini.loadgen.runner import LoadConfig, run_load_sync
from kvmini.utils import write_json

# (key, label, lower_is_better)
METRICS = [
    ("ttfb_p50_ms", "time to first byte p50 (ms)", True),
    ("time_to_first_token_ms", "time to first token, mean (ms)", True),
    ("ttft_p95_ms", "time to first token p95 (ms)", True),
    ("itl_p50_ms", "inter-token latency p50 (ms)", True),
    ("itl_p95_ms", "inter-token latency p95 (ms)", True),
    ("per_token_p50_ms", "generation time per token p50 (ms)", True),
    ("p50_ms", "request latency p50 (ms)", True),
    ("p95_ms", "request latency p95 (ms)", True),
    ("throughput_rps", "throughput (requests/s)", False),
    ("tokens_per_sec", "tokens/s", False),
]


def scrape_counters(url: str) -> Dict[str, float]:
    out: Dict[str, float] = {}
    try:
        with urllib.request.urlopen(url + "/metrics", timeout=5) as r:
            for line in r.read().decode().splitlines():
                m = re.match(r"^(kvmini_spec_tokens_(?:proposed|accepted)_total|vllm:generation_tokens_total)\s+([0-9.eE+-]+)", line)
                if m:
                    out[m.group(1)] = float(m.group(2))
    except Exception:
        pass
    return out


def run_variant(name: str, a: argparse.Namespace, serve_args: List[str], out_dir: str) -> Dict[str, Any]:
    from kvmini.deploy.cli import start_local_server

    service = f"spec-{name}"
    ns = argparse.Namespace(local_server=a.local, port=a.port, model_name=a.model if a.local == "native" else None, service=service, namespace="local", profile=None, timeout=900.0, serve_args=serve_args)
    info = start_local_server(ns)
    run_dir = os.path.join(out_dir, name)
    try:
        spec = RequestSpec(model=a.model, max_tokens=a.max_tokens, stream=True)
        # one untimed pass so both variants are measured warm
        run_load_sync(LoadConfig(url=info["url"], requests=max(4, a.concurrency), concurrency=a.concurrency, spec=spec, quiet=True, trace=False, token_events=False))
        before = scrape_counters(info["url"])
        run_load_sync(LoadConfig(url=info["url"], run_dir=run_dir, requests=a.requests, concurrency=a.concurrency, pattern=a.pattern, target_rps=a.target_rps, spec=spec, quiet=True, seed=0))
        after = scrape_counters(info["url"])
        res = analyze_run(run_dir, "local", service, backend=LocalBackend(), quiet=True)
        write_json(os.path.join(run_dir, "results.json"), res)
    finally:
        try:
            os.kill(info["pid"], signal.SIGTERM)
        except OSError:
            pass
        LocalBackend().mark_finished(info["pid"])
        time.sleep(0.5)
    tt = res.get("token_timing_metrics") or res
    row = {k: res.get(k, tt.get(k)) for k, _, _ in METRICS}
    proposed = after.get("kvmini_spec_tokens_proposed_total", 0.0) - before.get("kvmini_spec_tokens_proposed_total", 0.0)
    accepted = after.get("kvmini_spec_tokens_accepted_total", 0.0) - before.get("kvmini_spec_tokens_accepted_total", 0.0)
    row.update({"variant": name, "ready_s": info["ready_s"], "spec_proposed": proposed, "spec_accepted": accepted, "acceptance_rate": accepted / proposed if proposed else None, "error_rate": res.get("error_rate"), "run_dir": run_dir})
    return row


def delta_table(base: Dict[str, Any], spec: Dict[str, Any]) -> List[Dict[str, Any]]:
    rows = []
    for key, label, lower in METRICS:
        b, s = base.get(key), spec.get(key)
        rel = None
        if isinstance(b, (int, float)) and isinstance(s, (int, float)) and b:
            rel = (s - b) / b
        better = None if rel is None else (rel < 0) == lower if rel != 0 else None
        rows.append({"metric": key, "label": label, "baseline": b, "speculative": s, "delta_pct": None if rel is None else 100.0 * rel, "improved": better})
    return rows


def write_html(path: str, doc: Dict[str, Any]) -> None:
    def fmt(v: Any) -> str:
        return "n/a" if v is None else f"{v:,.3f}" if isinstance(v, float) else html.escape(str(v))

    trs = []
    for r in doc["deltas"]:
        cls = "" if r["improved"] is None else ("good" if r["improved"] else "bad")
        d = "n/a" if r["delta_pct"] is None else f"{r['delta_pct']:+.1f} %"
        trs.append(f"<tr><td>{html.escape(r['label'])}</td><td>{fmt(r['baseline'])}</td><td>{fmt(r['speculative'])}</td><td class='{cls}'>{d}</td></tr>")
    acc = doc["speculative"].get("acceptance_rate")
    body = f"""<!doctype html><html><head><meta charset="utf-8"><title>Speculative decoding delta</title>
<style>body{{font-family:system-ui,sans-serif;margin:2em;max-width:60em}}table{{border-collapse:collapse}}td,th{{border:1px solid #ccc;padding:.35em .7em;text-align:right}}
td:first-child,th:first-child{{text-align:left}}.good{{color:#0a7a2f;font-weight:600}}.bad{{color:#b00020;font-weight:600}}</style></head><body>
<h1>Speculative decoding: TTFT / inter-token delta</h1>
<p>Model <b>{html.escape(doc['model'])}</b>, draft <b>{html.escape(str(doc['draft']))}</b>, {doc['num_speculative_tokens']} draft tokens per step;
{doc['requests']} requests at concurrency {doc['concurrency']}, {doc['max_tokens']} output tokens, streaming.</p>
<p>Draft acceptance rate: <b>{'n/a' if acc is None else f'{100 * acc:.1f} %'}</b>
({int(doc['speculative'].get('spec_accepted') or 0)} of {int(doc['speculative'].get('spec_proposed') or 0)} proposed tokens).
{html.escape(doc.get('weights_note', ''))}</p>
<table><tr><th>metric</th><th>baseline</th><th>speculative</th><th>delta</th></tr>{''.join(trs)}</table>
</body></html>"""
    with open(path, "w") as f:
        f.write(body)


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(prog="kvmini spec-delta", description="Serve a model with and without speculative decoding and report the TTFT / inter-token deltas")
    ap.add_argument("--model", default="llama-3-8b")
    ap.add_argument("--profile", default="speculative-decoding")
    ap.add_argument("--draft", default=None, help="draft preset or self:<K> (default: the profile's native_engine.draft_model)")
    ap.add_argument("--num-speculative-tokens", type=int, default=None)
    ap.add_argument("--tail-damping", type=float, default=0.0, help="self:<K> + random-init: synthetic agreement knob (see kvmini serve --help)")
    ap.add_argument("--requests", type=int, default=None)
    ap.add_argument("--concurrency", type=int, default=None)
    ap.add_argument("--max-tokens", type=int, default=None)
    ap.add_argument("--pattern", default=None)
    ap.add_argument("--target-rps", type=float, default=0.0, help="0 = saturate at --concurrency")
    ap.add_argument("--local", default="native", choices=["native", "mock"])
    ap.add_argument("--port", type=int, default=8000)
    ap.add_argument("--serve-args", default="", help="extra flags for both servers, e.g. '--max-num-seqs 16'")
    ap.add_argument("--output-dir", default="runs/spec_delta")
    a = ap.parse_args(argv)

    prof = load_profile(a.profile)
    ne = prof.native_engine or {}
    a.draft = a.draft or ne.get("draft_model")
    a.num_speculative_tokens = a.num_speculative_tokens or int(ne.get("num_speculative_tokens", 4))
    a.requests = a.requests or prof.requests
    a.concurrency = a.concurrency or prof.concurrency
    a.max_tokens = a.max_tokens or prof.max_tokens
    a.pattern = a.pattern or prof.pattern
    a.target_rps = a.target_rps if a.target_rps > 0 else 1e6
    if not a.draft:
        print("no draft model: pass --draft or use a profile with native_engine.draft_model", file=sys.stderr)
        return 2
    os.makedirs(a.output_dir, exist_ok=True)
    extra = a.serve_args.split()
    damp = ["--tail-damping", str(a.tail_damping)] if a.tail_damping > 0 else []
    common = extra + (["--max-num-seqs", str(max(a.concurrency, 8))] if a.local == "native" and "--max-num-seqs" not in extra else [])
    if a.local == "native" and damp:
        # the baseline must serve the SAME weights: the damping is keyed on the draft depth
        base_args = common + ["--speculative-model", a.draft, "--num-speculative-tokens", "0"] + damp
    else:
        base_args = list(common)
    spec_args = common + (["--speculative-model", a.draft, "--num-speculative-tokens", str(a.num_speculative_tokens)] + damp if a.local == "native" else [])

    print(f"=== baseline ({a.model}) ===", flush=True)
    base = run_variant("baseline", a, base_args, a.output_dir)
    print(f"=== speculative ({a.model} + {a.draft}, k={a.num_speculative_tokens}) ===", flush=True)
    spec = run_variant("speculative", a, spec_args, a.output_dir)
    doc = {
        "model": a.model, "draft": a.draft, "num_speculative_tokens": a.num_speculative_tokens, "tail_damping": a.tail_damping,
        "requests": a.requests, "concurrency": a.concurrency, "max_tokens": a.max_tokens, "pattern": a.pattern,
        "baseline": base, "speculative": spec, "deltas": delta_table(base, spec),
        "weights_note": ("Random-init weights with the tail layers damped by %g (synthetic agreement between the early-exit draft and the full model)." % a.tail_damping) if a.tail_damping > 0
        else "Random-init weights: an independent draft cannot agree with the target, so this run bounds the overhead of speculation, not its benefit.",
    }
    write_json(os.path.join(a.output_dir, "spec_delta.json"), doc)
    write_html(os.path.join(a.output_dir, "spec_delta.html"), doc)
    for r in doc["deltas"]:
        d = "n/a" if r["delta_pct"] is None else f"{r['delta_pct']:+.1f}%"
        print(f"  {r['label']:<40} {str(r['baseline'])[:10]:>12} -> {str(r['speculative'])[:10]:>12}  {d}")
    acc = spec.get("acceptance_rate")
    print(f"  acceptance rate: {'n/a' if acc is None else f'{100 * acc:.1f}%'}")
    print(f"Report: {os.path.join(a.output_dir, 'spec_delta.html')}")
    return 0


if __name__ == "__main__":
    sys.exit(main())

"""Prompt-cache probe (``kvmini cache-probe``).

Two deterministic prompt sets (seed 42) are driven through the load engine at
concurrency 10 / 64 max tokens: *repeat-80* (20 % distinct prompts, the rest
re-draws of them, shuffled) and *unique-100* (all distinct).  The hit ratio is
inferred from the TTFT gap, ``min(1, ((ttft_unique - ttft_repeat) / ttft_unique) / 0.8)``,
with a pooled-std t statistic for confidence — the heuristic and the
``cache_analysis.json`` layout of the reference's ``cache-probe.sh:75-135,229-330``.

Additions: when the server exposes vLLM-style prefix-cache counters on
``/metrics`` (the built-in native and mock servers do) the measured hit ratio of
each experiment is recorded next to the inferred one, and first-content-token
TTFT is preferred over first-byte when the run recorded it.
"""

from __future__ import annotations

import argparse
import asyncio
import json
import math
import os
import random
import statistics
import sys
import time
import urllib.request
from typing import Any, Dict, List, Optional, Tuple

from kvmini.loadgen.protocols import RequestSpec
from kvmini.loadgen.runner import LoadConfig, ReqResult, run_load
from kvmini.utils import write_json

CAVEATS = [
    "Hit ratio is inferred from TTFT differences, not direct server metrics",
    "Assumes cache hits primarily affect prompt processing time",
    "Results may vary with model size, hardware, and engine configuration",
    "Small sample sizes may reduce statistical confidence",
]


def generate_prompt_sets(base_requests: int, seed: int = 42, pad_tokens: int = 0) -> Tuple[List[str], List[str]]:
    """(repeat_80, unique_100).  ``pad_tokens`` prepends a per-prompt
    deterministic filler so the shared prefix spans whole KV blocks (a 16-token
    block cannot be reused by a 12-token question)."""
    rng = random.Random(seed)
    n_unique = max(1, base_requests // 5)
    concepts = ["recursion", "inheritance", "polymorphism", "encapsulation"]
    processes = ["photosynthesis", "mitosis", "meiosis", "evolution"]

    def pad(tag: str) -> str:
        if pad_tokens <= 0:
            return ""
        r = random.Random(f"{seed}:{tag}")
        return " ".join(f"ctx{r.randrange(100000)}" for _ in range(pad_tokens)) + " "

    base: List[str] = []
    for i in range(n_unique):
        kinds = [
            f"What are the main features of Python version 3.{i % 12}?",
            f"Explain the concept of {concepts[i % 4]} in programming.",
            f"Write a brief summary of the year {2020 + (i % 5)}.",
            f"List {3 + (i % 7)} benefits of using cloud computing.",
            f"Describe the process of {processes[i % 4]}.",
        ]
        base.append(pad(f"b{i}") + kinds[i % len(kinds)])
    repeat = [base[i] if i < n_unique else rng.choice(base) for i in range(base_requests)]
    rng.shuffle(repeat)
    unique = [pad(f"u{i}") + f"Generate a unique response about topic #{i:04d}: What is the significance of the number {1000 + i} in mathematics and science?" for i in range(base_requests)]
    return repeat, unique


def ttft_stats(results: List[ReqResult]) -> Optional[Dict[str, Any]]:
    vals = []
    for r in results:
        if r.status == 200:
            v = r.ttft_ms if r.ttft_ms is not None else r.ttfb_ms
            if v is not None and v == v:
                vals.append(float(v))
    if not vals:
        return None
    s = sorted(vals)
    return {
        "count": len(vals),
        "mean_ttft": statistics.mean(vals),
        "median_ttft": statistics.median(vals),
        "p95_ttft": s[min(len(s) - 1, int(0.95 * len(s)))],
        "std_ttft": statistics.stdev(vals) if len(vals) > 1 else 0.0,
    }


def analyse(rep: Dict[str, Any], uni: Dict[str, Any]) -> Dict[str, Any]:
    r, u = rep["mean_ttft"], uni["mean_ttft"]
    improvement = max(0.0, (u - r) / u) if u > 0 else 0.0
    hit = min(1.0, improvement / 0.8) if u > 0 else 0.0
    pooled = math.sqrt((rep["std_ttft"] ** 2 + uni["std_ttft"] ** 2) / 2)
    t = abs(r - u) / (pooled * math.sqrt(2 / min(rep["count"], uni["count"]))) if pooled > 0 else 0.0
    return {
        "inferred_hit_ratio": hit,
        "ttft_improvement_pct": (u - r) / u * 100 if u > 0 else 0.0,
        "statistical_significance": "significant" if t > 2.0 else "not_significant",
        "t_statistic": t,
        "interpretation": {
            "cache_effectiveness": "high" if hit > 0.6 else "moderate" if hit > 0.3 else "low",
            "ttft_benefit_ms": u - r,
            "confidence": "medium" if t > 1.5 else "low",
        },
    }


def scrape_prefix_counters(url: str, timeout: float = 3.0) -> Optional[Tuple[float, float]]:
    """(hits, queries) summed over labels from a vLLM-style /metrics page."""
    try:
        with urllib.request.urlopen(url.rstrip("/") + "/metrics", timeout=timeout) as r:
            text = r.read().decode("utf-8", "replace")
    except Exception:
        return None
    hits = queries = None
    for line in text.splitlines():
        if line.startswith("#") or " " not in line:
            continue
        name, val = line.rsplit(" ", 1)
        base = name.split("{", 1)[0]
        try:
            v = float(val)
        except ValueError:
            continue
        if base in ("vllm:gpu_prefix_cache_hits_total", "vllm:prefix_cache_hits_total", "vllm_prompt_cache_hits_total"):
            hits = (hits or 0.0) + v
        elif base in ("vllm:gpu_prefix_cache_queries_total", "vllm:prefix_cache_queries_total", "vllm_prompt_cache_requests_total"):
            queries = (queries or 0.0) + v
    return (hits, queries) if hits is not None and queries is not None else None


async def run_experiment(url: str, prompts: List[str], run_dir: str, a: argparse.Namespace) -> Tuple[List[ReqResult], Optional[float]]:
    before = scrape_prefix_counters(url)
    cfg = LoadConfig(
        url=url, run_dir=run_dir, requests=len(prompts), concurrency=a.concurrency, pattern="steady", target_rps=a.target_rps,
        spec=RequestSpec(model=a.model, prompt=prompts[0], max_tokens=a.max_tokens, stream=a.stream), api_key=a.api_key or None,
        insecure=a.insecure, prompts=prompts, quiet=True, seed=42,
    )
    results = await run_load(cfg)
    after = scrape_prefix_counters(url)
    measured = None
    if before and after and after[1] > before[1]:
        measured = (after[0] - before[0]) / (after[1] - before[1])
    return results, measured


def resolve_url(a: argparse.Namespace) -> str:
    if a.url:
        return a.url
    from kvmini.cluster import get_backend

    url = get_backend(a.cluster_backend).get_isvc_url(a.namespace, a.service)
    if not url:
        raise SystemExit(f"ERROR: could not determine InferenceService URL for {a.service} in {a.namespace}")
    return url


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(prog="kvmini cache-probe", description="Infer prompt-cache hit ratio from TTFT deltas")
    ap.add_argument("--namespace", default="ml-prod")
    ap.add_argument("--service", default="demo-llm")
    ap.add_argument("--url", default=None, help="endpoint (skips the cluster lookup)")
    ap.add_argument("--base-requests", type=int, default=100)
    ap.add_argument("--prom-url", default="")
    ap.add_argument("--api-key", default="")
    ap.add_argument("--run-dir", default=None)
    ap.add_argument("--model", default="cache-test")
    ap.add_argument("--max-tokens", type=int, default=64)
    ap.add_argument("--concurrency", type=int, default=10)
    ap.add_argument("--target-rps", type=float, default=None)
    ap.add_argument("--pad-tokens", type=int, default=0, help="prepend N deterministic filler tokens so prompts span whole KV blocks")
    ap.add_argument("--no-stream", dest="stream", action="store_false")
    ap.add_argument("--settle-sec", type=float, default=0.0, help="pause between the two experiments")
    ap.add_argument("--insecure", action="store_true")
    ap.add_argument("--cluster-backend", default=None)
    a = ap.parse_args(argv)

    url = resolve_url(a)
    run_dir = a.run_dir or os.path.join("runs", "cache_analysis_" + time.strftime("%Y-%m-%d_%H-%M-%S"))
    os.makedirs(run_dir, exist_ok=True)
    repeat, unique = generate_prompt_sets(a.base_requests, 42, a.pad_tokens)
    files = {}
    for name, prompts in (("repeat_80", repeat), ("unique_100", unique)):
        files[name] = os.path.join(run_dir, f"{name}_prompts.json")
        with open(files[name], "w") as f:
            json.dump(prompts, f, indent=2)
    print(f"=== Prompt cache analysis: {url} ({a.base_requests} requests per probe) ===")

    async def both() -> Dict[str, Tuple[List[ReqResult], Optional[float]]]:
        out = {"repeat_80": await run_experiment(url, repeat, os.path.join(run_dir, "repeat_80"), a)}
        if a.settle_sec > 0:
            await asyncio.sleep(a.settle_sec)
        out["unique_100"] = await run_experiment(url, unique, os.path.join(run_dir, "unique_100"), a)
        return out

    res = asyncio.run(both())
    rep, uni = ttft_stats(res["repeat_80"][0]), ttft_stats(res["unique_100"][0])
    if not rep or not uni:
        print("ERROR: could not load experiment results", file=sys.stderr)
        return 1
    desc = {"repeat_80": "80% repeated prompts (high cache hit expected)", "unique_100": "100% unique prompts (low cache hit expected)"}
    experiments = {}
    for name, st in (("repeat_80", rep), ("unique_100", uni)):
        experiments[name] = {"description": desc[name], "prompts_file": files[name], "results_dir": os.path.join(run_dir, name), **st, "measured_hit_ratio": res[name][1]}
    an = analyse(rep, uni)
    doc = {
        "cache_analysis": {
            "timestamp": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()),
            "service": a.service,
            "namespace": a.namespace,
            "url": url,
            "method": "ttft_heuristic",
            "experiments": experiments,
            "analysis": an,
            "caveats": CAVEATS,
        }
    }
    out = os.path.join(run_dir, "cache_analysis.json")
    write_json(out, doc)
    print(f"  repeat-80 : {rep['mean_ttft']:.1f} ms avg TTFT (n={rep['count']}, measured hit ratio {res['repeat_80'][1]})")
    print(f"  unique-100: {uni['mean_ttft']:.1f} ms avg TTFT (n={uni['count']}, measured hit ratio {res['unique_100'][1]})")
    print(f"  inferred hit ratio {an['inferred_hit_ratio']:.1%}, effectiveness {an['interpretation']['cache_effectiveness']}, confidence {an['interpretation']['confidence']}")
    print(f"Detailed analysis: {out}")
    return 0


if __name__ == "__main__":
    sys.exit(main())

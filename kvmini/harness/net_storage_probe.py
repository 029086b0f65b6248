"""Network / storage probe (``kvmini probe``, stage between load test and analyze).

Five GETs against the first answering path of ``/v1/models``, ``/health``, ``/``
give endpoint RTT p50 / p95 / mean; optionally an object URL (S3/MinIO presigned,
plain HTTP) is fetched three times for a MB/s estimate.  Output ``io_probe.json``
= ``{endpoint, rtt, s3, generated_at}`` — what the analyzer merges as
``network_rtt_p50_ms``, ``network_rtt_p95_ms``, ``s3_avg_MBps`` (reference
``tools/net_storage_probe.py:16-118``).  Quantiles use the shared quantile routine
(``nearest_rank`` reproduces the reference's ``sorted[int(p*n)]``).
"""

from __future__ import annotations

import argparse
import asyncio
import json
import sys
import time
from typing import Any, Dict, List, Optional

from kvmini.analysis.quantiles import percentile
from kvmini.loadgen.httpclient import HttpClient
from kvmini.utils import write_json


async def _measure_rtt(url: str, attempts: int, timeout: float, insecure: bool) -> Dict[str, Any]:
    vals: List[float] = []
    async with HttpClient(pooled=True, verify_tls=not insecure) as c:
        for _ in range(attempts):
            t0 = time.perf_counter()
            try:
                r = await c.get(url, timeout=timeout)
                await r.read()
                if r.status >= 400:
                    continue
                vals.append((time.perf_counter() - t0) * 1000.0)
            except Exception:
                continue
    return {
        "attempts": attempts,
        "success": len(vals),
        "p50_ms": percentile(vals, 0.5, "nearest_rank") if vals else None,
        "p95_ms": percentile(vals, 0.95, "nearest_rank") if vals else None,
        "avg_ms": sum(vals) / len(vals) if vals else None,
    }


async def _measure_fetch(url: str, attempts: int, timeout: float, insecure: bool) -> Dict[str, Any]:
    tps: List[float] = []
    sizes: List[int] = []
    async with HttpClient(pooled=True, verify_tls=not insecure) as c:
        for _ in range(attempts):
            t0 = time.perf_counter()
            try:
                r = await c.get(url, timeout=timeout)
                data = await r.read()
                if r.status >= 400:
                    continue
                dt = time.perf_counter() - t0
                sizes.append(len(data))
                if dt > 0:
                    tps.append(len(data) / dt / (1024 * 1024))
            except Exception:
                continue
    return {
        "attempts": attempts,
        "success": len(tps),
        "avg_MBps": sum(tps) / len(tps) if tps else None,
        "min_MBps": min(tps) if tps else None,
        "max_MBps": max(tps) if tps else None,
        "avg_size_bytes": sum(sizes) / len(sizes) if sizes else None,
    }


def measure_http_rtt(url: str, attempts: int = 5, timeout: float = 5.0, insecure: bool = False) -> Dict[str, Any]:
    return asyncio.run(_measure_rtt(url, attempts, timeout, insecure))


def measure_object_fetch(url: str, attempts: int = 3, timeout: float = 30.0, insecure: bool = False) -> Dict[str, Any]:
    return asyncio.run(_measure_fetch(url, attempts, timeout, insecure))


def probe(endpoint: str, out: Optional[str] = None, s3_object_url: Optional[str] = None, paths: Optional[List[str]] = None, insecure: bool = False) -> Dict[str, Any]:
    rtt = None
    for p in paths or ["/v1/models", "/health", "/"]:
        try:
            rtt = measure_http_rtt(endpoint.rstrip("/") + p, insecure=insecure)
        except Exception:
            continue
        if rtt and rtt.get("success"):
            rtt["path"] = p
            break
    s3 = None
    if s3_object_url:
        try:
            s3 = measure_object_fetch(s3_object_url, insecure=insecure)
        except Exception:
            s3 = {"error": "fetch_failed"}
    result = {"endpoint": endpoint, "rtt": rtt, "s3": s3, "generated_at": time.time()}
    if out:
        write_json(out, result)
    return result


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(prog="kvmini probe", description="Network & storage profiling probe")
    ap.add_argument("--endpoint", required=True)
    ap.add_argument("--s3-object-url", default=None)
    ap.add_argument("--out", required=True)
    ap.add_argument("--insecure", action="store_true")
    a = ap.parse_args(argv)
    print(json.dumps(probe(a.endpoint, a.out, a.s3_object_url, insecure=a.insecure), indent=2))
    return 0


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python3
"""Throughput of the HIP quantile / histogram / energy kernels vs NumPy on the host.

Sizes span one benchmark run (500 latencies) to a fleet-scale reduction
(16M samples); prints one JSON line per size."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from kvmini.ops import hist_quantile as hq


def live() -> None:
    """The live path's consumer side: raw (request, index, t_ns) events -> gaps -> histogram bins,
    in events per second, one drain at a time (ring slice -> device -> token-gap kernels ->
    stream_hist_kernel), against the per-event Python loop it replaced."""
    from kvmini.loadgen.live import LiveQuantiles

    rng = np.random.default_rng(1)
    for n in (1 << 12, 1 << 16, 1 << 18, 1 << 20):
        n_req = max(16, n // 64)
        req = rng.integers(0, n_req, n)
        order = np.argsort(req, kind="stable")
        idx = np.empty(n, dtype=np.int64)
        idx[order] = np.arange(n) - np.searchsorted(req[order], req[order], side="left")  # 0, 1, 2, ... inside each request, in arrival order
        ev = np.stack([req, idx, np.arange(n, dtype=np.int64) * 3_000_000 // max(1, n_req)], axis=1).astype(np.int64)
        out = {"bench": "live_drain", "events": n, "requests": n_req}
        for device in ("cuda", "cpu"):
            lq = LiveQuantiles(device=device, capacity=max(n, 1 << 18))
            reps = 5
            best = 1e9
            for _ in range(reps):
                lq.ring.buf[:n] = ev
                lq.ring.head += n
                if device == "cuda":
                    torch.cuda.synchronize()
                t0 = time.perf_counter()
                lq.drain()
                if device == "cuda":
                    torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
                ev[:, 2] += 10_000_000_000
            out[f"{device}_ms"] = round(best * 1e3, 3)
            out[f"{device}_events_per_s"] = round(n / best)
        # the loop this path replaced (per-event Python on the host)
        last, gaps = {}, []
        t0 = time.perf_counter()
        for r, k, t in ev[: min(n, 1 << 16)].tolist():
            prev = last.get(r)
            if prev is not None and k > 0:
                gaps.append((t - prev) * 1e-6)
            last[r] = t
        out["python_loop_events_per_s"] = round(min(n, 1 << 16) / (time.perf_counter() - t0))
        print(json.dumps(out), flush=True)


def main() -> None:
    if "--live" in sys.argv[1:]:
        return live()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    ps = (0.5, 0.95, 0.99)
    for n in (500, 1 << 16, 1 << 20, 1 << 24):
        x = rng.lognormal(5.0, 0.6, n)
        t0 = time.perf_counter()
        ref = np.quantile(x, ps)
        t_np = time.perf_counter() - t0
        xd = torch.from_numpy(x).to(dev)
        for _ in range(3):
            got = hq.batched_quantiles(xd.view(1, -1), ps)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            got = hq.batched_quantiles(xd.view(1, -1), ps)
        torch.cuda.synchronize()
        t_hip = (time.perf_counter() - t0) / reps
        got = np.asarray(got).reshape(-1)[:3]
        sh = hq.StreamingHistogram(1, dev)
        sh.update(xd)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            sh.update(xd)
        torch.cuda.synchronize()
        t_stream = (time.perf_counter() - t0) / reps
        print(json.dumps({"n": n, "numpy_ms": t_np * 1e3, "hip_exact_ms": t_hip * 1e3, "hip_stream_hist_ms": t_stream * 1e3, "speedup_exact": t_np / t_hip,
                          "GBps_stream": n * 8 / t_stream / 1e9, "max_abs_err": float(np.max(np.abs(got - ref)))}), flush=True)


if __name__ == "__main__":
    main()

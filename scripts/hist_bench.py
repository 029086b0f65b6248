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


def main() -> None:
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

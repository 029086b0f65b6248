#!/usr/bin/env python3
"""Per-step timeline summary from a rocprofv3 ``--kernel-trace`` CSV.

A decode step of the engine ends with ``argmax_kernel``; the trace is cut at those
dispatches and, for the last ``--steps`` steps, this prints the span of a step (first
kernel start to last kernel end), the sum of kernel durations, the sum of the idle gaps
between consecutive kernels, and per kernel name the mean duration and the mean gap in
front of it.  Output: one JSON document (stdout or ``--out``).
"""
from __future__ import annotations

import argparse
import csv
import glob
import json
import os
import re
import statistics
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(<[^>]*>)?)", name)
    return (m.group(1) if m else name)[:80]


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("trace", help="kernel_trace.csv or a directory that contains one")
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--delimiter", default="argmax_kernel")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    path = a.trace
    if os.path.isdir(path):
        hits = glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
        if not hits:
            print("no kernel_trace.csv under", path, file=sys.stderr)
            return 1
        path = hits[0]
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    cuts = [i for i, r in enumerate(rows) if a.delimiter in r[2]]
    if len(cuts) < a.steps + 1:
        print("not enough steps in trace", file=sys.stderr)
        return 1
    cuts = cuts[-(a.steps + 1):]
    spans, sums, gaps = [], [], []
    dur = defaultdict(list)
    gap_before = defaultdict(list)
    counts = defaultdict(int)
    for s in range(a.steps):
        seg = rows[cuts[s] + 1: cuts[s + 1] + 1]
        if len(seg) < 10:
            continue
        spans.append((seg[-1][1] - seg[0][0]) / 1e3)
        sums.append(sum(e - b for b, e, _ in seg) / 1e3)
        g = 0.0
        for i, (b, e, n) in enumerate(seg):
            k = short(n)
            dur[k].append((e - b) / 1e3)
            counts[k] += 1
            if i > 0:
                gg = max(0, b - seg[i - 1][1]) / 1e3
                g += gg
                gap_before[k].append(gg)
        gaps.append(g)
    out = {
        "trace": os.path.basename(path),
        "steps": len(spans),
        "kernels_per_step": statistics.median(len(rows[cuts[s] + 1: cuts[s + 1] + 1]) for s in range(a.steps)),
        "step_span_us_median": statistics.median(spans),
        "kernel_sum_us_median": statistics.median(sums),
        "gap_sum_us_median": statistics.median(gaps),
        "per_kernel": {
            k: {"per_step": counts[k] / len(spans), "mean_us": round(statistics.mean(v), 2), "mean_gap_before_us": round(statistics.mean(gap_before[k]), 2) if gap_before[k] else None, "total_us_per_step": round(sum(v) / len(spans), 1)}
            for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))
        },
    }
    txt = json.dumps(out, indent=1)
    if a.out:
        with open(a.out, "w") as f:
            f.write(txt)
    print(txt)
    return 0


if __name__ == "__main__":
    sys.exit(main())

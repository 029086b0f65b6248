"""Legacy latency-based cost calculator (``kvmini calculate``).

Formula (reference ``cost_calculator.py:42-57``): cost per 1k tokens = GPU $/s x
mean successful latency (s) x requests needed for 1k tokens.  Input is the old
two-column text file (``<latency_ms> <status>`` per line); a ``requests.csv`` is
accepted as well.
"""

from __future__ import annotations

import argparse
import csv
import sys
from typing import List, Optional, Tuple


def read_latencies(path: str) -> Tuple[List[float], int]:
    """(latencies of successful requests in ms, number of parsed rows)."""
    ok: List[float] = []
    n = 0
    with open(path, newline="") as f:
        head = f.readline()
        f.seek(0)
        if "," in head and "latency_ms" in head:
            for row in csv.DictReader(f):
                n += 1
                try:
                    if int(float(row.get("status") or 0)) == 200:
                        ok.append(float(row["latency_ms"]))
                except (TypeError, ValueError):
                    continue
        else:
            for line in f:
                parts = line.split()
                if len(parts) < 2:
                    continue
                n += 1
                try:
                    if parts[1] == "200":
                        ok.append(float(parts[0]))
                except ValueError:
                    continue
    return ok, n


def cost_per_1k_tokens(latencies_ms: List[float], gpu_hourly_cost: float, requests_per_1k_tokens: int = 10) -> float:
    avg_s = sum(latencies_ms) / len(latencies_ms) / 1000.0
    return gpu_hourly_cost / 3600.0 * avg_s * requests_per_1k_tokens


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(prog="kvmini calculate", description="Cost per 1k tokens from mean latency (legacy model)")
    ap.add_argument("results_file", help="'<latency_ms> <status>' lines, or a requests.csv")
    ap.add_argument("gpu_hourly_cost", type=float, help="GPU hourly cost in USD (MI355X default in cost.yaml: 3.45)")
    ap.add_argument("--requests-per-1k-tokens", type=int, default=10)
    a = ap.parse_args(argv)
    try:
        ok, _n = read_latencies(a.results_file)
    except FileNotFoundError:
        print(f"Error: Results file '{a.results_file}' not found!", file=sys.stderr)
        return 1
    if not ok:
        print("No successful requests found. Cannot calculate cost.")
        return 1
    avg_ms = sum(ok) / len(ok)
    per_s = a.gpu_hourly_cost / 3600.0
    c = cost_per_1k_tokens(ok, a.gpu_hourly_cost, a.requests_per_1k_tokens)
    print("=== COST CALCULATION ===")
    print(f"GPU hourly cost: ${a.gpu_hourly_cost:.2f}")
    print(f"Requests per 1K tokens: {a.requests_per_1k_tokens}")
    print(f"Results file: {a.results_file}\n")
    print("=== METRICS ===")
    print(f"Successful requests: {len(ok)}")
    print(f"Average latency: {avg_ms:.2f}ms ({avg_ms / 1000:.4f}s)\n")
    print("=== COST ESTIMATION ===")
    print(f"GPU price per second: ${per_s:.6f}")
    print(f"Cost per 1K tokens: ${c:.6f}")
    return 0


if __name__ == "__main__":
    sys.exit(main())

"""Arrival-time schedules for the four traffic patterns.

Shapes follow the reference generator (``scripts/loadtest.py:178-257``):

* ``steady``  — even spacing at the target rate;
* ``poisson`` — exponential inter-arrival times at the target rate;
* ``bursty``  — 80 % of the requests uniformly inside the first 20 % of the
  test duration, the rest uniformly over the remainder;
* ``heavy``   — Pareto(α=1.2) offsets scaled by ``duration/10`` and capped at
  95 % of the duration (most requests early, long tail).

Target rate heuristics are the reference's too (``2 x concurrency`` for steady,
``1.5 x`` otherwise, duration stretched 1.5x for the stochastic patterns); an
explicit ``target_rps`` overrides them.  All randomness comes from one seeded
``random.Random`` so that a schedule is reproducible.
"""

from __future__ import annotations

import random
from typing import List, Optional, Tuple

PATTERNS = ("steady", "poisson", "bursty", "heavy")


def calculate_duration_and_rps(requests: int, concurrency: int, pattern: str, target_rps: Optional[float] = None) -> Tuple[float, float]:
    if pattern not in PATTERNS:
        raise ValueError(f"Unknown pattern: {pattern}")
    if target_rps is not None and target_rps > 0:
        stretch = 1.0 if pattern == "steady" else 1.5
        return requests / target_rps * stretch, float(target_rps)
    if pattern == "steady":
        rps = concurrency * 2.0
        return requests / rps, rps
    rps = concurrency * 1.5
    return requests / rps * 1.5, rps


def generate_arrival_times(pattern: str, num_requests: int, duration_sec: float, rps: float, rng: Optional[random.Random] = None) -> List[float]:
    rng = rng or random.Random()
    if pattern == "steady":
        interval = 1.0 / rps
        return [i * interval for i in range(num_requests)]
    if pattern == "poisson":
        t = 0.0
        out = []
        for _ in range(num_requests):
            t += rng.expovariate(rps)
            out.append(t)
        return out
    if pattern == "bursty":
        burst_n = int(num_requests * 0.8)
        burst_dur = duration_sec * 0.2
        out = [rng.uniform(0.0, burst_dur) for _ in range(burst_n)]
        out += [burst_dur + rng.uniform(0.0, duration_sec - burst_dur) for _ in range(num_requests - burst_n)]
        out.sort()
        return out
    if pattern == "heavy":
        out = []
        for _ in range(num_requests):
            u = 1.0 - rng.random()  # (0, 1]
            pareto = (1.0 / u) ** (1.0 / 1.2) - 1.0
            out.append(min(pareto / 10.0 * duration_sec, duration_sec * 0.95))
        out.sort()
        return out
    raise ValueError(f"Unknown pattern: {pattern}")

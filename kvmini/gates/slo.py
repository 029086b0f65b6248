"""SLO gate (``kvmini gate`` / ``tools/gate.py``): results vs budgets, exit code 3.

Budgets (``slo.json``): ``p95_ms``, ``error_rate``, ``$per_1k_tokens``,
``cold_multiplier_max`` (cold p95 / warm p95), ``Wh_per_1k_tokens_max`` (from
``energy.json:Wh_per_1k_tokens_active`` else ``results.json:energy_wh_per_1k_tokens``)
and an optional ``fairness`` block (``tenant_p95_ms_max``,
``throughput_share_diff_max``, ``guard_triggers_max``) checked against a
``fairness_summary.json`` — the budget set and exit code of the reference
(``tools/gate.py:46-153``).  Additions: ``ttft_ms_max`` and ``throughput_rps_min``
budgets, and ``--strict`` which turns a budget whose metric is missing into a
failure (the reference silently skips those).
"""

from __future__ import annotations

import argparse
import os
import sys
from typing import Any, Dict, List, Optional, Tuple

from kvmini.utils import load_json

EXIT_VIOLATION = 3
Row = Tuple[str, Any, Any, bool]


def _load(path: Optional[str]) -> Dict[str, Any]:
    if not path or not os.path.exists(path):
        return {}
    return load_json(path)


def evaluate(slo: Dict[str, Any], res: Dict[str, Any], eng: Optional[Dict[str, Any]] = None, fair: Optional[Dict[str, Any]] = None, strict: bool = False) -> Tuple[List[Row], List[str]]:
    eng = eng or {}
    summary: List[Row] = []
    failures: List[str] = []

    def check(name: str, actual: Any, budget: Any, upper: bool = True) -> None:
        if budget is None:
            return
        if actual is None:
            if strict:
                summary.append((name, None, budget, False))
                failures.append(f"{name} (missing)")
            return
        ok = actual <= budget if upper else actual >= budget
        summary.append((name, actual, budget, ok))
        if not ok:
            failures.append(name)

    check("p95_ms", res.get("p95_ms"), slo.get("p95_ms"))
    check("error_rate", res.get("error_rate"), slo.get("error_rate"))
    check("$per_1k_tokens", res.get("cost_per_1k_tokens"), slo.get("$per_1k_tokens"))
    cold, warm = res.get("cold_p95_ms"), res.get("warm_p95_ms")
    mult = cold / warm if cold and warm and warm > 0 else None
    if mult is not None or strict and res.get("cold_start_count"):
        check("cold_multiplier", mult, slo.get("cold_multiplier_max"))
    energy = eng.get("Wh_per_1k_tokens_active")
    if energy is None:
        energy = res.get("energy_wh_per_1k_tokens")
    check("Wh_per_1k_tokens", energy, slo.get("Wh_per_1k_tokens_max"))
    check("ttft_ms", res.get("time_to_first_token_ms"), slo.get("ttft_ms_max"))
    check("throughput_rps", res.get("throughput_rps"), slo.get("throughput_rps_min"), upper=False)

    if fair is not None:
        fs = slo.get("fairness") or {}
        s = fair.get("summary") or {}
        a, b, shares = s.get("A") or {}, s.get("B") or {}, s.get("throughput_share") or {}
        tp95 = fs.get("tenant_p95_ms_max")
        if tp95 is not None:
            for name, t in (("fairness_tenantA_p95_ms", a), ("fairness_tenantB_p95_ms", b)):
                v = t.get("p95_ms")
                ok = (v if v is not None else float("inf")) <= tp95
                summary.append((name, v, tp95, ok))
                if not ok:
                    failures.append(name)
        sd = fs.get("throughput_share_diff_max")
        if sd is not None and shares:
            diff = abs((shares.get("A") or 0) - (shares.get("B") or 0))
            check("fairness_share_diff", diff, sd)
        check("fairness_guard_triggers", fair.get("guard_triggers"), fs.get("guard_triggers_max"))
    return summary, failures


def print_summary(summary: List[Row]) -> None:
    print("SLO Gate Summary:\n")
    if not summary:
        return
    print(f"{'Metric':25} {'Actual':>12} {'Budget':>12}  Result")
    print("-" * 60)
    for name, actual, budget, ok in summary:
        a = "n/a" if actual is None else f"{float(actual):12.4f}"
        print(f"{name:25} {a:>12} {float(budget):12.4f}  {'PASS' if ok else 'FAIL'}")


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(prog="kvmini gate", description="SLO Gate")
    ap.add_argument("--results", default=None)
    ap.add_argument("--energy", default=None)
    ap.add_argument("--slo", required=True)
    ap.add_argument("--fairness", default=None, help="fairness_summary.json for the multi-tenant gate")
    ap.add_argument("--strict", action="store_true", help="a budget whose metric is missing fails")
    a = ap.parse_args(argv)
    if not os.path.exists(a.slo):
        print(f"ERROR: SLO file {a.slo} not found", file=sys.stderr)
        return 2
    summary, failures = evaluate(_load(a.slo), _load(a.results), _load(a.energy), _load(a.fairness) if a.fairness else None, a.strict)
    print_summary(summary)
    if failures:
        print(f"\nFAIL: {len(failures)} budget violation(s): {', '.join(failures)}", file=sys.stderr)
        return EXIT_VIOLATION
    print("\nPASS: All budgets met")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())

"""Canary comparator (``kvmini canary`` / ``tools/canary_compare.py``): baseline vs
candidate run, exit code 2 on regression.

Inputs are run directories, ``results.json`` files or bundle ``.tar.gz`` archives.
Metrics, directions and relative thresholds are the reference's
(``tools/canary_compare.py:19-25``): p95 +10 %, RPS -10 %, error rate +1 %,
$/1k tokens +10 %, Wh/1k tokens +10 %; plus TTFT +10 %.  A zero baseline with a
non-zero candidate on a lower-is-better metric counts as a regression.
"""

from __future__ import annotations

import argparse
import html
import io
import json
import os
import sys
import tarfile
from typing import Any, Dict, List, Optional, Tuple

EXIT_REGRESSION = 2

KEYS = [
    ("p95_ms", "lower_better", 0.10),
    ("throughput_rps", "higher_better", 0.10),
    ("error_rate", "lower_better", 0.01),
    ("cost_per_1k_tokens", "lower_better", 0.10),
    ("energy_wh_per_1k_tokens", "lower_better", 0.10),
    ("time_to_first_token_ms", "lower_better", 0.10),
]


def load_results_from_path(path: str) -> Dict[str, Any]:
    if os.path.isdir(path):
        with open(os.path.join(path, "results.json")) as f:
            return json.load(f)
    if path.endswith(".json") and os.path.exists(path):
        with open(path) as f:
            return json.load(f)
    if (path.endswith(".tar.gz") or path.endswith(".tgz")) and os.path.exists(path):
        with tarfile.open(path, "r:gz") as tar:
            member = next((m for m in tar.getmembers() if m.isfile() and (m.name.endswith("/results.json") or m.name == "results.json")), None)
            if member is None:
                raise FileNotFoundError("results.json not found in bundle")
            f = tar.extractfile(member)
            if f is None:
                raise FileNotFoundError("unable to read results.json from bundle")
            return json.load(io.TextIOWrapper(f))
    raise FileNotFoundError(path)


def compare(b: Dict[str, Any], c: Dict[str, Any]) -> Tuple[Dict[str, Any], bool]:
    deltas: Dict[str, Any] = {}
    regression = False
    for key, direction, thr in KEYS:
        bv, cv = b.get(key), c.get(key)
        if not isinstance(bv, (int, float)) or not isinstance(cv, (int, float)):
            deltas[key] = {"baseline": bv, "candidate": cv, "delta": None, "regression": False}
            continue
        if bv == 0:
            rel = float("inf") if direction == "lower_better" and cv > 0 else 0.0
        else:
            rel = (cv - bv) / bv
        reg = rel > thr if direction == "lower_better" else rel < -thr
        deltas[key] = {"baseline": bv, "candidate": cv, "delta": rel if rel != float("inf") else None, "regression": reg, "threshold": thr, "direction": direction}
        regression = regression or reg
    return deltas, regression


def write_reports(deltas: Dict[str, Any], out_html: str, out_json: str) -> None:
    os.makedirs(os.path.dirname(out_html) or ".", exist_ok=True)
    with open(out_json, "w") as f:
        json.dump(deltas, f, indent=2)
    rows = []
    for k, v in deltas.items():
        d = v.get("delta")
        ds = f"{d:+.1%}" if isinstance(d, (int, float)) else "NA"
        cls = "bad" if v.get("regression") else "ok"
        rows.append(f"<tr class='{cls}'><td>{html.escape(k)}</td><td>{v.get('baseline')}</td><td>{v.get('candidate')}</td><td>{ds}</td><td>{'REGRESS' if v.get('regression') else 'OK'}</td></tr>")
    doc = (
        "<!DOCTYPE html><html><head><meta charset='utf-8'><title>Canary Compare</title>"
        "<style>body{font-family:system-ui,sans-serif;margin:2em}table{border-collapse:collapse}td,th{border:1px solid #ccc;padding:6px 10px}"
        "tr.bad td{background:#fde8e8}tr.ok td{background:#f2fbf2}</style></head><body><h2>Canary Comparison</h2>"
        "<table><tr><th>Metric</th><th>Baseline</th><th>Candidate</th><th>Δ rel</th><th>Status</th></tr>" + "".join(rows) + "</table></body></html>"
    )
    with open(out_html, "w") as f:
        f.write(doc)


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(prog="kvmini canary", description="Canary comparator")
    ap.add_argument("--baseline", required=True)
    ap.add_argument("--candidate", required=True)
    ap.add_argument("--out", required=True, help="Output HTML path for the delta view")
    ap.add_argument("--json", default=None, help="Output JSON path (default: alongside HTML)")
    a = ap.parse_args(argv)
    try:
        b, c = load_results_from_path(a.baseline), load_results_from_path(a.candidate)
    except (FileNotFoundError, ValueError) as e:
        print(f"ERROR: {e}", file=sys.stderr)
        return 1
    deltas, regression = compare(b, c)
    out_json = a.json or os.path.splitext(a.out)[0] + ".json"
    write_reports(deltas, a.out, out_json)
    print(f"Wrote {a.out} and {out_json}")
    return EXIT_REGRESSION if regression else 0


if __name__ == "__main__":
    raise SystemExit(main())

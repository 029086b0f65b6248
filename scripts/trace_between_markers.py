#!/usr/bin/env python3
"""Per-kernel totals of the part of a rocprofv3 kernel trace that lies between two marker launches.

    python scripts/trace_between_markers.py <kernel_trace.csv> [--marker arange] [--out summary.json]

The workload launches a kernel whose name contains ``--marker`` once before and once after the
region of interest (scripts/prefill_profile.py does, around its last prompt); everything else of
the process — model build, warm-up, graph capture — is left out of the table.  Prints kernels by
total time with calls, average and share, the busy time and the wall time of the region."""
from __future__ import annotations

import argparse
import csv
import json
import re
import sys


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "").replace("at::native::", "").replace("void ", "")
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)I(.*?)EEv", name)
    if m:
        args = re.findall(r"L[ib](\d+)E", m.group(2))
        return f"{m.group(1)}<{','.join(args)}>"
    return name.split("(")[0][:100]


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--marker", default="arange")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if a.marker in r["Kernel_Name"]]
    if len(marks) < 2:
        print(f"need two launches matching {a.marker!r}, found {len(marks)}", file=sys.stderr)
        return 1
    region = rows[marks[-2] + 1 : marks[-1]]
    wall = int(rows[marks[-1]]["Start_Timestamp"]) - int(rows[marks[-2]]["End_Timestamp"])
    tot: dict = {}
    for r in region:
        k = short(r["Kernel_Name"])
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        c = tot.setdefault(k, [0, 0])
        c[0] += 1
        c[1] += d
    busy = sum(v[1] for v in tot.values())
    table = [{"kernel": k, "calls": v[0], "total_us": round(v[1] / 1e3, 1), "avg_us": round(v[1] / v[0] / 1e3, 2), "share": round(v[1] / busy, 4)} for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])]
    for t in table:
        print(f"{t['calls']:6d} {t['total_us']:10.1f} us  avg {t['avg_us']:8.2f}  {100 * t['share']:5.1f} %  {t['kernel']}")
    print(f"kernels {len(region)}  busy {busy / 1e6:.3f} ms  wall {wall / 1e6:.3f} ms  gaps {(wall - busy) / 1e6:.3f} ms")
    if a.out:
        json.dump({"marker": a.marker, "kernels": len(region), "busy_ms": busy / 1e6, "wall_ms": wall / 1e6, "table": table}, open(a.out, "w"), indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())

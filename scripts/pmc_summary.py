#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc output (``*counter_collection.csv``) per kernel name:
mean counter value per dispatch.  Usage: pmc_summary.py DIR [DIR ...] > summary.json"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(<[^>]*>)?)", name)
    return (m.group(1) if m else name)[:90]


acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in acc.items():
    row = {c: sum(v) / len(v) for c, v in cs.items()}
    row["dispatches"] = max(len(v) for v in cs.values())
    if row.get("SQ_LDS_IDX_ACTIVE"):
        row["lds_bank_conflict_rate"] = row.get("SQ_LDS_BANK_CONFLICT", 0.0) / row["SQ_LDS_IDX_ACTIVE"]
    out[k] = row
keep = {k: v for k, v in out.items() if any(t in k for t in ("stream_gemm", "attention", "residual_reduce", "rope_kv", "argmax", "radix", "stream_hist", "equal_hist", "trapz", "sample_rows", "embed_fed", "scale_rows", "wide_gemm", "Cijk", "allreduce", "logits_mask"))}
print(json.dumps(keep, indent=1))

#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of scripts/live_quantiles_gpu.py: how many stream_hist_kernel
dispatches ran, and how many engine kernels were dispatched AFTER each of them (a histogram
update that ran during the load phase is followed by decode kernels)."""
import csv
import glob
import json
import sys

rows = []
for path in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), r["Kernel_Name"]))
rows.sort()
hist = [i for i, (_, n) in enumerate(rows) if "stream_hist_kernel" in n]
eng = [i for i, (_, n) in enumerate(rows) if "stream_gemm3_kernel" in n or "decode_attention" in n]
after = [sum(1 for j in eng if j > i) for i in hist]
t0, t1 = (rows[eng[0]][0], rows[eng[-1]][0]) if eng else (0, 0)
inside = sum(1 for i in hist if t0 <= rows[i][0] <= t1)
print(json.dumps({"kernel_dispatches": len(rows), "stream_hist_dispatches": len(hist), "hist_dispatches_inside_the_engine_kernel_span": inside,
                  "engine_kernels_after_first_hist": after[0] if after else None, "engine_kernels_after_last_hist": after[-1] if after else None}))

#!/bin/bash
# Per-kernel times of the long-context decode path: one sequence of $1 tokens (default 32768).
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
LEN="${1:-32768}"
KV="${2:-auto}"
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH="$R"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_longctx" -o lc -- python3 "$R/scripts/gpu_microbench.py" --what longctx --lens "$LEN" --kv-dtypes "$KV" --out "$R/gpurun_out/longctx_prof.json" > "$R/gpurun_out/prof_longctx.log" 2>&1 || { tail -20 "$R/gpurun_out/prof_longctx.log"; exit 1; }
tail -2 "$R/gpurun_out/prof_longctx.log"
python3 - "$R/gpurun_out/prof_longctx" <<'PY'
import csv, glob, sys
for p in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(p)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:14]:
        print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>7s} avg_us {float(r["AverageNs"])/1e3:9.1f} total_ms {float(r["TotalDurationNs"])/1e6:9.1f} {r["Percentage"]}%')
PY
find "$R/gpurun_out/prof_longctx" -type f ! -name "*stats.csv" -delete

#!/bin/bash
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH="$R"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_live" -o live -- python3 "$R/scripts/live_quantiles_gpu.py" > "$R/gpurun_out/prof_live.log" 2>&1 || { tail -30 "$R/gpurun_out/prof_live.log"; exit 1; }
grep '"live"' "$R/gpurun_out/prof_live.log" | tail -1 | tee "$R/gpurun_out/live_vs_offline.json"
python3 "$R/scripts/live_trace_check.py" "$R/gpurun_out/prof_live" | tee "$R/gpurun_out/live_trace_check.json"
find "$R/gpurun_out/prof_live" -type f ! -name "*stats.csv" -delete

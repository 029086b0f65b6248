#!/bin/bash
# Per-kernel times of a decode step: scripts/gpu_profile_step.sh <quantization|none> [batch]
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
Q="${1:-none}"
B="${2:-20}"
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH="$R"
QA=()
[ "$Q" != none ] && QA=(--quantization "$Q")
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_step_$Q" -o st -- python3 "$R/scripts/gpu_microbench.py" --what step --kernels hip --batch "$B" "${QA[@]}" --out "$R/gpurun_out/step_prof_$Q.json" > "$R/gpurun_out/prof_step_$Q.log" 2>&1 || { tail -20 "$R/gpurun_out/prof_step_$Q.log"; exit 1; }
grep decode_step "$R/gpurun_out/prof_step_$Q.log" | cut -c1-200
python3 - "$R/gpurun_out/prof_step_$Q" <<'PY'
import csv, glob, sys
for p in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(p)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:16]:
        print(f'{r["Name"][:110]:110s} calls {r["Calls"]:>7s} avg_us {float(r["AverageNs"])/1e3:8.1f} {r["Percentage"]}%')
PY
find "$R/gpurun_out/prof_step_$Q" -type f ! -name "*stats.csv" -delete

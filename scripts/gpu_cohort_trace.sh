#!/bin/bash
# rocprofv3 kernel trace of the headline's once-per-cycle 254-token step (scripts/cohort_step.py),
# cut to that one step (scripts/trace_between_markers.py).  Writes gpurun_out/<name>/{region.json,run.log}
# (name = $1, default cohort_trace).  GPU box.
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
D="${1:-cohort_trace}"
mkdir -p "$R/gpurun_out/$D"
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH="$R"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/$D" -o cohort -- python3 "$R/scripts/cohort_step.py" > "$R/gpurun_out/$D/run.log" 2>&1 || { tail -30 "$R/gpurun_out/$D/run.log"; exit 1; }
tail -1 "$R/gpurun_out/$D/run.log"
T=$(find "$R/gpurun_out/$D" -name "*kernel_trace.csv" | head -1)
python3 "$R/scripts/trace_between_markers.py" "$T" --out "$R/gpurun_out/$D/region.json" || exit 1
find "$R/gpurun_out/$D" -type f -name "*.csv" -delete

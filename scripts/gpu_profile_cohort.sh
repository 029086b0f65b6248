#!/bin/bash
# rocprofv3 kernel statistics of the headline's once-per-cycle 254-token step and of a 2048-token prompt
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH="$R"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_cohort" -o cohort -- python3 "$R/scripts/cohort_step.py" > "$R/gpurun_out/prof_cohort.log" 2>&1 || { tail -30 "$R/gpurun_out/prof_cohort.log"; exit 1; }
tail -2 "$R/gpurun_out/prof_cohort.log"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_prefill2048" -o prefill -- python3 "$R/scripts/prefill_profile.py" > "$R/gpurun_out/prof_prefill2048.log" 2>&1 || { tail -30 "$R/gpurun_out/prof_prefill2048.log"; exit 1; }
tail -2 "$R/gpurun_out/prof_prefill2048.log"
find "$R/gpurun_out/prof_cohort" "$R/gpurun_out/prof_prefill2048" -type f ! -name "*stats.csv" -delete
find "$R/gpurun_out/prof_cohort" "$R/gpurun_out/prof_prefill2048" -name "*kernel_stats.csv"

#!/bin/bash
# GEMM microbench + rocprofv3 kernel-trace of the engine decode step and of the analysis kernels.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONPATH="$R"
if [ "${SKIP_MICRO:-0}" != "1" ]; then
timeout -k 10 400 python -u "$R/scripts/gpu_microbench.py" --out "$R/gpurun_out/microbench.json" > "$R/gpurun_out/microbench.log" 2>&1 || { tail -30 "$R/gpurun_out/microbench.log"; exit 1; }
tail -16 "$R/gpurun_out/microbench.log"
fi
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_step" -o step -- python3 "$R/scripts/gpu_microbench.py" --what step --out "$R/gpurun_out/microbench_prof.json" > "$R/gpurun_out/prof_step.log" 2>&1 || { tail -30 "$R/gpurun_out/prof_step.log"; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_hist" -o hist -- python3 "$R/scripts/hist_bench.py" > "$R/gpurun_out/prof_hist.log" 2>&1 || { tail -30 "$R/gpurun_out/prof_hist.log"; exit 1; }
tail -12 "$R/gpurun_out/prof_hist.log"
find "$R/gpurun_out/prof_step" "$R/gpurun_out/prof_hist" -name "*kernel_stats.csv" | head
# keep only the stats summaries (the raw traces are large)
find "$R/gpurun_out/prof_step" "$R/gpurun_out/prof_hist" -type f ! -name "*stats.csv" -delete

#!/bin/bash
# rocprofv3 kernel trace of the engine decode step -> per-step timeline summary (kernel
# time vs idle gaps) + kernel statistics.  Writes gpurun_out/trace_step/{gaps.json,*stats.csv}.
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p "$R/gpurun_out/trace_step"
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH="$R"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/trace_step" -o step -- python3 "$R/scripts/gpu_microbench.py" --what step --kernels hip --out "$R/gpurun_out/trace_step/microbench.json" > "$R/gpurun_out/trace_step/run.log" 2>&1 || { tail -30 "$R/gpurun_out/trace_step/run.log"; exit 1; }
python3 "$R/scripts/trace_gaps.py" "$R/gpurun_out/trace_step" --steps 32 --out "$R/gpurun_out/trace_step/gaps.json" > /dev/null || exit 1
find "$R/gpurun_out/trace_step" -type f -name "*.csv" ! -name "*kernel_stats.csv" -delete
head -c 1500 "$R/gpurun_out/trace_step/gaps.json"

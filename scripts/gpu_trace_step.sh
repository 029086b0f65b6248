#!/bin/bash
# rocprofv3 kernel trace of the engine decode step -> per-step timeline summary (kernel
# time vs idle gaps) + kernel statistics.  Writes gpurun_out/<name>/{gaps.json,*stats.csv}
# (name = $1, default trace_step; environment such as KVMINI_PREFETCH passes through).
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
D="${1:-trace_step}"
mkdir -p "$R/gpurun_out/$D"
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH="$R"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/$D" -o step -- python3 "$R/scripts/gpu_microbench.py" --what step --kernels hip --out "$R/gpurun_out/$D/microbench.json" > "$R/gpurun_out/$D/run.log" 2>&1 || { tail -30 "$R/gpurun_out/$D/run.log"; exit 1; }
python3 "$R/scripts/trace_gaps.py" "$R/gpurun_out/$D" --steps 32 --out "$R/gpurun_out/$D/gaps.json" > /dev/null || exit 1
find "$R/gpurun_out/$D" -type f -name "*.csv" ! -name "*kernel_stats.csv" -delete
head -c 1500 "$R/gpurun_out/$D/gaps.json"

#!/bin/bash
# One GPU-box visit: kernel tests, smoke, headline bench, decode-step microbench.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/pytest_gpu.log; exit 1; }
tail -3 gpurun_out/pytest_gpu.log
timeout -k 10 300 python -u -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1 || { tail -30 gpurun_out/smoke.log; exit 1; }
tail -2 gpurun_out/smoke.log
timeout -k 10 420 python -u bench.py --steps 2 --warmup 1 > gpurun_out/bench.log 2> gpurun_out/bench.err || { tail -30 gpurun_out/bench.err; exit 1; }
tail -1 gpurun_out/bench.log
if [ "${WITH_MICRO:-0}" = "1" ]; then
timeout -k 10 300 python -u scripts/gpu_microbench.py > gpurun_out/microbench.log 2>&1 || { tail -30 gpurun_out/microbench.log; exit 1; }
tail -25 gpurun_out/microbench.log
fi

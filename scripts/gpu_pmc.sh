#!/bin/bash
# Hardware-counter passes over the engine's decode kernels (one rocprofv3 --pmc run per counter
# group; no tracing domains besides the kernel trace).  Output: gpurun_out/pmc/summary.json.
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/pmc"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp PYTHONPATH="$R"
timeout -s KILL 240 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pass1" -o p1 -- python3 "$R/scripts/pmc_step.py" > "$OUT/pass1.log" 2>&1 || { tail -20 "$OUT/pass1.log"; exit 1; }
timeout -s KILL 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pass2" -o p2 -- python3 "$R/scripts/pmc_step.py" > "$OUT/pass2.log" 2>&1 || { tail -20 "$OUT/pass2.log"; exit 1; }
timeout -s KILL 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pass3" -o p3 -- python3 "$R/scripts/pmc_step.py" > "$OUT/pass3.log" 2>&1 || { tail -20 "$OUT/pass3.log"; exit 1; }
python3 "$R/scripts/pmc_summary.py" "$OUT/pass1" "$OUT/pass2" "$OUT/pass3" > "$OUT/summary.json" || exit 1
find "$OUT" -name "*.csv" -size +2M -delete
head -c 1800 "$OUT/summary.json"

#!/bin/bash
# Hardware-counter passes (one rocprofv3 --pmc run per counter group; no tracing domains besides
# the kernel trace).  Usage: gpu_pmc.sh [NAME SCRIPT [ARGS...]]   default: the engine's decode
# kernels (scripts/pmc_step.py).  Output: gpurun_out/pmc_<NAME>/summary.json
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
NAME="${1:-decode}"; shift || true
PROG=("$@"); [ ${#PROG[@]} -eq 0 ] && PROG=("$R/scripts/pmc_step.py")
OUT="$R/gpurun_out/pmc_$NAME"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp PYTHONPATH="$R"
pass() {  # name counters...
  local n="$1"; shift
  timeout -s KILL 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$n" -o "$n" -- python3 "${PROG[@]}" > "$OUT/$n.log" 2>&1 || { tail -20 "$OUT/$n.log"; exit 1; }
}
pass pass1 SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE
pass pass2 FETCH_SIZE
pass pass3 WRITE_SIZE
python3 "$R/scripts/pmc_summary.py" "$OUT/pass1" "$OUT/pass2" "$OUT/pass3" > "$OUT/summary.json" || exit 1
find "$OUT" -name "*.csv" -size +2M -delete
head -c 1500 "$OUT/summary.json"

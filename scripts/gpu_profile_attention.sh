#!/bin/bash
# rocprofv3 evidence for the prompt attention kernel: per-kernel statistics of a 2048-token prompt
# through the engine, and hardware counters of the kernel alone (scripts/gpu_microbench.py --what
# attn; 2048 queries at positions 2048..4095 unless ATTN_SHAPES says otherwise), one counter group per run.  Output: gpurun_out/prof_prefill/, gpurun_out/pmc_attn/summary.json
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp PYTHONPATH="$R" ATTN_SHAPES="${ATTN_SHAPES:-2048:2048}"
OUT="$R/gpurun_out/prof_prefill"; rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o prefill -- python3 "$R/scripts/prefill_profile.py" 2048 > "$OUT/run.log" 2>&1 || { tail -20 "$OUT/run.log"; exit 1; }
find "$OUT" -type f ! -name "*stats.csv" ! -name "run.log" -delete
P="$R/gpurun_out/pmc_attn"; rm -rf "$P"; mkdir -p "$P"
pass() {  # name counters...
  local n="$1"; shift
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$P/$n" -o "$n" -- python3 "$R/scripts/gpu_microbench.py" --what attn > "$P/$n.log" 2>&1 || { tail -20 "$P/$n.log"; exit 1; }
}
pass pass1 SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE
pass pass2 SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
python3 "$R/scripts/pmc_summary.py" "$P/pass1" "$P/pass2" > "$P/summary.json" || exit 1
find "$P" -name "*.csv" -size +2M -delete
cat "$P/summary.json"

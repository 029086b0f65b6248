#!/bin/bash
# Hardware-counter passes over one wide-GEMM configuration (each pass is its own run: the
# profiler does not split counters over passes).  Usage: scripts/gpu_pmc_wide.sh <tag> <profile args...>
set -u
tag=$1; shift
out=gpurun_out/pmc_wide_$tag
mkdir -p $out
cd /tmp 2>/dev/null && export TMPDIR=/tmp; cd - >/dev/null
pass() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python3 scripts/wide_gemm_profile.py "${ARGS[@]}" > $out/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $out/$name.log; return 1; }
}
ARGS=("$@")
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS &&
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM &&
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum &&
pass tcp TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum &&
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
python3 scripts/pmc_summary.py $out 2>/dev/null | tail -40 || true

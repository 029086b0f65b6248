#!/bin/bash
# Telemetry all-gather on one box, world = 1: the three implementations of `kvmini telemetry`
# (torch.distributed all_gather, the C++ ncclAllGather on a private communicator, the one-shot
# peer-memory kernel), real amd-smi samples, 0.25 s gather period for 12 s each.  Writes
# gpurun_out/gather_impls.json with the per-gather host latency (p50 / p95, microseconds).
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
export PYTHONPATH="$R" MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
mkdir -p gpurun_out/gather
port=29611
for impl in torch rccl p2p; do
  MASTER_PORT=$port timeout -k 10 60 python3 -m kvmini telemetry --interval 0.05 --gather-every 0.25 --duration 12 --gather-impl $impl --out gpurun_out/gather/power_$impl.json > gpurun_out/gather/$impl.log 2>&1 || { tail -20 gpurun_out/gather/$impl.log; exit 1; }
  port=$((port + 1))
done
python3 - <<'PY'
import json
out = {}
for impl in ("torch", "rccl", "p2p"):
    d = json.load(open(f"gpurun_out/gather/power_{impl}.json"))
    out[impl] = {k: d.get(k) for k in ("gather_impl", "gather_us_p50", "gather_us_p95", "dropped_ticks", "source")}
json.dump(out, open("gpurun_out/gather_impls.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY

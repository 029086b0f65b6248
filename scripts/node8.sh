#!/bin/bash
# Everything that needs a whole 8 x MI355X node, in one command.  Nothing here has run on such a
# node yet (the build pool gives one GPU per box): the command lines are rehearsed on CPU with the
# tiny model (tests/test_bench_flow.py, tests/test_local_backend.py) and on one GPU with several
# rank processes sharing it.  Artifacts: gpurun_out/node8/.
#
#   scripts/node8.sh [scale] [xgmi] [tp8] [quant] [ab] [dp8]        (default: all)
#
#   scale  bench.py --gpus 1,2,4,8 back to back -> scale.json with the efficiency per N
#   xgmi   peer-memory collectives across the 8 devices: correctness + one-shot / two-shot / RCCL latency table
#   tp8    Llama-3-70B, tensor parallel over 8 GPUs: grid sweep c1..64 with energy per 1k tokens
#   quant  the same deployment with FP8 weights: quantisation sweep
#   ab     A/B over the partition profiles (SPX / DPX / QPX / CPX shares of a GPU) -> HTML cost + energy report
#   dp8    8 engine replicas behind the data-parallel router, headline mix
set -o pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"
OUT="$R/gpurun_out/node8"
mkdir -p "$OUT"
export PYTHONPATH="$R" KVMINI_CLUSTER_BACKEND=local DISABLE_IO_PROBE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
WHAT="${*:-scale xgmi tp8 quant ab dp8}"
NGPU="$(python -c 'import torch; print(torch.cuda.device_count())')"
if [ "$NGPU" -lt 8 ]; then echo "node8.sh needs 8 GPUs, found $NGPU" >&2; exit 2; fi

for what in $WHAT; do
case "$what" in
scale)
  : > "$OUT/scale_lines.jsonl"
  for n in 1 2 4 8; do
    timeout -k 10 900 python bench.py --gpus "$n" --steps 3 --warmup 1 --out-dir "$OUT/bench_n$n" >> "$OUT/scale_lines.jsonl" 2> "$OUT/bench_n$n.err" || { tail -20 "$OUT/bench_n$n.err"; exit 1; }
  done
  python - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
rows = [json.loads(l) for l in open(f"{out}/scale_lines.jsonl") if l.startswith("{")]
base = next(r for r in rows if r["n_gpus"] == 1)["value"]
table = [{"n_gpus": r["n_gpus"], "rps": r["value"], "p95_ms": r["p95_ms"], "itl_p99_ms": r["itl_p99_ms"], "efficiency": round(r["value"] / (base * r["n_gpus"]), 4),
          "rccl_ranks": r["rccl_ranks"], "collective_backend": r.get("collective_backend")} for r in rows]
json.dump({"scaling": "weak", "metric": rows[0]["metric"], "rows": table}, open(f"{out}/scale.json", "w"), indent=1)
print(json.dumps(table, indent=1))
PY
  ;;
xgmi)
  timeout -k 10 300 python scripts/xgmi_selftest.py --world 8 --iters 60 > "$OUT/xgmi_selftest_world8.json" 2> "$OUT/xgmi_selftest.err" || { tail -20 "$OUT/xgmi_selftest.err"; exit 1; }
  for impl in p2p rccl torch; do  # the lock-step telemetry table: one-shot peer-memory store vs RCCL ring vs torch.distributed, 64 gathers each
    timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=8 --master-addr 127.0.0.1 --master-port 29611 -m kvmini telemetry --gather-impl "$impl" \
      --interval 0.05 --gather-every 0.25 --duration 16 --out "$OUT/telemetry_$impl.json" > "$OUT/telemetry_$impl.log" 2>&1 || tail -5 "$OUT/telemetry_$impl.log"
  done
  tail -3 "$OUT/xgmi_selftest_world8.json" ;;
tp8)
  python -m kvmini deploy --cluster-backend local --backend native --model-name llama-3-70b --service l70-tp8 --namespace local --tensor-parallel-size 8 --port 8301 --timeout 1500 \
    --serve-arg=--max-num-seqs --serve-arg=64 --serve-arg=--max-model-len --serve-arg=4096 > "$OUT/tp8_deploy.log" 2>&1 || { tail -20 "$OUT/tp8_deploy.log"; exit 1; }
  timeout -k 10 1800 python -m kvmini grid-sweep --model llama-3-70b --service l70-tp8 --namespace local --cluster-backend local --requests 200 --concurrency 1,2,4,8,16,32,64 \
    --max-tokens 64 --patterns steady --runs-root "$OUT/tp8_grid_runs" --output "$OUT/tp8_grid_sweep.csv" --report > "$OUT/tp8_grid.log" 2>&1 || tail -20 "$OUT/tp8_grid.log"
  python -m kvmini deploy --cluster-backend local --service l70-tp8 --namespace local --delete >> "$OUT/tp8_deploy.log" 2>&1
  tail -12 "$OUT/tp8_grid.log" ;;
quant)
  timeout -k 10 2400 python -m kvmini quant-sweep --local native --port 8302 --config sweeps/quantization_sweep_native.yaml --model llama-3-70b --tensor-parallel-size 8 \
    --output-dir "$OUT/tp8_quant" --runs-root "$OUT/tp8_quant_runs" > "$OUT/tp8_quant.log" 2>&1 || tail -20 "$OUT/tp8_quant.log"
  tail -12 "$OUT/tp8_quant.log" ;;
ab)
  timeout -k 10 2400 python -m kvmini partition-sweep --cluster-backend local --backend native --native-model llama-3-8b --namespace local --service part --profiles cpx,qpx,dpx,spx \
    --requests 200 --concurrency 10 --max-tokens 64 --model llama-3-8b --runs-root "$OUT/partition_runs" --output "$OUT/partition_matrix.csv" --report > "$OUT/partition.log" 2>&1 || tail -20 "$OUT/partition.log"
  tail -8 "$OUT/partition.log" ;;
dp8)
  python -m kvmini deploy --cluster-backend local --backend native --model-name llama-3-8b --service dp8 --namespace local --replicas 8 --port 8303 --timeout 1500 > "$OUT/dp8_deploy.log" 2>&1 || { tail -20 "$OUT/dp8_deploy.log"; exit 1; }
  timeout -k 10 900 python -m kvmini bench --profile headline --model llama-3-8b --service dp8 --namespace local --cluster-backend local --requests 4000 --concurrency 160 --report --run-dir "$OUT/dp8_headline" \
    > "$OUT/dp8_headline.log" 2>&1 || tail -20 "$OUT/dp8_headline.log"
  python -m kvmini deploy --cluster-backend local --service dp8 --namespace local --delete >> "$OUT/dp8_deploy.log" 2>&1
  tail -12 "$OUT/dp8_headline.log" ;;
*) echo "unknown step $what" >&2; exit 2 ;;
esac
done

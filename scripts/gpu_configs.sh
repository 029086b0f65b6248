#!/bin/bash
# The benchmark configurations of BASELINE.json on ONE MI355X box, through the harness CLI
# (native server + load generator + analyzer + cost + reports).  Artifacts: gpurun_out/configs/.
# Usage: scripts/gpu_configs.sh [headline] [grid] [quant] [70b] [ab] [spec] [probes] [build] [structured]   (default: all but probes, build and structured)
set -o pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"
OUT="$R/gpurun_out/configs"
mkdir -p "$OUT"
export PYTHONPATH="$R" KVMINI_CLUSTER_BACKEND=local DISABLE_IO_PROBE=1
WHAT="${*:-headline grid quant 70b ab spec}"
MODEL="${MODEL:-llama-3-8b}"
SRV_PID=""

start_server() {  # name port args...
  local name="$1" port="$2"; shift 2
  python -m kvmini serve --port "$port" --service "$name" "$@" > "$OUT/server_$name.log" 2>&1 &
  SRV_PID=$!
  for _ in $(seq 1 600); do
    if ! kill -0 "$SRV_PID" 2>/dev/null; then echo "server $name died"; tail -20 "$OUT/server_$name.log"; return 1; fi
    if python - "$port" <<'PY' 2>/dev/null
import sys, urllib.request
urllib.request.urlopen(f"http://127.0.0.1:{sys.argv[1]}/health", timeout=1)
PY
    then return 0; fi
    sleep 0.5
  done
  echo "server $name not ready"; return 1
}
stop_server() { [ -n "$SRV_PID" ] && kill "$SRV_PID" 2>/dev/null; wait "$SRV_PID" 2>/dev/null; SRV_PID=""; }
trap stop_server EXIT

for what in $WHAT; do
case "$what" in
headline)  # config 2: 500 req / c20 with a cold restart half-way -> cold/warm split, report, SLO gate
  timeout -k 10 500 python -m kvmini bench --profile headline --model "$MODEL" --serve native --serve-model "$MODEL" --serve-port 8101 --cluster-backend local \
    --cold-restart-at 0.5 --report --run-dir "$OUT/headline" > "$OUT/headline.log" 2>&1 || { tail -30 "$OUT/headline.log"; exit 1; }
  # the gate is expected to flag the provoked cold start (exit 3): record its verdict, do not stop
  python -m kvmini gate --results "$OUT/headline/results.json" --slo slo.json > "$OUT/headline_gate.log" 2>&1; echo "gate exit code: $?" >> "$OUT/headline_gate.log"
  tail -14 "$OUT/headline.log"; tail -12 "$OUT/headline_gate.log" ;;
grid)      # config 4 (single-GPU part): concurrency 1..64 grid with the energy per 1k tokens column
  start_server grid 8102 --model "$MODEL" --max-num-seqs 64 || exit 1
  timeout -k 10 900 python -m kvmini grid-sweep --url http://127.0.0.1:8102 --model "$MODEL" --service grid --namespace local --cluster-backend local --requests 200 --concurrency 1,2,4,8,16,32,64 \
    --max-tokens 64 --patterns steady --runs-root "$OUT/grid_runs" --output "$OUT/grid_sweep.csv" --report > "$OUT/grid.log" 2>&1 || { tail -30 "$OUT/grid.log"; exit 1; }
  stop_server; tail -12 "$OUT/grid.log" ;;
quant)     # config 4: bf16 vs fp8 quantization sweep on the native engine
  timeout -k 10 900 python -m kvmini quant-sweep --local native --port 8103 --config sweeps/quantization_sweep_native.yaml --output-dir "$OUT/quant" --runs-root "$OUT/quant_runs" \
    > "$OUT/quant.log" 2>&1 || { tail -30 "$OUT/quant.log"; exit 1; }
  tail -12 "$OUT/quant.log" ;;
70b)       # Llama-3-70B bf16 on ONE 288 GB GPU (TP=8 needs the 8-GPU node): 141 GB of weights resident
  start_server l70 8104 --model llama-3-70b --max-num-seqs 32 --max-batched-tokens 64 --max-model-len 1024 --kv-memory-fraction 0.5 || exit 1
  timeout -k 10 600 python -m kvmini bench --url http://127.0.0.1:8104 --model llama-3-70b --cluster-backend local --service l70 --namespace local --requests 60 --concurrency 20 --max-tokens 64 \
    --report --run-dir "$OUT/llama70b_1gpu" > "$OUT/llama70b.log" 2>&1 || { tail -30 "$OUT/llama70b.log"; exit 1; }
  stop_server; tail -14 "$OUT/llama70b.log" ;;
ab)        # config 5 (single-box part): native engine vs protocol mocks of the two cluster backends
  start_server "$MODEL-native" 8105 --model "$MODEL" --namespace ab || exit 1
  python -m kvmini mock --port 8106 --model mock-vllm --service "$MODEL-vllm-rocm" --register > "$OUT/mock_vllm.log" 2>&1 & M1=$!
  python -m kvmini mock --port 8107 --model mock-tgi --service "$MODEL-tgi-rocm" --register > "$OUT/mock_tgi.log" 2>&1 & M2=$!
  sleep 2
  timeout -k 10 600 python -m kvmini ab-compare --backends native,vllm-rocm,tgi-rocm --model "$MODEL" --profile standard --toggle-streaming --cluster-backend local --namespace ab \
    --endpoints native=http://127.0.0.1:8105 vllm-rocm=http://127.0.0.1:8106 tgi-rocm=http://127.0.0.1:8107 --run-dir "$OUT/ab" > "$OUT/ab.log" 2>&1 || { kill $M1 $M2; tail -30 "$OUT/ab.log"; exit 1; }
  kill $M1 $M2 2>/dev/null; stop_server; tail -14 "$OUT/ab.log" ;;
spec)      # config 3: speculative decoding TTFT / inter-token delta report
  # (a) early-exit self-draft (first 8 layers) on synthetic weights whose tail is damped so the draft agrees
  timeout -k 10 600 python -m kvmini spec-delta --model "$MODEL" --profile speculative-decoding --draft self:8 --tail-damping 0.02 --port 8108 --output-dir "$OUT/spec_self8" \
    > "$OUT/spec_self8.log" 2>&1 || { tail -30 "$OUT/spec_self8.log"; exit 1; }
  tail -14 "$OUT/spec_self8.log"
  # (b) the profile's independent draft preset: random weights never agree -> overhead bound
  timeout -k 10 600 python -m kvmini spec-delta --model "$MODEL" --profile speculative-decoding --port 8109 --output-dir "$OUT/spec_llama1b" \
    > "$OUT/spec_llama1b.log" 2>&1 || { tail -30 "$OUT/spec_llama1b.log"; exit 1; }
  tail -14 "$OUT/spec_llama1b.log" ;;
probes)    # prefix-cache probe, OpenAI parity matrix and the fairness guard against the native engine
  start_server probes 8108 --model "$MODEL" --max-num-seqs 32 || exit 1
  timeout -k 10 300 python -m kvmini cache-probe --url http://127.0.0.1:8108 --model "$MODEL" --base-requests 100 --concurrency 8 --max-tokens 16 --pad-tokens 512 --settle-sec 1 \
    --cluster-backend local --service probes --namespace local --run-dir "$OUT/cache_probe" > "$OUT/cache_probe.log" 2>&1 || { tail -30 "$OUT/cache_probe.log"; exit 1; }
  tail -8 "$OUT/cache_probe.log"
  timeout -k 10 120 python -m kvmini parity --url http://127.0.0.1:8108 --model "$MODEL" --out "$OUT/parity/capability_matrix.json" --html "$OUT/parity/capability_matrix.html" > "$OUT/parity.log" 2>&1
  tail -8 "$OUT/parity.log"
  timeout -k 10 200 python -m kvmini fairness --url http://127.0.0.1:8108 --model "$MODEL" --max-tokens 64 --tenant-a-requests 200 --tenant-b-requests 600 --tenant-a-concurrency 8 \
    --tenant-b-concurrency 40 --p95-budget-ms 280 --run-dir "$OUT/fairness" > "$OUT/fairness.log" 2>&1
  tail -12 "$OUT/fairness.log"
  stop_server ;;
structured) # the structured-output and tool-calling profiles (guided JSON on the device) and the capability matrix
  start_server structured 8110 --model "$MODEL" --max-num-seqs 32 || exit 1
  for prof in structured-output tool-calling; do
    timeout -k 10 400 python -m kvmini bench --profile "$prof" --url http://127.0.0.1:8110 --model "$MODEL" --service structured --namespace local --cluster-backend local \
      --report --run-dir "$OUT/$prof" > "$OUT/$prof.log" 2>&1 || { tail -30 "$OUT/$prof.log"; exit 1; }
    tail -14 "$OUT/$prof.log"
  done
  timeout -k 10 120 python -m kvmini parity --url http://127.0.0.1:8110 --model "$MODEL" --out "$OUT/parity/capability_matrix.json" --html "$OUT/parity/capability_matrix.html" > "$OUT/parity.log" 2>&1
  tail -8 "$OUT/parity.log"
  stop_server ;;
build)     # engine "build" (start -> healthy: weight re-ordering, KV sizing, graph capture) vs served performance, per engine profile
  for prof in profiles/engine/*.yaml; do
    timeout -k 10 400 python -m kvmini build-vs-perf --profile "$prof" --mode local --port 8109 --requests 200 --concurrency 10 --max-tokens 64 --streaming \
      --output "$OUT/build_vs_perf.csv" >> "$OUT/build_vs_perf.log" 2>&1 || { tail -30 "$OUT/build_vs_perf.log"; exit 1; }
  done
  cat "$OUT/build_vs_perf.csv" ;;
esac
done

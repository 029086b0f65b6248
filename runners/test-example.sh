#!/usr/bin/env bash
# Worked examples of the backend A/B harness (`runners/ab-compare.sh` = `kvmini ab-compare`).
#
#   ./runners/test-example.sh          print the example commands and what they produce
#   ./runners/test-example.sh --run    run example 1 for real against two mock servers on this host
#                                      (no cluster, no GPU): two endpoints with different latency
#                                      profiles stand in for two runtimes
set -euo pipefail
cd "$(dirname "${BASH_SOURCE[0]}")/.."
PY="${PYTHON:-python3}"

show() {
  cat <<'TXT'
=== Backend A/B harness: examples ===

1. Two runtimes, same model, same traffic:
     ./runners/ab-compare.sh --backends vllm-rocm,native --model llama-3-8b --profile standard

2. What streaming costs on one runtime:
     ./runners/ab-compare.sh --backends native --model llama-3-8b --toggle-streaming --profile burst

3. Sustained throughput, three runtimes, 1000 requests:
     ./runners/ab-compare.sh --backends vllm-rocm,tgi-rocm,native --model llama-3-8b --profile sustained --requests 1000

4. GPU partition modes of one MI355X (SPX vs CPX) as the "backends":
     ./runners/ab-compare.sh --backends native --model llama-3-8b --partition-profile cpx --profile standard

5. Endpoints that already run (nothing is deployed):
     ./runners/ab-compare.sh --backends a,b --endpoints a=http://10.0.0.5:8000 b=http://10.0.0.6:8000 --model llama-3-8b

Backends:  vllm-rocm (vLLM on ROCm), tgi-rocm (Text Generation Inference on ROCm), native (this repository's engine)
Profiles:  standard (200 requests, concurrency 10), burst (300 / 25), sustained (500 / 50); see runners/profiles/

Output:
  runs/ab_compare_<timestamp>/
    ab_comparison.csv         one row per backend x streaming mode: p50 / p95 / TTFT / throughput / cost / energy
    comparison_report.json    winners per metric and the streaming trade-off
    ab_report.html            the same as a page
    <backend>_streaming_<bool>/   the run directory of each cell (requests.csv, results.json, meta.json)

The harness names the fastest time to first token, the highest throughput, the cheapest 1K tokens and the
least energy per 1K tokens, and never deploys two cells at once: one GPU serves one backend at a time.
TXT
}

run() {
  tmp="$(mktemp -d)"
  trap 'kill "${pids[@]}" 2>/dev/null || true; rm -rf "$tmp"' EXIT
  pids=()
  "$PY" -m kvmini mock --port 18081 --ttft-ms 20 --tpot-ms 4 >"$tmp/a.log" 2>&1 &
  pids+=($!)
  "$PY" -m kvmini mock --port 18082 --ttft-ms 45 --tpot-ms 3 >"$tmp/b.log" 2>&1 &
  pids+=($!)
  for port in 18081 18082; do
    for _ in $(seq 1 50); do
      if "$PY" -c "import urllib.request,sys; urllib.request.urlopen('http://127.0.0.1:$port/health', timeout=1)" 2>/dev/null; then break; fi
      sleep 0.1
    done
  done
  DISABLE_IO_PROBE=1 "$PY" -m kvmini ab-compare --backends fast-first-token,fast-decode \
    --endpoints fast-first-token=http://127.0.0.1:18081 fast-decode=http://127.0.0.1:18082 \
    --model mock-llm --requests 40 --concurrency 4 --toggle-streaming --cluster-backend fake --no-energy --wait-between 0 \
    --run-dir "${RUN_DIR:-runs/ab_compare_example}"
  echo "report: ${RUN_DIR:-runs/ab_compare_example}/comparison_report.json"
}

case "${1:-}" in
  --run) run ;;
  "" | --help | -h) show ;;
  *) echo "usage: $0 [--run]" >&2; exit 2 ;;
esac

#!/bin/bash
# Sampled decoding (temperature 0.8, top-p 0.9) through the native server, launch-ahead stepping
# against one step at a time: 200 requests, concurrency 10, 64 tokens.  Prints p50 / p95 / RPS.
set -o pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"
export PYTHONPATH="$R" KVMINI_CLUSTER_BACKEND=local DISABLE_IO_PROBE=1
OUT="$R/gpurun_out/sampled"
mkdir -p "$OUT"
for mode in sync async; do
  extra=""; [ "$mode" = sync ] && extra="--no-async-scheduling"
  python -m kvmini serve --port 8120 --service sampled-$mode --model llama-3-8b $extra > "$OUT/server_$mode.log" 2>&1 &
  PID=$!
  for _ in $(seq 1 600); do
    kill -0 $PID 2>/dev/null || { tail -20 "$OUT/server_$mode.log"; exit 1; }
    python -c "import urllib.request; urllib.request.urlopen('http://127.0.0.1:8120/health', timeout=1)" 2>/dev/null && break
    sleep 0.5
  done
  timeout -k 10 200 python -m kvmini loadtest --url http://127.0.0.1:8120 --model llama-3-8b --requests 200 --concurrency 10 --max-tokens 64 --target-rps 1000 \
    --temperature 0.8 --top-p 0.9 --run-dir "$OUT/$mode" --quiet > "$OUT/load_$mode.log" 2>&1 || { tail "$OUT/load_$mode.log"; kill $PID; exit 1; }
  kill $PID; wait $PID 2>/dev/null
  python - "$OUT/$mode/requests.csv" "$mode" <<'PY'
import csv, sys
import numpy as np
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["status"] == "200"]
lat = np.array([float(r["latency_ms"]) for r in rows])
start = np.array([float(r["start_ms"]) for r in rows])
span = (start + lat).max() - start.min()
print(sys.argv[2], "n", len(rows), "p50 %.1f p95 %.1f ms, %.1f req/s" % (np.percentile(lat, 50), np.percentile(lat, 95), len(rows) / span * 1e3))
PY
done

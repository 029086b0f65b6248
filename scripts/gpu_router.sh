#!/bin/bash
# The data-parallel front-end on a one-GPU box: `kvmini router --replicas 1`, headline mix through it
# (500 requests, concurrency 20, 64 tokens).  What the extra hop costs.
set -o pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"
export PYTHONPATH="$R" KVMINI_CLUSTER_BACKEND=local DISABLE_IO_PROBE=1
OUT="$R/gpurun_out/router"
rm -rf "$OUT"; mkdir -p "$OUT"
python -m kvmini router --replicas 1 --port 8200 --no-register -- --model llama-3-8b --max-batched-tokens 256 > "$OUT/router.log" 2>&1 &
PID=$!
for _ in $(seq 1 600); do
  kill -0 $PID 2>/dev/null || { tail -20 "$OUT/router.log"; exit 1; }
  python -c "import urllib.request; urllib.request.urlopen('http://127.0.0.1:8200/health', timeout=1)" 2>/dev/null && break
  sleep 0.5
done
for target in 8200 8201; do
  python -m kvmini loadtest --url http://127.0.0.1:$target --model llama-3-8b --requests 100 --concurrency 20 --max-tokens 64 --target-rps 1000 --quiet --no-trace --run-dir "$OUT/warm_$target" > /dev/null 2>&1
  timeout -k 10 200 python -m kvmini loadtest --url http://127.0.0.1:$target --model llama-3-8b --requests 500 --concurrency 20 --max-tokens 64 --target-rps 1000 --quiet --no-trace --run-dir "$OUT/run_$target" > "$OUT/load_$target.log" 2>&1 || { tail "$OUT/load_$target.log"; kill -TERM $PID; exit 1; }
  python - "$OUT/run_$target/requests.csv" "$target" <<'PY'
import csv, sys
import numpy as np
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["status"] == "200"]
lat = np.array([float(r["latency_ms"]) for r in rows]); st = np.array([float(r["start_ms"]) for r in rows])
tt = np.array([float(r["ttft_ms"]) for r in rows if r["ttft_ms"] not in ("", "None")])
span = (st + lat).max() - st.min()
print(("router :8200" if sys.argv[2] == "8200" else "direct :8201"), "n", len(rows), "p50 %.1f p95 %.1f ms, first token %.1f ms, %.1f req/s" % (np.percentile(lat, 50), np.percentile(lat, 95), tt.mean(), len(rows) / span * 1e3))
PY
done
kill -TERM $PID; wait $PID 2>/dev/null
exit 0

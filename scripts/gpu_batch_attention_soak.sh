#!/bin/bash
# 140 concurrent streams with 300-token prompts against one native server with an e4m3 KV cache: every step carries
# more than 64 decode rows, so attention runs on the batch decode kernel (MFMA scores, rotating K/V stages) next to
# prompt tiles.  Every request must succeed; the engine's counters are printed at the end.
set -o pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"
export PYTHONPATH="$R" KVMINI_CLUSTER_BACKEND=local DISABLE_IO_PROBE=1
KV="${KV:-fp8}"
OUT="$R/gpurun_out/batch_soak_$KV"
rm -rf "$OUT"; mkdir -p "$OUT"
python -m kvmini serve --port 8131 --service soak --model llama-3-8b --max-num-seqs 160 --max-model-len 2048 --max-batched-tokens 2048 --kv-cache-dtype "$KV" --no-register > "$OUT/server.log" 2>&1 &
PID=$!
for _ in $(seq 1 600); do
  kill -0 $PID 2>/dev/null || { tail -20 "$OUT/server.log"; exit 1; }
  python -c "import urllib.request; urllib.request.urlopen('http://127.0.0.1:8131/health', timeout=1)" 2>/dev/null && break
  sleep 0.5
done
timeout -k 10 250 python -m kvmini loadtest --url http://127.0.0.1:8131 --model llama-3-8b --quiet --no-trace --requests 2500 --concurrency 140 --max-tokens 48 --prompt-tokens 300 --target-rps 100000 --run-dir "$OUT/run" > "$OUT/load.log" 2>&1
rc=$?
python - "$OUT" <<'PY'
import csv, sys, os
rows = list(csv.DictReader(open(os.path.join(sys.argv[1], "run", "requests.csv"))))
ok = [r for r in rows if r["status"] == "200"]
lat = sorted(float(r["latency_ms"]) for r in ok)
dur = max(float(r["end_ts"]) for r in rows) - min(float(r["start_ts"]) for r in rows) if "end_ts" in rows[0] else 0
print(f"{len(ok)}/{len(rows)} ok, p50 {lat[len(lat) // 2]:.0f} ms, p99 {lat[int(len(lat) * 0.99)]:.0f} ms" + (f", {len(ok) / dur:.0f} requests/s" if dur else ""))
sys.exit(0 if len(ok) == len(rows) else 1)
PY
rc2=$?
python -c "
import json, urllib.request
m = json.loads(urllib.request.urlopen('http://127.0.0.1:8131/info', timeout=5).read())['engine']
print({k: m[k] for k in ('steps', 'pipelined_steps', 'step_ms_ewma', 'prefix_cache_hits', 'running', 'waiting', 'preemptions')})"
kill $PID; wait $PID 2>/dev/null
rm -f "$OUT/run/requests.csv"; rm -rf "$OUT"/run/traces*
[ $rc -eq 0 ] && [ $rc2 -eq 0 ]

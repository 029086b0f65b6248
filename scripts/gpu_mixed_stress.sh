#!/bin/bash
# Mixed traffic against one native server: greedy streaming, unseeded sampling, seeded sampling with
# logprobs, long prompts, 5000-token prompts that hit the prefix cache (their short tails go to the
# split-KV / split-tile attention forms, whose graphs are captured while other steps are in flight)
# guided JSON, and penalised requests with a logit bias and a minimum length — all at once.  Every request must succeed.
set -o pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"
export PYTHONPATH="$R" KVMINI_CLUSTER_BACKEND=local DISABLE_IO_PROBE=1
OUT="$R/gpurun_out/stress"
rm -rf "$OUT"; mkdir -p "$OUT"
python -m kvmini serve --port 8130 --service stress --model llama-3-8b --max-num-seqs 48 --max-model-len 8192 --no-register > "$OUT/server.log" 2>&1 &
PID=$!
for _ in $(seq 1 600); do
  kill -0 $PID 2>/dev/null || { tail -20 "$OUT/server.log"; exit 1; }
  python -c "import urllib.request; urllib.request.urlopen('http://127.0.0.1:8130/health', timeout=1)" 2>/dev/null && break
  sleep 0.5
done
echo '{"seed": 11, "logprobs": true, "top_logprobs": 2}' > "$OUT/extra.json"
LT="python -m kvmini loadtest --url http://127.0.0.1:8130 --model llama-3-8b --quiet --no-trace"
timeout -k 10 200 $LT --requests 400 --concurrency 12 --max-tokens 48 --target-rps 1000 --run-dir "$OUT/greedy" > "$OUT/greedy.log" 2>&1 & P1=$!
timeout -k 10 200 $LT --requests 200 --concurrency 6 --max-tokens 40 --target-rps 1000 --temperature 0.9 --top-p 0.9 --top-k 40 --run-dir "$OUT/sampled" > "$OUT/sampled.log" 2>&1 & P2=$!
timeout -k 10 200 $LT --requests 60 --concurrency 2 --max-tokens 24 --target-rps 5 --pattern poisson --no-stream --temperature 0.7 --extra-openai-json "$OUT/extra.json" --run-dir "$OUT/seeded" > "$OUT/seeded.log" 2>&1 & P3=$!
timeout -k 10 200 $LT --requests 40 --concurrency 2 --max-tokens 16 --target-rps 4 --prompt-tokens 700 --run-dir "$OUT/long" > "$OUT/long.log" 2>&1 & P4=$!
echo '{"response_format": {"type": "json_object"}}' > "$OUT/json.json"
timeout -k 10 200 $LT --requests 40 --concurrency 2 --max-tokens 8 --target-rps 4 --prompt-tokens 5000 --run-dir "$OUT/deep" > "$OUT/deep.log" 2>&1 & P5=$!
timeout -k 10 200 $LT --requests 60 --concurrency 3 --max-tokens 48 --target-rps 10 --no-stream --extra-openai-json "$OUT/json.json" --run-dir "$OUT/guided" > "$OUT/guided.log" 2>&1 & P6=$!
echo '{"frequency_penalty": 0.8, "presence_penalty": 0.4, "repetition_penalty": 1.1, "logit_bias": {"1000": -100, "2000": 5}, "min_tokens": 8}' > "$OUT/pen.json"
timeout -k 10 200 $LT --requests 120 --concurrency 4 --max-tokens 32 --target-rps 1000 --temperature 0.8 --extra-openai-json "$OUT/pen.json" --run-dir "$OUT/penalised" > "$OUT/penalised.log" 2>&1 & P7=$!
rc=0
for p in $P1 $P2 $P3 $P4 $P5 $P6 $P7; do wait $p || rc=1; done
python - "$OUT" <<'PY'
import csv, sys, os
bad = 0
for name in ("greedy", "sampled", "seeded", "long", "deep", "guided", "penalised"):
    rows = list(csv.DictReader(open(os.path.join(sys.argv[1], name, "requests.csv"))))
    ok = sum(1 for r in rows if r["status"] == "200")
    lat = sorted(float(r["latency_ms"]) for r in rows if r["status"] == "200")
    print(f"{name:8s} {ok}/{len(rows)} ok, p50 {lat[len(lat) // 2]:.0f} ms, max {lat[-1]:.0f} ms")
    bad += len(rows) - ok
sys.exit(1 if bad else 0)
PY
rc2=$?
python -c "
import json, urllib.request
m = json.loads(urllib.request.urlopen('http://127.0.0.1:8130/info', timeout=5).read())['engine']
print({k: m[k] for k in ('steps', 'pipelined_steps', 'variant_graphs', 'variant_graph_steps', 'prefix_cache_hits', 'running', 'waiting', 'preemptions', 'kv_blocks_free', 'kv_blocks_total')})"
kill $PID; wait $PID 2>/dev/null
[ $rc -eq 0 ] && [ $rc2 -eq 0 ]

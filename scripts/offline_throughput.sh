#!/bin/bash
# Offline throughput of the native engine (`kvmini generate`): N short prompts, T tokens each, all in flight at once.
set -o pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"
export PYTHONPATH="$R"
OUT="$R/gpurun_out/offline"; mkdir -p "$OUT"
N="${N:-256}"; T="${T:-128}"
python - "$N" > "$OUT/prompts.txt" <<'PY'
import sys
for i in range(int(sys.argv[1])):
    print(f"Request {i}: tell me about item {i * 7919 % 1000} in a few words.")
PY
: > "$OUT/summary.jsonl"
for q in ${QUANTS:-none fp8 mxfp4}; do
  QF=""; [ "$q" != none ] && QF="--quantization $q"
  timeout -k 10 200 python -m kvmini generate --model llama-3-8b --prompts-file "$OUT/prompts.txt" --max-tokens "$T" --max-num-seqs "$N" --max-batched-tokens 2048 --max-model-len 2048 $QF $EXTRA --out "$OUT/gen_$q.jsonl" | tail -1 | sed "s/^{/{\"quantization\": \"$q\", /" >> "$OUT/summary.jsonl" || exit 1
  rm -f "$OUT/gen_$q.jsonl"
done
cat "$OUT/summary.jsonl"

#!/bin/bash
# Backend adapter: drive load at a vllm-rocm endpoint with the shared load engine
# (protocol plug-in: openai), so every backend yields the canonical requests.csv /
# meta.json / traces and honours --pattern.
# Usage: invoke.sh --url URL --requests N --concurrency C --pattern P --max-tokens T --streaming true|false --run-dir DIR [--model NAME]
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/../../.." && pwd)"
ARGS=(--protocol openai) ; STREAMING=true
while [[ $# -gt 0 ]]; do
  case "$1" in
    --streaming) STREAMING="$2"; shift 2 ;;
    *) ARGS+=("$1"); shift ;;
  esac
done
[[ "$STREAMING" == "true" ]] || ARGS+=(--no-stream)
exec env PYTHONPATH="$ROOT${PYTHONPATH:+:$PYTHONPATH}" python3 -m kvmini loadtest "${ARGS[@]}"

#!/bin/bash
# Backend adapter: deploy MODEL as an InferenceService served by tgi-rocm.
# Usage: deploy.sh --model NAME [--namespace NS] [--model-uri URI] [--tensor-parallel-size N] [--dry-run]
# Thin shim over `kvmini deploy --backend tgi-rocm`: the manifest is built in
# kvmini/deploy/manifests.py (amd.com/gpu resources, RCCL /dev/shm volume for TP > 1).
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/../../.." && pwd)"
MODEL="" ; NS="ml-prod" ; EXTRA=()
while [[ $# -gt 0 ]]; do
  case "$1" in
    --model) MODEL="$2"; shift 2 ;;
    --namespace) NS="$2"; shift 2 ;;
    --streaming) shift 2 ;;  # request-side option; accepted for adapter symmetry
    *) EXTRA+=("$1"); shift ;;
  esac
done
[[ -n "$MODEL" ]] || { echo "--model is required" >&2; exit 1; }
exec env PYTHONPATH="$ROOT${PYTHONPATH:+:$PYTHONPATH}" python3 -m kvmini deploy --backend tgi-rocm --service "$MODEL" --model-name "$MODEL" --namespace "$NS" "${EXTRA[@]}"

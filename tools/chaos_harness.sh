#!/usr/bin/env bash
# Thin shim: `./tools/chaos_harness.sh ...` == `python -m kvmini chaos ...`.
set -euo pipefail
cd "$(dirname "${BASH_SOURCE[0]}")/.."
exec "${PYTHON:-python3}" -m kvmini chaos "$@"

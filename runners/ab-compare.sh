#!/usr/bin/env bash
# Thin shim: `./runners/ab-compare.sh ...` == `python -m kvmini ab-compare ...`.
set -euo pipefail
cd "$(dirname "${BASH_SOURCE[0]}")/.."
exec "${PYTHON:-python3}" -m kvmini ab-compare "$@"

#!/usr/bin/env bash
# Thin shim: `./scripts/preflight-checks.sh ...` == `python -m kvmini deploy --dry-run ...`.
set -euo pipefail
cd "$(dirname "${BASH_SOURCE[0]}")/.."
exec "${PYTHON:-python3}" -m kvmini preflight "$@"

#!/usr/bin/env bash
# Thin shim: `./tools/sign.sh ...` == `python -m kvmini sign ...`.
set -euo pipefail
cd "$(dirname "${BASH_SOURCE[0]}")/.."
exec "${PYTHON:-python3}" -m kvmini sign "$@"

#!/usr/bin/env bash
# Thin shim: `./tools/bundle_run.sh ...` == `python -m kvmini bundle ...`.
set -euo pipefail
cd "$(dirname "${BASH_SOURCE[0]}")/.."
exec "${PYTHON:-python3}" -m kvmini bundle "$@"

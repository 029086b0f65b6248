#!/usr/bin/env bash
# Thin shim: `./grid-sweep.sh ...` == `python -m kvmini grid-sweep ...` (same flags as the kvmini subcommand).
set -euo pipefail
cd "$(dirname "${BASH_SOURCE[0]}")"
exec "${PYTHON:-python3}" -m kvmini grid-sweep "$@"

#!/usr/bin/env bash
# Thin shim: `./sweeps/partition-sweep.sh ...` == `python -m kvmini partition-sweep ...`.
set -euo pipefail
cd "$(dirname "${BASH_SOURCE[0]}")/.."
exec "${PYTHON:-python3}" -m kvmini partition-sweep "$@"

#!/usr/bin/env bash
# Thin shim: `./sweeps/autoscale-sweep.sh ...` == `python -m kvmini autoscale-sweep ...`.
set -euo pipefail
cd "$(dirname "${BASH_SOURCE[0]}")/.."
exec "${PYTHON:-python3}" -m kvmini autoscale-sweep "$@"

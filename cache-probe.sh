#!/usr/bin/env bash
# Thin shim: `./cache-probe.sh ...` == `python -m kvmini cache-probe ...` (same flags as the kvmini subcommand).
set -euo pipefail
cd "$(dirname "${BASH_SOURCE[0]}")"
exec "${PYTHON:-python3}" -m kvmini cache-probe "$@"

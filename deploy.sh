#!/usr/bin/env bash
# Thin shim: `./deploy.sh ...` == `python -m kvmini deploy ...` (same flags as the kvmini subcommand).
set -euo pipefail
cd "$(dirname "${BASH_SOURCE[0]}")"
exec "${PYTHON:-python3}" -m kvmini deploy "$@"

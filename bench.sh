#!/usr/bin/env bash
# Thin shim: `./bench.sh ...` == `python -m kvmini bench ...` (same flags as the kvmini subcommand).
set -euo pipefail
cd "$(dirname "${BASH_SOURCE[0]}")"
exec "${PYTHON:-python3}" -m kvmini bench "$@"

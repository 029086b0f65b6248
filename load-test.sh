#!/usr/bin/env bash
# Thin shim: `./load-test.sh ...` == `python -m kvmini loadtest ...` (same flags as the kvmini subcommand).
set -euo pipefail
cd "$(dirname "${BASH_SOURCE[0]}")"
exec "${PYTHON:-python3}" -m kvmini loadtest "$@"

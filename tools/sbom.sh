#!/usr/bin/env bash
# Thin shim: `./tools/sbom.sh ...` == `python -m kvmini sbom ...`.
set -euo pipefail
cd "$(dirname "${BASH_SOURCE[0]}")/.."
exec "${PYTHON:-python3}" -m kvmini sbom "$@"

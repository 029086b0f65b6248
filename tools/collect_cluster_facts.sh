#!/usr/bin/env bash
# Thin shim: `./tools/collect_cluster_facts.sh ...` == `python -m kvmini cluster-facts ...`.
set -euo pipefail
cd "$(dirname "${BASH_SOURCE[0]}")/.."
exec "${PYTHON:-python3}" -m kvmini cluster-facts "$@"

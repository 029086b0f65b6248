"""``kvmini`` — one CLI for the whole harness.

The reference's six subcommands (``plan analyze report calculate deploy bench`` →
``subprocess`` into scripts, ``kvmini/cli.py:41-117``) are kept with the same
names; here every subcommand is an in-process call into the package, and the
tools that were only reachable as loose scripts get subcommands too.
Arguments after the subcommand are passed to that tool unchanged.
"""

from __future__ import annotations

import importlib
import sys
from typing import Dict, List, Optional, Tuple

from kvmini import __version__

# subcommand -> (module, function, help)
COMMANDS: Dict[str, Tuple[str, str, str]] = {
    # the reference's six
    "plan": ("kvmini.cost.planner", "main", "capacity & budget planner"),
    "analyze": ("kvmini.analysis.reduce", "main", "reduce a run directory into results.json"),
    "report": ("kvmini.report.generator", "main", "HTML reports (single run / grid sweep / partition matrix)"),
    "calculate": ("kvmini.cost.calculator", "main", "legacy latency-based cost calculator"),
    "deploy": ("kvmini.deploy.cli", "main", "deploy an InferenceService or a local server"),
    "bench": ("kvmini.harness.bench", "main", "full pipeline: validate -> load test -> analyze -> cost"),
    # pipeline stages
    "loadtest": ("kvmini.loadgen.runner", "main", "load generator (OpenAI / TGI / KServe-v2)"),
    "cost": ("kvmini.cost.estimator", "main", "resource-seconds cost estimator"),
    "energy": ("kvmini.telemetry.energy", "main", "collect / integrate GPU energy (amd-smi)"),
    "validate": ("kvmini.gates.validate_config", "main", "validate a profile / configuration"),
    "preflight": ("kvmini.deploy.preflight", "main", "kubectl / KServe / GPU-node pre-flight checks"),
    "probe": ("kvmini.harness.net_storage_probe", "main", "network / storage probe"),
    "bundle": ("kvmini.harness.bundle", "main", "bundle a run with provenance"),
    "cluster-facts": ("kvmini.harness.bundle", "main_facts", "collect cluster facts"),
    "sbom": ("kvmini.harness.bundle", "main_sbom", "SBOMs of the serving images"),
    "sign": ("kvmini.harness.bundle", "main_sign", "cosign the serving images"),
    # gates
    "gate": ("kvmini.gates.slo", "main", "SLO gate (exit 3 on violation)"),
    "canary": ("kvmini.gates.canary", "main", "canary comparison (exit 2 on regression)"),
    # harnesses
    "grid-sweep": ("kvmini.harness.sweeps", "main_grid", "concurrency x max_tokens x pattern sweep"),
    "autoscale-sweep": ("kvmini.harness.sweeps", "main_autoscale", "Knative autoscaling parameter sweep"),
    "partition-sweep": ("kvmini.harness.sweeps", "main_partition", "MI355X partition profile sweep"),
    "quant-sweep": ("kvmini.harness.quantization_sweep", "main", "quantization / decoding sweep with Pareto analysis"),
    "ab-compare": ("kvmini.harness.ab_compare", "main", "A/B backend comparison (vLLM-ROCm vs TGI-ROCm vs native)"),
    "compare-backends": ("kvmini.harness.compare_backends", "main", "backend comparison with JSON/HTML/CSV reports"),
    "reference-matrix": ("kvmini.harness.reference_matrix", "main", "GPU x model x traffic reference matrix"),
    "build-vs-perf": ("kvmini.harness.build_vs_perf", "main", "engine build time vs served performance"),
    "cache-probe": ("kvmini.harness.cache_probe", "main", "prefix-cache effectiveness probe"),
    "fairness": ("kvmini.harness.fairness", "main", "dual-tenant fairness / backpressure harness"),
    "parity": ("kvmini.harness.parity_probe", "main", "OpenAI API parity probe"),
    "chaos": ("kvmini.harness.chaos", "main", "fault injection + MTTR table"),
    "spec-delta": ("kvmini.harness.spec_delta", "main", "speculative decoding TTFT / inter-token delta report"),
    "quality": ("kvmini.harness.quality", "main", "quality mini-eval + Pareto bucket"),
    "telemetry": ("kvmini.parallel.telemetry_daemon", "main", "lock-step multi-GPU telemetry over RCCL"),
    "docs-site": ("kvmini.report.docsite", "main", "render README + docs/ into a static documentation site"),
    # servers
    "serve": ("kvmini.server.native", "main", "native MI355X OpenAI-compatible server"),
    "generate": ("kvmini.server.offline", "main", "offline batch generation on the native engine (no server)"),
    "router": ("kvmini.server.router", "main", "data-parallel front-end over N engine replicas"),
    "mock": ("kvmini.server.mock", "main", "deterministic mock LLM endpoint"),
}


def usage() -> str:
    lines = [f"kvmini {__version__} — MI355X-native LLM serving benchmark harness", "", "usage: kvmini <command> [args...]", "", "commands:"]
    for name, (_m, _f, h) in COMMANDS.items():
        lines.append(f"  {name:18} {h}")
    lines += ["", "`kvmini <command> --help` shows the command's flags."]
    return "\n".join(lines)


def main(argv: Optional[List[str]] = None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ("-h", "--help", "help"):
        print(usage())
        return 0
    if argv[0] in ("-V", "--version"):
        print(__version__)
        return 0
    cmd, rest = argv[0], argv[1:]
    if cmd not in COMMANDS:
        print(f"kvmini: unknown command '{cmd}'\n\n{usage()}", file=sys.stderr)
        return 2
    mod_name, fn_name, _ = COMMANDS[cmd]
    try:
        fn = getattr(importlib.import_module(mod_name), fn_name)
    except ImportError as e:
        print(f"kvmini {cmd}: cannot load {mod_name}: {e}", file=sys.stderr)
        return 2
    try:
        rc = fn(rest)
    except KeyboardInterrupt:
        print("Aborted", file=sys.stderr)
        return 130
    return int(rc or 0)


if __name__ == "__main__":
    sys.exit(main())

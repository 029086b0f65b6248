"""Energy collector / integrator (``kvmini energy collect|integrate``).

``collect`` samples GPU power at a fixed interval into ``power.json`` — from
amd-smi directly (default) or, with ``--prom-url``, from the ROCm metrics exporter
through Prometheus (the reference's only mode, against DCGM:
``energy/collector.py:203-251``).  ``integrate`` clips the samples to the request
window of ``requests.csv`` and integrates with the trapezoid rule, with the
reference's options and output keys (``energy/collector.py:254-381``):
``--include-warmup`` (whole sample span), ``--idle-tax baseline|series``,
``energy.json`` keys ``Wh_active, Wh_idle_tax, Wh_per_1k_tokens_active,
Wh_per_request_active, window, samples`` and ``--merge-results`` into results.json
as ``energy_wh_active, energy_wh_idle_tax, energy_wh_per_1k_tokens,
energy_wh_per_request``.  When several GPUs were sampled, ``per_gpu`` series are
integrated in one batched GPU launch (``csrc/energy_trapz.hip``) if available.
"""

from __future__ import annotations

import argparse
import json
import os
import signal
import sys
import time
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence, Tuple

from kvmini.utils import load_json, merge_json_file, write_json


@dataclass
class PowerSample:
    ts: float
    watts: Optional[float]


def trapezoidal_wh(samples: Sequence[PowerSample], t0: float, t1: float) -> float:
    """Wh over [t0, t1]: samples without a value or outside the window are
    dropped, the rest sorted by time and joined by trapezoids."""
    if not samples or t1 <= t0:
        return 0.0
    ss = sorted((s for s in samples if s.watts is not None and t0 <= s.ts <= t1), key=lambda s: s.ts)
    if len(ss) < 2:
        return 0.0
    wh = 0.0
    for a, b in zip(ss, ss[1:]):
        dt_h = (b.ts - a.ts) / 3600.0
        if dt_h > 0:
            wh += ((a.watts or 0.0) + (b.watts or 0.0)) / 2.0 * dt_h
    return wh


def load_power_samples(path: str) -> List[PowerSample]:
    data = load_json(path)
    items = data if isinstance(data, list) else data.get("samples", []) if isinstance(data, dict) else []
    out = []
    for it in items:
        w = it.get("watts")
        out.append(PowerSample(float(it.get("ts_s", it.get("ts", 0))), float(w) if w is not None else None))
    return out


def per_gpu_series(path: str) -> Dict[int, List[PowerSample]]:
    """Per-GPU series from a power.json written by the amd-smi sampler."""
    data = load_json(path)
    items = data.get("samples", []) if isinstance(data, dict) else data
    out: Dict[int, List[PowerSample]] = {}
    for it in items:
        ts = float(it.get("ts_s", it.get("ts", 0)))
        for g in it.get("per_gpu") or []:
            out.setdefault(int(g.get("index", 0)), []).append(PowerSample(ts, g.get("watts")))
    return out


def integrate_series_batched(series: Dict[int, List[PowerSample]], t0: float, t1: float, engine: str = "auto") -> Dict[int, float]:
    """Wh per GPU; one HIP launch for all series when the extension and a GPU are there."""
    if not series:
        return {}
    use_hip = False
    if engine in ("auto", "hip"):
        try:
            from kvmini.ops import gpu_available, native_built

            use_hip = native_built() and gpu_available()
        except Exception:
            use_hip = False
        if engine == "hip" and not use_hip:
            raise RuntimeError("HIP energy integration requested but unavailable")
    if not use_hip:
        return {g: trapezoidal_wh(s, t0, t1) for g, s in series.items()}
    import numpy as np
    import torch

    from kvmini.ops import native

    keys = sorted(series)
    smax = max(len(series[k]) for k in keys)
    ts = np.zeros((len(keys), smax))
    w = np.full((len(keys), smax), np.nan)
    n = np.zeros(len(keys), dtype=np.int32)
    for i, k in enumerate(keys):
        ss = sorted(series[k], key=lambda s: s.ts)
        n[i] = len(ss)
        ts[i, : len(ss)] = [s.ts for s in ss]
        w[i, : len(ss)] = [np.nan if s.watts is None else s.watts for s in ss]
    try:
        out = native().energy_trapz(torch.from_numpy(ts).cuda(), torch.from_numpy(w).cuda(), torch.from_numpy(n).cuda(), float(t0), float(t1)).cpu().numpy()
    except RuntimeError as e:
        if engine == "hip":
            raise
        # an analysis stage must not die because the GPU is busy or hidden from this process
        print(f"[energy] HIP integration unavailable ({e}); integrating on the host", file=sys.stderr)
        return {g: trapezoidal_wh(s, t0, t1) for g, s in series.items()}
    return {k: float(out[i, 0]) for i, k in enumerate(keys)}


def run_window_bounds(rows: List[Dict[str, Any]]) -> Tuple[float, float]:
    from kvmini.analysis.reduce import window_bounds

    return window_bounds(rows)


def collect_power(args: argparse.Namespace) -> int:
    if not args.out:
        print("ERROR: --out required to write power samples", file=sys.stderr)
        return 2
    interval = max(0.05 if not args.prom_url else 0.5, float(args.interval or 1.0))
    duration = float(args.duration) if args.duration else 0.0
    stop = {"v": False}

    def _sigint(_s: int, _f: Any) -> None:
        stop["v"] = True

    signal.signal(signal.SIGINT, _sigint)
    signal.signal(signal.SIGTERM, _sigint)
    samples: List[Dict[str, Any]] = []
    if args.prom_url:
        from .prometheus import power_instant

        def read() -> Optional[Dict[str, Any]]:
            return {"ts_s": time.time(), "watts": power_instant(args.prom_url, args.namespace, args.service)}

    else:
        from .amdsmi_sampler import FakeSampler, read_once

        src = FakeSampler() if args.fake else read_once
        gpus = [int(x) for x in args.gpus.split(",")] if args.gpus else None

        def read() -> Optional[Dict[str, Any]]:
            return src(gpus)

    start = time.time()
    while True:
        s = read()
        if s is None:
            s = {"ts_s": time.time(), "watts": None}
        samples.append(s)
        if args.verbose:
            print(json.dumps({"ts_s": s["ts_s"], "watts": s.get("watts")}))
        if stop["v"] or (duration and time.time() - start >= duration):
            break
        time.sleep(interval)
    write_json(args.out, {"samples": samples, "interval_s": interval, "source": "prometheus" if args.prom_url else "amd-smi"})
    print(f"Wrote power samples: {args.out} ({len(samples)} points)")
    return 0


def integrate_energy(args: argparse.Namespace) -> int:
    from kvmini.analysis.coldwarm import is_success
    from kvmini.analysis.reduce import read_requests_csv

    run_dir = args.run_dir
    req_csv = os.path.join(run_dir, "requests.csv")
    power_path = args.power or os.path.join(run_dir, "power.json")
    energy_out = os.path.join(run_dir, "energy.json")
    if not os.path.exists(req_csv):
        print(f"ERROR: requests.csv not found at {req_csv}", file=sys.stderr)
        return 2
    if not os.path.exists(power_path):
        print(f"WARNING: power.json not found at {power_path}; energy fields will be null", file=sys.stderr)
        write_json(energy_out, {"Wh_active": None, "Wh_idle_tax": None, "Wh_per_1k_tokens_active": None, "Wh_per_request_active": None, "warnings": ["Missing power samples"]})
        if args.merge_results:
            merge_json_file(os.path.join(run_dir, "results.json"), {"energy_wh_active": None, "energy_wh_idle_tax": None, "energy_wh_per_1k_tokens": None, "energy_wh_per_request": None})
        return 0

    rows = read_requests_csv(req_csv)
    t0, t1 = run_window_bounds(rows)
    samples = load_power_samples(power_path)
    if args.include_warmup and samples:
        t0, t1 = min(s.ts for s in samples), max(s.ts for s in samples)
    if t1 <= t0:
        print("ERROR: Invalid integration window", file=sys.stderr)
        return 2
    wh_active = trapezoidal_wh(samples, t0, t1)

    wh_idle: Optional[float] = None
    if args.idle_tax == "series" and samples:
        smin, smax = min(s.ts for s in samples), max(s.ts for s in samples)
        before = trapezoidal_wh(samples, smin, min(t0, smax)) if smin < t0 else 0.0
        after = trapezoidal_wh(samples, max(smin, t1), smax) if smax > t1 else 0.0
        wh_idle = before + after
    elif args.idle_tax == "baseline":
        outside = [s.watts for s in samples if s.watts is not None and (s.ts < t0 or s.ts > t1)]
        if outside:
            p_idle = sorted(outside)[len(outside) // 2]
            if len(samples) > 1:
                steps = [b.ts - a.ts for a, b in zip(samples, samples[1:])]
                avg_step = sum(steps) / len(steps)
            else:
                avg_step = t1 - t0
            n_out = sum(1 for s in samples if s.ts < t0 or s.ts > t1)
            wh_idle = p_idle * (n_out * avg_step) / 3600.0

    ok = [r for r in rows if is_success(r)]
    total_tokens = sum(float(r["total_tokens"]) for r in ok if isinstance(r.get("total_tokens"), (int, float)))
    energy: Dict[str, Any] = {
        "Wh_active": wh_active,
        "Wh_idle_tax": wh_idle,
        "Wh_per_1k_tokens_active": wh_active / total_tokens * 1000.0 if total_tokens > 0 else None,
        "Wh_per_request_active": wh_active / len(ok) if ok else None,
        "window": {"start": t0, "end": t1},
        "samples": os.path.basename(power_path),
    }
    per_gpu = per_gpu_series(power_path)
    if per_gpu:
        energy["Wh_active_per_gpu"] = {str(k): v for k, v in integrate_series_batched(per_gpu, t0, t1).items()}
    write_json(energy_out, energy)
    print(json.dumps(energy, indent=2))
    if args.merge_results:
        merge_json_file(
            os.path.join(run_dir, "results.json"),
            {
                "energy_wh_active": energy["Wh_active"],
                "energy_wh_idle_tax": energy["Wh_idle_tax"],
                "energy_wh_per_1k_tokens": energy["Wh_per_1k_tokens_active"],
                "energy_wh_per_request": energy["Wh_per_request_active"],
            },
        )
        print(f"Merged energy fields into {os.path.join(run_dir, 'results.json')}")
    return 0


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(prog="kvmini energy", description="Energy collector and integrator (amd-smi / ROCm exporter)")
    sub = ap.add_subparsers(dest="cmd", required=True)
    c = sub.add_parser("collect", help="sample GPU power into power.json")
    c.add_argument("--out", required=True)
    c.add_argument("--interval", type=float, default=1.0)
    c.add_argument("--duration", type=float, default=0.0, help="seconds (0 = until SIGINT)")
    c.add_argument("--gpus", default=None, help="comma-separated GPU indices (default: all)")
    c.add_argument("--prom-url", default=None, help="use the ROCm exporter through Prometheus instead of amd-smi")
    c.add_argument("--namespace", default="default")
    c.add_argument("--service", default="demo-llm")
    c.add_argument("--fake", action="store_true", help="synthetic power source (tests)")
    c.add_argument("--verbose", action="store_true")
    i = sub.add_parser("integrate", help="integrate power.json over the run window")
    i.add_argument("--run-dir", required=True)
    i.add_argument("--power", default=None)
    i.add_argument("--include-warmup", action="store_true")
    i.add_argument("--idle-tax", choices=["none", "baseline", "series"], default="none")
    i.add_argument("--merge-results", action="store_true")
    args = ap.parse_args(argv)
    return collect_power(args) if args.cmd == "collect" else integrate_energy(args)


if __name__ == "__main__":
    sys.exit(main())

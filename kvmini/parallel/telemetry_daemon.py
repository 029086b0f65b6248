"""Per-rank GPU telemetry daemon (``kvmini telemetry``).

One process per GPU (launched with ``torch.distributed.run`` next to a serving
job, or standalone on a single GPU): each rank samples its own device through
amdsmi, and every ``--gather-every`` seconds all ranks exchange their latest
128-byte snapshot with ONE ``all_gather`` (RCCL over xGMI on GPUs, gloo on CPU),
so rank 0 holds the whole node's table without N Prometheus scrapes.  Rank 0
appends the table to ``power.json`` in the collector's format — the energy
integrator and the analyzer consume it unchanged — and can serve it as
Prometheus text on ``--metrics-port`` using the ROCm exporter's metric names, so
the Grafana dashboards work against a node that runs no exporter.
"""

from __future__ import annotations

import argparse
import os
import signal
import sys
import threading
import time
from typing import Any, Dict, List, Optional

import numpy as np

from kvmini.utils import write_json


def _prom_text(table: Optional[np.ndarray]) -> str:
    """Prometheus exposition of the latest ``[world, 16]`` snapshot table with
    the ROCm exporter's metric names (VRAM in MiB, as the exporter reports it)."""
    if table is None:
        return ""
    spec = [("gpu_gfx_activity", 4, 1.0), ("gpu_package_power", 3, 1.0), ("gpu_used_vram", 5, 1.0 / (1 << 20)), ("gpu_total_vram", 6, 1.0 / (1 << 20)), ("gpu_edge_temperature", 7, 1.0), ("gpu_clock", 8, 1.0)]
    lines = []
    for name, col, scale in spec:
        lines.append(f"# TYPE {name} gauge")
        for row in table:
            v = row[col]
            if not np.isnan(v):
                lines.append(f'{name}{{gpu_id="{int(row[0])}"}} {float(v) * scale:.3f}')
    return "\n".join(lines) + "\n"


def serve_metrics(port: int, latest: Dict[str, Any]) -> threading.Thread:
    import http.server

    class H(http.server.BaseHTTPRequestHandler):
        def do_GET(self) -> None:  # noqa: N802
            body = _prom_text(latest.get("table")).encode()
            self.send_response(200)
            self.send_header("Content-Type", "text/plain; version=0.0.4")
            self.send_header("Content-Length", str(len(body)))
            self.end_headers()
            self.wfile.write(body)

        def log_message(self, *a: Any) -> None:
            pass

    srv = http.server.ThreadingHTTPServer(("0.0.0.0", port), H)
    t = threading.Thread(target=srv.serve_forever, daemon=True)
    t.start()
    return t


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(prog="kvmini telemetry", description="Per-rank amdsmi sampler with collective aggregation")
    ap.add_argument("--interval", type=float, default=0.2, help="local sampling period (s)")
    ap.add_argument("--gather-every", type=float, default=1.0, help="collective aggregation period (s)")
    ap.add_argument("--duration", type=float, default=0.0, help="stop after this many seconds (0 = until SIGTERM)")
    ap.add_argument("--out", default="power.json")
    ap.add_argument("--metrics-port", type=int, default=0, help="serve the node table as Prometheus text on rank 0")
    ap.add_argument("--backend", default=None, choices=[None, "nccl", "gloo"], help="collective backend (default: nccl on GPUs, gloo otherwise)")
    ap.add_argument("--fake", action="store_true", help="synthetic sampler (no GPU needed)")
    ap.add_argument("--gather-impl", default="torch", choices=["torch", "rccl", "p2p"],
                    help="how the per-tick snapshots cross the node: torch.distributed all-gather (default; gloo on CPU), the C++ ncclAllGather on its own communicator (rccl), "
                         "or the one-shot xGMI peer-memory kernel (p2p); rccl and p2p need one GPU per rank")
    a = ap.parse_args(argv)

    import torch
    import torch.distributed as dist

    from kvmini.parallel.telemetry_gather import TelemetryGather, table_to_power_json
    from kvmini.telemetry.amdsmi_sampler import FakeSampler, TelemetrySampler

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    use_gpu = torch.cuda.is_available() and not a.fake
    backend = a.backend or ("nccl" if use_gpu else "gloo")
    if world > 1 and not dist.is_initialized():
        if use_gpu:
            torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    device = torch.device("cuda", local_rank) if use_gpu and backend == "nccl" else torch.device("cpu")

    source = FakeSampler(base_w=400.0 + 25.0 * rank) if (a.fake or not use_gpu) else None
    sampler = TelemetrySampler(interval_s=a.interval, gpu_indices=None if source else [local_rank], source=source).start()
    native = a.gather_impl in ("rccl", "p2p")
    if native and not (use_gpu and backend == "nccl"):
        ap.error("--gather-impl rccl / p2p needs GPUs (one rank per GPU)")
    gather = TelemetryGather(device=device, native=native, device_index=local_rank, impl=a.gather_impl if native else "rccl")
    latest: Dict[str, Any] = {"table": None}
    gather_us: List[float] = []
    if rank == 0 and a.metrics_port:
        serve_metrics(a.metrics_port, latest)

    stop = threading.Event()
    for sig in (signal.SIGTERM, signal.SIGINT):
        try:
            signal.signal(sig, lambda *_: stop.set())
        except ValueError:
            pass
    # ticks are counted, not timed, so every rank issues the same number of collectives
    n_ticks = max(1, int(round(a.duration / a.gather_every))) if a.duration > 0 else 0
    tables: List[np.ndarray] = []

    def flush() -> None:
        if rank == 0 and tables:
            doc = table_to_power_json(np.stack(tables, axis=1), a.gather_every, source=gather.source)
            doc["dropped_ticks"] = gather.dropped_ticks
            if gather_us:
                srt = sorted(gather_us)
                doc["gather_impl"] = a.gather_impl
                doc["gather_us_p50"], doc["gather_us_p95"] = srt[len(srt) // 2], srt[min(len(srt) - 1, int(0.95 * len(srt)))]
            write_json(a.out, doc)

    ticks = 0
    try:
        while n_ticks == 0 or ticks < n_ticks:
            # a SIGTERM on one rank must end all ranks at the same tick: share the flag
            flag = torch.tensor([1.0 if stop.is_set() else 0.0], device=device)
            if world > 1:
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if flag.item() > 0:
                break
            sample = sampler.samples[-1] if sampler.samples else None
            t_g = time.perf_counter()
            table = gather.gather_tick(sample)
            gather_us.append((time.perf_counter() - t_g) * 1e6)
            ticks += 1
            if table is not None and rank == 0:
                if np.isnan(table[0, 2]):
                    table[:, 2] = time.time()
                latest["table"] = table
                tables.append(table)
                if ticks % 10 == 0:
                    flush()
            stop.wait(a.gather_every)
    finally:
        sampler.stop()
        flush()
        if rank == 0:
            print(f"telemetry: {ticks} gather ticks x {world} ranks -> {a.out}")
        if world > 1 and dist.is_initialized():
            dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())

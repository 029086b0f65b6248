#!/usr/bin/env python3
"""Headline benchmark: Llama-3-8B (bf16, random-init) served by the native MI355X
engine, 500 requests at concurrency 20, 64 max tokens, streaming — the request
mix of the reference's headline example (README.md:66-76): p95 latency, TTFT,
RPS and $/1k tokens.

One *step* = one complete pass of that mix (``--requests`` requests through the
OpenAI endpoint, closed at ``--concurrency`` in flight) against one model replica
per GPU.  With N GPUs every rank (one process per GPU, launched by torchrun) runs
its own replica + load generator (data parallel, weak scaling: per-GPU work is
fixed), per-request samples and lock-step amd-smi telemetry are all-gathered over
RCCL, and rank 0 reduces them (p50/p95/p99 by the HIP histogram/quantile kernel)
and prints ONE JSON line.  ``value`` is the aggregate request throughput (RPS)
over all GPUs; p95 / TTFT / $ per 1k tokens / Wh per 1k tokens ride along in the
same line with their own ratios against the baseline numbers.

Arrival pacing: the reference's "steady" pattern offers 2 x concurrency = 40 RPS,
which caps measured throughput below its own published 47.2 RPS; the default here
is ``--target-rps 0`` = saturating arrivals (all requests due immediately, 20 in
flight at all times), i.e. the same mix at the highest load concurrency 20 allows.
``--target-rps 40`` reproduces the reference pacing.

Launching: ``python bench.py --gpus N`` starts the N rank processes itself (a parent
that never imports torch or touches HIP spawns N fresh children, each with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in its environment, relays rank 0's JSON line and
fails if any child fails); under ``python -m torch.distributed.run ... bench.py --gpus N``
the ranks already exist and the script runs as one of them.  Either way a rank whose world
size differs from ``--gpus`` exits 2, and ``rccl_ranks`` in the JSON line is the sum of a
``1`` per rank over a real all-reduce (RCCL on GPUs, gloo in the CPU rehearsal).
"""

from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time
from typing import Any, Dict, List

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BASELINE = {"rps": 47.2, "p95_ms": 342.1, "ttft_ms": 23.4, "cost_per_1k_tokens": 0.0023, "wh_per_1k_tokens": 15.7}


class LoadChild:
    """Persistent load-generator process (see kvmini/loadgen/worker.py)."""

    def __init__(self) -> None:
        env = dict(os.environ)
        env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
        env["OMP_NUM_THREADS"] = "1"
        # the child never touches the GPU
        env["HIP_VISIBLE_DEVICES"] = ""
        env["CUDA_VISIBLE_DEVICES"] = ""
        self.p = subprocess.Popen([sys.executable, "-u", "-m", "kvmini.loadgen.worker"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, env=env, cwd=ROOT)

    def call(self, cmd: Dict[str, Any]) -> Dict[str, Any]:
        assert self.p.stdin and self.p.stdout
        self.p.stdin.write(json.dumps(cmd) + "\n")
        self.p.stdin.flush()
        line = self.p.stdout.readline()
        if not line:
            raise RuntimeError("load generator child died")
        out = json.loads(line)
        if not out.get("ok"):
            raise RuntimeError(f"load generator failed: {out.get('error')}")
        return out

    def close(self) -> None:
        try:
            if self.p.stdin:
                self.p.stdin.write(json.dumps({"cmd": "quit"}) + "\n")
                self.p.stdin.flush()
            self.p.wait(10)
        except Exception:
            self.p.kill()


def _free_port() -> int:
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return int(sk.getsockname()[1])


def self_launch(n: int, argv: List[str], isolate: bool) -> int:
    """Run ``n`` ranks of this script as fresh child processes (no torch, no HIP in this
    parent).  Children see every GPU and pick ``LOCAL_RANK`` (what torchrun does: RCCL's
    peer-to-peer transport over xGMI needs the peers' devices visible); ``isolate`` gives
    each child only its own GPU through HIP_VISIBLE_DEVICES instead."""
    port = _free_port()
    procs: List[subprocess.Popen] = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
        if isolate:
            env["HIP_VISIBLE_DEVICES"] = str(r)
            env["LOCAL_RANK"] = "0"
        else:
            env["LOCAL_RANK"] = str(r)
        # rank 0 owns stdout (the ONE JSON line); the other ranks' stdout goes to stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, cwd=ROOT, stdout=None if r == 0 else sys.stderr))
    rc = 0
    try:
        deadline = None
        pending = set(range(n))
        while pending:
            for r in list(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0:
                    print(f"bench.py: rank {r} exited with {code}", file=sys.stderr)
                    rc = rc or code
                    deadline = deadline or time.time() + 30.0  # the others cannot finish a collective without it
            if deadline is not None and time.time() > deadline:
                break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="llama-3-8b")
    ap.add_argument("--requests", type=int, default=500)
    ap.add_argument("--concurrency", type=int, default=20)
    ap.add_argument("--max-tokens", type=int, default=64)
    ap.add_argument("--pattern", default="steady")
    ap.add_argument("--target-rps", type=float, default=0.0, help="0 = saturate (closed loop at --concurrency)")
    ap.add_argument("--kernels", default="hip", choices=["hip", "torch"])
    ap.add_argument("--no-graphs", action="store_true")
    ap.add_argument("--quantization", default=None, choices=["fp8", "mxfp4"], help="weight-only FP8 (e4m3) or MXFP4 (NOT the headline configuration: the headline is bf16)")
    ap.add_argument("--speculative-model", default=None, help="draft preset for speculative decoding (not the headline configuration)")
    ap.add_argument("--num-speculative-tokens", type=int, default=2)
    ap.add_argument("--max-num-seqs", type=int, default=32)
    ap.add_argument("--max-batched-tokens", type=int, default=256)
    ap.add_argument("--max-model-len", type=int, default=2048)
    ap.add_argument("--no-async-scheduling", action="store_true", help="one step at a time: read the sampled ids back before launching the next step")
    ap.add_argument("--launch-margin-ms", type=float, default=None, help="pipelined stepping: launch this long before the step in flight should end (engine default when omitted; 0 = at once)")
    ap.add_argument("--ride-along-limit", type=int, default=None, help="prompt backlog (tokens) up to which prompts ride along with running decodes in <= 64-token steps (engine default when omitted; 0 = off)")
    ap.add_argument("--tune-gemm", action="store_true", help=argparse.SUPPRESS)  # accepted for old command lines: no library GEMM is left to tune
    ap.add_argument("--out-dir", default=None, help="write run artifacts (requests.csv, power.json, results.json) here")
    ap.add_argument("--gpu-hourly", type=float, default=None, help="$/GPU-hour (default: cost.yaml MI355X entry)")
    ap.add_argument("--device", default="cuda", choices=["cuda", "cpu"], help="cpu = rehearsal of the multi-rank flow over gloo with the reference ops (not a benchmark)")
    ap.add_argument("--isolate-devices", action="store_true", help="self-launch only: give every rank just its own GPU (HIP_VISIBLE_DEVICES=<rank>) instead of all GPUs + LOCAL_RANK")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return self_launch(args.gpus, sys.argv[1:], args.isolate_devices)

    # the host driver offers dmabuf IPC only: RCCL's peer-memory exchange between the ranks' processes needs this
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}: launch with --nproc-per-node {args.gpus} (or plain `python bench.py --gpus {args.gpus}`)", file=sys.stderr)
        return 2
    on_gpu = args.device == "cuda"
    if on_gpu and not torch.cuda.is_available():
        print("bench.py needs a ROCm GPU (use --device cpu only to rehearse the multi-rank flow)", file=sys.stderr)
        return 2
    if on_gpu:
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    else:
        device = torch.device("cpu")
        args.kernels, args.no_graphs = "torch", True
    # `cdev`: where the tensors of the result reductions live (a few KiB per rank).  On GPUs that is the device and
    # the collectives are RCCL's over xGMI; when RCCL does not come up on this node (initialisation or the first
    # all-reduce raises) every rank falls back to gloo with host tensors, says so in the JSON line, and the
    # measurement goes ahead: the replicas do not talk to each other while they serve.
    cdev = device
    collective_note = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        want = os.environ.get("KVMINI_BENCH_COLLECTIVES", "rccl")  # "gloo": skip RCCL; "force-rccl": try it even in the CPU rehearsal (tests the fallback)
        if (on_gpu and want != "gloo") or want == "force-rccl":
            try:
                dist.init_process_group("nccl", **({"device_id": device} if on_gpu else {}))  # RCCL over xGMI
                probe = torch.ones(1, dtype=torch.int32, device=device)
                dist.all_reduce(probe)
                if on_gpu:
                    torch.cuda.synchronize(device)
                if int(probe.item()) != world:
                    raise RuntimeError(f"probe all-reduce returned {int(probe.item())}, expected {world}")
            except Exception as e:  # noqa: BLE001 - whatever RCCL raises, the fallback is the same
                collective_note = f"rccl unavailable ({type(e).__name__}: {str(e)[:120]}); reductions over gloo"
                print(f"bench.py rank {rank}: {collective_note}", file=sys.stderr)
                if dist.is_initialized():
                    dist.destroy_process_group()
                os.environ["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 1)  # the first store may still hold the port
                dist.init_process_group("gloo")
                cdev = torch.device("cpu")
        else:
            dist.init_process_group("gloo")
            cdev = torch.device("cpu")

    from kvmini.engine.engine import EngineConfig, LLMEngine
    from kvmini.server.native import start_native_in_thread
    from kvmini.telemetry.native_rt import make_sampler

    proc_start = time.time()
    engine = LLMEngine(
        EngineConfig(
            model=args.model,
            device=str(device),
            dtype="bf16" if on_gpu else "fp32",
            seed=rank,
            max_num_seqs=args.max_num_seqs,
            max_batched_tokens=args.max_batched_tokens,
            max_model_len=args.max_model_len,
            use_graphs=not args.no_graphs,
            quantization=args.quantization,
            draft_model=args.speculative_model,
            num_speculative_tokens=args.num_speculative_tokens if args.speculative_model else 0,
            kernels=args.kernels,
            kv_memory_fraction=0.3,
            async_scheduling=not args.no_async_scheduling,
            **({} if args.ride_along_limit is None else {"ride_along_limit": args.ride_along_limit}),
            **({} if args.launch_margin_ms is None else {"launch_margin_ms": args.launch_margin_ms}),
        )
    )
    engine.warmup()
    _srv, url, stop_server = start_native_in_thread(engine, model_name=args.model)
    child = LoadChild()
    sampler = make_sampler(0.05, gpu_indices=[local_rank]).start()  # native C++ sampling thread when available (no-op readings without a GPU)

    target = args.target_rps if args.target_rps > 0 else 1e6
    cmd = {
        "cmd": "run",
        "url": url,
        "model": args.model,
        "requests": args.requests,
        "concurrency": args.concurrency,
        "max_tokens": args.max_tokens,
        "pattern": args.pattern,
        "target_rps": target,
        "stream": True,
        "seed": rank,
    }

    def barrier() -> None:
        if world > 1:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        child.call(cmd)

    passes: List[Dict[str, Any]] = []
    steps_before = engine.stats.steps
    barrier()
    t0_wall = time.time()
    cpu0 = os.times()
    t0 = time.perf_counter()
    for k in range(args.steps):
        c = dict(cmd)
        if args.out_dir and k == args.steps - 1:
            c["run_dir"] = os.path.join(args.out_dir, f"rank{rank}")
            c["trace"] = True
            c["token_events"] = True
        passes.append(child.call(c))
    barrier()
    elapsed = time.perf_counter() - t0
    t1_wall = time.time()
    cpu1 = os.times()
    engine_steps = engine.stats.steps - steps_before
    samples = sampler.stop()

    # ---- reduce across ranks -------------------------------------------------
    def cat(key: str) -> "np.ndarray":
        return np.array([np.nan if v is None else v for p in passes for v in p[key]], dtype=np.float64)

    local = {k: cat(k) for k in ("latency_ms", "ttfb_ms", "ttft_ms", "status", "completion_tokens", "total_tokens")}
    itl = cat("itl_ms")
    el = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    one = torch.ones(1, dtype=torch.int32, device=cdev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(one)  # every rank contributes a 1: the sum is the number of ranks that took part in a real collective
    elapsed_max = float(el.item())
    collective_ranks = int(one.item())

    n_local = local["latency_ms"].size
    mat = torch.from_numpy(np.stack([local[k] for k in ("latency_ms", "ttfb_ms", "ttft_ms", "status", "completion_tokens", "total_tokens")])).to(cdev)
    if world > 1:
        g = torch.empty((world,) + tuple(mat.shape), dtype=mat.dtype, device=cdev)
        dist.all_gather_into_tensor(g.view(-1), mat.reshape(-1))
        mat_all = g.permute(1, 0, 2).reshape(mat.shape[0], world * n_local)
    else:
        mat_all = mat

    # inter-token gaps: one rank keeps its samples (exact quantiles); several ranks bin theirs
    # into the 2048-bin log-spaced streaming histogram (the kernel of the live view) and merge
    # the bins with ONE 8 KiB all-reduce instead of all-gathering every sample
    from kvmini.ops.hist_quantile import StreamingHistogram

    itl_hist = None
    itl_exact = None
    if world > 1:
        itl_hist = StreamingHistogram(1, device=cdev)
        if itl.size:
            itl_hist.update(itl.astype(np.float32))
        itl_hist.all_reduce(dist)
        # the reported quantiles are EXACT at every rank count (the same definition as one rank's radix
        # select): the ranks sum the radix-digit histograms of each selection pass instead of their samples
        from kvmini.ops.hist_quantile import distributed_quantiles

        itl_exact = distributed_quantiles(itl, (0.5, 0.95, 0.99), dist, device=cdev)

    # lock-step telemetry table: C++ ncclAllGather on its own communicator and stream when the
    # ranks are GPU processes, the torch.distributed path over gloo in the CPU rehearsal
    from kvmini.parallel.telemetry_gather import TelemetryGather, table_to_power_json

    tw = torch.tensor([t0_wall, t1_wall], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.broadcast(tw, 0)
    tg = TelemetryGather(device=cdev, native=on_gpu and world > 1 and cdev.type == "cuda", device_index=local_rank)
    table = tg.gather_table(samples, float(tw[0]), float(tw[1]), 0.1)

    # resource-seconds of this replica through the same record `kvmini cost` reads
    import tempfile

    from kvmini.cluster.backends import LocalBackend
    from kvmini.cost.estimator import sum_resource_seconds
    from kvmini.cost.pricing import load_pricing

    pricing = load_pricing(os.path.join(ROOT, "cost.yaml"))
    with tempfile.TemporaryDirectory() as td:
        lb = LocalBackend(os.path.join(args.out_dir or td, f"local_cluster_rank{rank}.json"))
        lb.register_server("bench", url, os.getpid(), proc_start, gpus=1)
        rsecs = sum_resource_seconds(lb.get_pods("local", "bench"), float(tw[0]), float(tw[0]) + elapsed_max, pricing)
    # CPU and memory are MEASURED for this server process (engine thread + HTTP front end; the load
    # generator is a child and not counted): user + system seconds over the timed window, memory
    # charged at the peak resident set.  The GPU is charged whole for the window.
    server_cpu_s = (cpu1.user - cpu0.user) + (cpu1.system - cpu0.system)
    peak_rss = 0.0
    try:
        with open("/proc/self/status") as f:
            for line in f:
                if line.startswith("VmHWM:"):
                    peak_rss = float(line.split()[1]) * 1024.0
    except OSError:
        pass
    if server_cpu_s > 0 and peak_rss > 0:
        rsecs["cpu_core_seconds"] = server_cpu_s
        rsecs["mem_gib_seconds"] = peak_rss / float(2**30) * elapsed_max
    rs_t = torch.tensor([rsecs["gpu_seconds"], rsecs["cpu_core_seconds"], rsecs["mem_gib_seconds"]], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(rs_t)
    gpu_s, cpu_s, mem_s = (float(v) for v in rs_t.tolist())

    rc = 0
    if rank == 0:
        from kvmini.ops import hist_quantile as hq

        status = mat_all[3]
        ok = status == 200
        lat = torch.where(ok, mat_all[0], torch.full_like(mat_all[0], float("nan")))
        ttfb = torch.where(ok, mat_all[1], torch.full_like(mat_all[1], float("nan")))
        ttft = torch.where(ok, mat_all[2], torch.full_like(mat_all[2], float("nan")))
        if on_gpu:
            q, stats = hq.batched_quantiles(torch.stack([lat, ttfb, ttft]).to(device), (0.5, 0.95, 0.99), return_stats=True)
        else:  # rehearsal: the same reduction on the host
            m3 = torch.stack([lat, ttfb, ttft]).numpy()
            q = np.nanquantile(m3, (0.5, 0.95, 0.99), axis=1).T
            stats = [{"mean": float(np.nanmean(r))} for r in m3]
        itl_hist_q = None
        if itl_hist is not None:
            snap = itl_hist.snapshot((0.5, 0.95, 0.99))[0]["quantiles"]
            itl_hist_q = [snap.get(p, float("nan")) for p in (0.5, 0.95, 0.99)]  # the live view's 2048-bin estimate, merged over the ranks
        if itl_exact is not None:
            qi = [itl_exact]
            itl_engine = ("hip:radix_hist/scan" if cdev.type == "cuda" else "numpy radix select") + " + all-reduce of the digit histograms per pass (exact)"
        elif on_gpu and itl.size:
            qi, _ = hq.batched_quantiles(torch.from_numpy(itl).to(device).reshape(1, -1), (0.5, 0.95, 0.99), return_stats=True)
            itl_engine = "hip:kv_radix_select"
        else:
            qi = [np.nanquantile(itl, (0.5, 0.95, 0.99)).tolist()] if itl.size else [[float("nan")] * 3]
            itl_engine = "numpy"
        n_ok = int(ok.sum().item())
        n_total = int(status.numel())
        completion = float(mat_all[4][ok].sum().item())
        total_tokens = float(mat_all[5][ok].sum().item())
        rps = n_ok / elapsed_max

        # cost: the resource-seconds model of `kvmini cost` (GPU + CPU + memory share of every replica's record, + overhead)
        gpu_hourly = args.gpu_hourly if args.gpu_hourly is not None else pricing.gpu_hourly("AMD_Instinct_MI355X")
        dollars = (gpu_s * gpu_hourly + cpu_s * pricing.cpu_hourly_per_core + mem_s * pricing.mem_hourly_per_gib) / 3600.0 * (1.0 + pricing.overhead_fraction)
        cost_1k = dollars / total_tokens * 1000.0 if total_tokens else None

        pj = table_to_power_json(table, 0.1, source=tg.source)
        watts = [s["watts"] for s in pj["samples"] if s["watts"] is not None]
        wh = None
        if len(watts) >= 2:
            from kvmini.telemetry.energy import PowerSample, trapezoidal_wh

            wh = trapezoidal_wh([PowerSample(s["ts_s"], s["watts"]) for s in pj["samples"]], float(tw[0]), float(tw[1]))
        wh_1k = wh / total_tokens * 1000.0 if wh and total_tokens else None

        p95 = float(q[0][1])
        ttft_mean = stats[1].get("mean")
        out = {
            "metric": "aggregate RPS (with p95 latency ms, TTFT ms, $/1k-tok), Llama-3-8B @ 500req/c20",
            "value": round(rps, 3),
            "unit": "requests/s",
            "n_gpus": world,
            "rccl_ranks": collective_ranks,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed_max / args.steps * 1000.0, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": round(rps / BASELINE["rps"], 4),
            "dtype": ("bf16" if on_gpu else "fp32 (cpu rehearsal)") if not args.quantization else f"bf16 activations, {args.quantization} weights",
            "data": "synthetic prompts ('Hello, world!' chat, 14 prompt tokens), random-init weights, synthetic tokenizer",
            "config": {
                "model": args.model,
                "requests_per_step_per_gpu": args.requests,
                "concurrency_per_gpu": args.concurrency,
                "max_tokens": args.max_tokens,
                "global_batch": args.concurrency * world,
                "seq_len": 14 + args.max_tokens,
                "pattern": args.pattern,
                "target_rps": "saturate" if args.target_rps <= 0 else args.target_rps,
                "stream": True,
                "parallelism": f"dp{world}",
                "kernels": args.kernels,
                "hip_graphs": not args.no_graphs,
                "async_scheduling": not args.no_async_scheduling,
            },
            "p50_ms": float(q[0][0]),
            "p95_ms": p95,
            "p99_ms": float(q[0][2]),
            "ttft_mean_ms": ttft_mean,
            "ttft_p50_ms": float(q[1][0]),
            "ttft_p95_ms": float(q[1][1]),
            "first_content_token_mean_ms": stats[2].get("mean"),
            "itl_p50_ms": float(qi[0][0]),
            "itl_p95_ms": float(qi[0][1]),
            "itl_p99_ms": float(qi[0][2]),
            "tokens_per_sec": completion / elapsed_max,
            "error_rate": (n_total - n_ok) / n_total if n_total else None,
            "cost_per_1k_tokens": cost_1k,
            "gpu_hourly_usd": gpu_hourly,
            "gpu_power_watts_avg": sum(watts) / len(watts) if watts else None,
            "energy_wh_per_1k_tokens": wh_1k,
            "engine_step_ms": round(engine.stats.busy_s / max(1, engine.stats.steps) * 1000.0, 4),
            "engine_steps_timed": engine_steps,
            "engine_first_token_steps": engine.metrics().get("engine_first_token_steps"),
            "engine_first_token_ms": engine.metrics().get("engine_first_token_ms"),  # submit -> first sampled token inside the engine (all passes)
            "model_load_s": round(engine.load_s, 2),
            "graph_warmup_s": round(engine.warm_s, 2),
            "vs_baseline_detail": {
                "rps": round(rps / BASELINE["rps"], 3),
                "p95_ms": round(p95 / BASELINE["p95_ms"], 3),
                # conservative: the first CONTENT token against the reference's first-chunk TTFT
                "ttft_ms": round(stats[2]["mean"] / BASELINE["ttft_ms"], 3) if stats[2].get("mean") else None,
                "ttfb_ms": round(ttft_mean / BASELINE["ttft_ms"], 3) if ttft_mean else None,
                "cost_per_1k_tokens": round(cost_1k / BASELINE["cost_per_1k_tokens"], 4) if cost_1k else None,
                "wh_per_1k_tokens": round(wh_1k / BASELINE["wh_per_1k_tokens"], 5) if wh_1k else None,
            },
            "quantile_engine": "hip:kv_radix_select" if on_gpu else "numpy (cpu rehearsal)",
            "itl_quantile_engine": itl_engine,
            "itl_histogram_p50_p95_p99_ms": itl_hist_q,
            "collective_backend": (dist.get_backend() if world > 1 else None),
            **({"collective_note": collective_note} if collective_note else {}),
            "telemetry": pj["source"] if watts else "unavailable",
            **({"telemetry_native_fallback": tg.native_error} if tg.native_error else {}),
            "resource_seconds": {"gpu": gpu_s, "cpu_core": cpu_s, "mem_gib": mem_s, "cpu_memory_source": "measured: os.times() and VmHWM of each server rank over the timed window"},
        }
        if n_ok == 0:
            out["error"] = passes[-1].get("errors")
            rc = 1
        if args.out_dir:
            from kvmini.utils import write_json

            write_json(os.path.join(args.out_dir, "power.json"), pj)
            write_json(os.path.join(args.out_dir, "bench.json"), out)
        print(json.dumps(out), flush=True)

    child.close()
    stop_server()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())

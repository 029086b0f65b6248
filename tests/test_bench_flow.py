"""The multi-rank flow of ``bench.py`` (one replica + load generator per rank, sample and
telemetry all-gathers, rank-0 reduction, ONE JSON line) rehearsed on CPU over gloo with the
tiny model — the same torchrun command line the GPU run uses, plus ``--device cpu``."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(nproc: int, port: int):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "2", "--warmup", "1", "--device", "cpu", "--model", "tiny", "--requests", "24",
           "--concurrency", "4", "--max-tokens", "6", "--max-model-len", "128"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines  # only rank 0 prints, exactly one line
    return json.loads(lines[0])


def test_bench_two_ranks_gloo_one_json_line():
    out = _run(2, 29741)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in out, key
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1
    assert out["scaling"] == "weak" and out["higher_is_better"] is True
    assert out["config"]["parallelism"] == "dp2" and out["config"]["global_batch"] == 8
    assert out["error_rate"] == 0.0
    # value is the whole-job aggregate: 2 ranks x 2 steps x 24 requests over the timed span
    assert abs(out["value"] - 2 * 2 * 24 / (out["ms_per_step"] * 2 / 1000.0)) / out["value"] < 1e-3
    assert out["tokens_per_sec"] > 0 and out["p95_ms"] >= out["p50_ms"] > 0
    assert out["ttft_mean_ms"] <= out["first_content_token_mean_ms"]


def test_bench_self_launches_ranks_without_torchrun():
    """``python bench.py --gpus 2`` (no torchrun): the parent spawns two fresh rank processes,
    relays rank 0's one JSON line, and the line proves both ranks took part in a collective."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--device", "cpu", "--model", "tiny", "--requests", "16",
           "--concurrency", "4", "--max-tokens", "6", "--max-model-len", "128"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2
    assert out["config"]["parallelism"] == "dp2"
    assert "exact" in out["itl_quantile_engine"] and out["itl_p50_ms"] > 0
    h50, h95, h99 = out["itl_histogram_p50_p95_p99_ms"]  # the 2048-bin estimate merged over the ranks agrees to a bin width
    assert abs(h50 - out["itl_p50_ms"]) <= 0.03 * out["itl_p50_ms"] + 1e-3 and h95 <= h99 * 1.0001
    assert out["resource_seconds"]["gpu"] > 0


def test_bench_rejects_world_size_mismatch():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--device", "cpu", "--model", "tiny"], capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr


def test_bench_four_and_eight_rank_rehearsals_on_cpu():
    """The N = 4 and N = 8 command lines of the scaling run, rehearsed over gloo with the tiny model:
    every rank takes part in the collectives, one JSON line, dp4 / dp8, exact ITL quantiles."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    for n in (4, 8):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0", "--device", "cpu", "--model", "tiny", "--requests", "8",
               "--concurrency", "2", "--max-tokens", "4", "--max-model-len", "128"]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, lines
        out = json.loads(lines[0])
        assert out["n_gpus"] == n and out["rccl_ranks"] == n and out["config"]["parallelism"] == f"dp{n}"
        assert out["error_rate"] == 0.0 and "exact" in out["itl_quantile_engine"]


def test_bench_goes_on_over_gloo_when_rccl_does_not_come_up():
    """Two ranks whose RCCL initialisation fails (forced here: there is no GPU) fall back to gloo for the result
    reductions, say so in the JSON line and finish with every rank counted."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    env["KVMINI_BENCH_COLLECTIVES"] = "force-rccl"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--device", "cpu", "--model", "tiny", "--requests", "6",
           "--concurrency", "2", "--max-tokens", "4", "--max-model-len", "128"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["collective_backend"] == "gloo"
    assert "rccl unavailable" in out["collective_note"] and out["error_rate"] == 0.0


def _dq_worker(rank, world, port, q):
    import numpy as np
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kvmini.ops.hist_quantile import distributed_quantiles

    rng = np.random.default_rng(100 + rank)
    out = {}
    for name, dt in (("f32", np.float32), ("f64", np.float64)):
        x = rng.lognormal(1.0, 0.8, 700 + 313 * rank).astype(dt)
        x[:: 41 + rank] = np.nan
        if rank == 1:
            x[3] = -2.25
        out[name] = (x, distributed_quantiles(x, (0.5, 0.95, 0.99), dist, device="cpu"))
    out["empty_rank"] = distributed_quantiles(np.array([1.0, 5.0]) if rank == 0 else np.zeros(0), (0.5,), dist, device="cpu")
    q.put((rank, out))
    dist.destroy_process_group()


def test_distributed_quantiles_equal_numpy_on_the_concatenation():
    """Exact quantiles of samples spread over ranks (summed radix-digit histograms per pass) ==
    numpy.quantile of all samples together, float32 and float64, NaNs skipped, a rank without samples."""
    import multiprocessing as mp

    import numpy as np

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 3
    procs = [ctx.Process(target=_dq_worker, args=(r, world, 29763, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(30)
    for name in ("f32", "f64"):
        allx = np.concatenate([res[r][name][0] for r in range(world)]).astype(np.float64)
        want = np.nanquantile(allx, (0.5, 0.95, 0.99)).tolist()
        for r in range(world):
            assert res[r][name][1] == want, (name, r, res[r][name][1], want)
    assert all(res[r]["empty_rank"] == [3.0] for r in range(world))

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
    assert out["itl_quantile_engine"].endswith("all-reduce of 2048 bins") and out["itl_p50_ms"] > 0
    assert out["resource_seconds"]["gpu"] > 0


def test_bench_rejects_world_size_mismatch():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--device", "cpu", "--model", "tiny"], capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr

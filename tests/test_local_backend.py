"""The local cluster backend's write side on CPU: an InferenceService manifest becomes server
processes on this host (``apply`` / ``wait_ready`` / ``delete_isvc``), `kvmini deploy` and the
sweeps go through it, and `kvmini serve --tensor-parallel-size N` starts its own ranks.

Reference flows these stand in for on a bare node: ``deploy.sh:108-114`` (apply + wait Ready),
``sweeps/autoscale-sweep.sh:131-163`` (one deployment per cell), ``sweeps/mig-sweep.sh:90-161``
(one deployment per partition profile), ``runners/backends/vllm/deploy.sh:78-79`` (the
tensor-parallel degree is a number the deployment sets).
"""

import csv
import json
import os
import signal
import socket
import subprocess
import sys
import time
import urllib.request

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TINY = ["--device", "cpu", "--max-model-len", "256", "--max-num-seqs", "8", "--max-batched-tokens", "64"]


def _free_port() -> int:
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return int(sk.getsockname()[1])


def _post(url, body, timeout=60):
    req = urllib.request.Request(url + "/v1/chat/completions", data=json.dumps(body).encode(), headers={"Content-Type": "application/json"})
    return json.loads(urllib.request.urlopen(req, timeout=timeout).read())


@pytest.fixture()
def local_env(in_tmp, monkeypatch):
    monkeypatch.setenv("KVMINI_CLUSTER_BACKEND", "local")
    monkeypatch.setenv("PYTHONPATH", ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    monkeypatch.setenv("DISABLE_IO_PROBE", "1")
    from kvmini.cluster import LocalBackend

    lb = LocalBackend()
    yield lb
    for key in list((lb._load().get("isvc") or {})):
        ns, name = key.split("/", 1)
        lb.delete_isvc(ns, name)


def test_local_spec_reads_replicas_gpus_partition_and_args():
    from kvmini.cluster import LocalBackend
    from kvmini.deploy import manifests as mf

    m = mf.native_isvc("svc", "ns1", "tiny", gpus=2, serve_args=["--quantization", "fp8"])
    m = mf.apply_autoscaling(m, initial_scale=3, min_scale=3)
    m = mf.apply_partition_profile(m, {"profile": "cpx", "limits": {"resource_key": "amd.com/gpu-cpx", "count": 1}})
    spec = LocalBackend.local_spec(m)
    assert spec["kind"] == "native" and spec["name"] == "svc" and spec["namespace"] == "ns1"
    assert spec["replicas"] == 3 and spec["gpus"] == 1 and spec["gpu_resource"] == "amd.com/gpu-cpx" and spec["partition"] == "cpx"
    assert spec["args"] == ["--model", "tiny", "--random-init", "--quantization", "fp8"]  # host / port are the backend's to choose
    with pytest.raises(ValueError):
        LocalBackend.local_spec(mf.vllm_rocm_isvc("svc", "ns1"))


def test_apply_wait_delete_and_replace(local_env):
    from kvmini.deploy import manifests as mf

    lb = local_env
    m = mf.native_isvc("eng", "local", "tiny", gpus=1, serve_args=TINY)
    lb.apply(m)
    waited = lb.wait_ready("local", "eng", 120)
    assert waited < 120
    url = lb.get_isvc_url("local", "eng")
    out = _post(url, {"model": "tiny", "messages": [{"role": "user", "content": "hi"}], "max_tokens": 4})
    assert out["usage"]["completion_tokens"] == 4
    pods = lb.get_pods("local", "eng")
    assert len(pods) == 1 and pods[0]["status"]["phase"] == "Running"
    pid1 = lb._load()["isvc"]["local/eng"]["pid"]
    # applying again replaces the running revision
    lb.apply(m)
    lb.wait_ready("local", "eng", 120)
    pid2 = lb._load()["isvc"]["local/eng"]["pid"]
    assert pid2 != pid1 and not lb._alive(pid1)
    lb.delete_isvc("local", "eng")
    assert lb.get_isvc("local", "eng") == {} and not lb._alive(pid2)
    # a server that cannot start is reported, not waited for until the timeout
    lb.apply(mf.native_isvc("bad", "local", "no-such-model", serve_args=TINY))
    with pytest.raises(RuntimeError):
        lb.wait_ready("local", "bad", 60)


def test_deploy_cli_bench_and_delete_on_the_local_backend(local_env, capsys):
    from kvmini.deploy import cli as deploy
    from kvmini.harness import bench

    args = ["--cluster-backend", "local", "--backend", "native", "--model-name", "tiny", "--device", "cpu", "--service", "demo", "--namespace", "local", "--port", "0", "--timeout", "120",
            "--serve-arg=--max-model-len", "--serve-arg=256"]
    assert deploy.main(args) == 0
    assert "Local server ready" in capsys.readouterr().out
    # the benchmark finds the service through the backend (no --url)
    rc = bench.main(["--namespace", "local", "--service", "demo", "--model", "tiny", "--requests", "6", "--concurrency", "2", "--max-tokens", "4", "--run-dir", "runs/local1", "--cluster-backend", "local"])
    assert rc == 0
    with open("runs/local1/results.json") as f:
        res = json.load(f)
    assert res["requests"] == {"total": 6, "success": 6}
    assert deploy.main(["--cluster-backend", "local", "--service", "demo", "--namespace", "local", "--delete"]) == 0
    assert local_env.get_isvc("local", "demo") == {}


def test_autoscale_and_partition_sweeps_deploy_through_the_local_backend(local_env):
    from kvmini.harness import sweeps

    common = ["--cluster-backend", "local", "--namespace", "local", "--native-model", "tiny"] + [f"--serve-arg={x}" for x in TINY]
    assert sweeps.main_autoscale(common + ["--requests", "4", "--container-concurrency", "4", "--initial-scale", "1,2", "--scale-to-zero-grace", "30s", "--stable-window", "60s",
                                           "--panic-window", "6s", "--settle-seconds", "0", "--output", "as.csv"]) == 0
    with open("as.csv", newline="") as f:
        rows = list(csv.reader(f))
    assert len(rows) == 3 and all(r[-1] == "success" for r in rows[1:]), rows
    assert all(float(r[rows[0].index("deploy_time_sec")]) > 0 for r in rows[1:])
    assert sweeps.main_partition(common + ["--profiles", "cpx,spx", "--requests", "4", "--concurrency", "2", "--max-tokens", "4", "--model", "tiny", "--output", "part.csv"]) == 0
    with open("part.csv", newline="") as f:
        prow = list(csv.reader(f))
    assert [r[0] for r in prow[1:]] == ["cpx", "spx"] and all(float(r[2]) > 0 for r in prow[1:]), prow
    assert (local_env._load().get("isvc") or {}) == {}  # every cell cleaned up after itself


def test_two_replicas_sit_behind_the_router(local_env):
    from kvmini.deploy import manifests as mf

    lb = local_env
    m = mf.apply_autoscaling(mf.native_isvc("dp", "local", "tiny", serve_args=TINY), initial_scale=2, min_scale=2)
    lb.apply(m)
    lb.wait_ready("local", "dp", 180)
    dep = lb.get_isvc("local", "dp")
    assert dep["status"]["replicas"] == 2
    url = dep["status"]["url"]
    for _ in range(4):
        assert _post(url, {"model": "tiny", "messages": [{"role": "user", "content": "hi"}], "max_tokens": 3})["usage"]["completion_tokens"] == 3
    pid = lb._load()["isvc"]["local/dp"]["pid"]
    lb.delete_isvc("local", "dp")
    time.sleep(0.3)
    assert not lb._alive(pid)


def test_serve_starts_its_own_tensor_parallel_ranks(tmp_path):
    """`kvmini serve --tensor-parallel-size 2` without torchrun: two rank processes (gloo on CPU), greedy
    output equal to the one-rank server's, sampled requests served, SIGTERM ends every rank."""
    env = dict(os.environ, PYTHONPATH=ROOT, KVMINI_LOCAL_REGISTRY=str(tmp_path / "reg.json"))
    outs = {}
    for tp in (1, 2):
        port = _free_port()
        cmd = [sys.executable, "-m", "kvmini", "serve", "--model", "tiny", "--port", str(port), "--no-register"] + TINY + (["--tensor-parallel-size", "2"] if tp == 2 else [])
        log = open(tmp_path / f"serve_tp{tp}.log", "wb")
        p = subprocess.Popen(cmd, env=env, stdout=log, stderr=subprocess.STDOUT, cwd=str(tmp_path), start_new_session=True)
        try:
            url = f"http://127.0.0.1:{port}"
            t_end = time.time() + 120
            while time.time() < t_end:
                assert p.poll() is None, open(tmp_path / f"serve_tp{tp}.log").read()[-2000:]
                try:
                    if urllib.request.urlopen(url + "/health", timeout=1).status == 200:
                        break
                except Exception:
                    time.sleep(0.2)
            body = {"model": "tiny", "messages": [{"role": "user", "content": "the quick brown fox"}], "max_tokens": 10}
            outs[tp] = _post(url, body)["choices"][0]["message"]["content"]
            sampled = _post(url, dict(body, temperature=0.9, top_p=0.9))
            assert sampled["usage"]["completion_tokens"] == 10
            if tp == 2:
                kids = subprocess.run(["ps", "-o", "pid=", "--ppid", str(p.pid)], capture_output=True, text=True).stdout.split()
                assert len(kids) == 2, kids
        finally:
            p.send_signal(signal.SIGTERM)
            try:
                p.wait(30)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)
                raise
        if tp == 2:
            assert subprocess.run(["ps", "-o", "pid=", "--ppid", str(p.pid)], capture_output=True, text=True).stdout.split() == []
    assert outs[1] == outs[2]


def test_follower_notices_a_dead_scheduling_rank():
    from kvmini.parallel.step_channel import StepChannel

    w = StepChannel(None, 16, ring=2, create=True, n_readers=1)
    r = StepChannel(w.name, 16, ring=2, reader=0)
    try:
        assert r.writer_alive()
        import numpy as np

        assert r.receive(np.zeros(16, dtype=np.int32), timeout_s=0.05) is False  # nothing published: a timeout, not a hang
        w.writer_pid[0] = 2 ** 22 + 12345  # a pid that does not exist
        assert not r.writer_alive()
    finally:
        r.close()
        w.close()


def test_cost_record_prices_measured_cpu_seconds_and_peak_rss(local_env):
    """`kvmini bench` on the local backend snapshots /proc of the service's process tree around the
    load test; the cost record then holds the CPU-seconds really used and memory at the peak
    resident set instead of the manifest's nominal 8 cores / 64 GiB (reference: ``cost_estimator.py:139-175``
    prices requests x lifetime, which is all a cluster exposes)."""
    from kvmini.deploy import manifests as mf
    from kvmini.harness import bench

    lb = local_env
    lb.apply(mf.native_isvc("meas", "local", "tiny", serve_args=TINY))
    lb.wait_ready("local", "meas", 120)
    snap = lb.usage_snapshot("local", "meas")
    assert snap and snap["processes"] >= 1 and snap["cpu_seconds"] > 0 and snap["peak_rss_bytes"] >= snap["rss_bytes"] > 50e6
    assert lb.usage_snapshot("local", "nobody") is None
    rc = bench.main(["--namespace", "local", "--service", "meas", "--model", "tiny", "--requests", "12", "--concurrency", "3", "--max-tokens", "8", "--run-dir", "runs/meas", "--cluster-backend", "local", "--no-energy"])
    assert rc == 0
    use = json.load(open("runs/meas/resource_usage.json"))
    assert use["cpu_core_seconds"] > 0 and use["window_s"] > 0 and use["peak_rss_bytes"] > 50e6
    cb = json.load(open("runs/meas/results.json"))["cost_breakdown"]
    assert cb["cpu_memory_source"].startswith("measured")
    assert 0 < cb["cpu_core_seconds"] <= use["cpu_core_seconds"] + 1e-9
    window = cb["mem_gib_seconds"] / (use["peak_rss_bytes"] / 2**30)  # memory is charged at the peak resident set for the request window
    assert 0 < window <= use["window_s"] + 1e-6
    assert cb["mem_gib_seconds"] < 0.5 * 64 * window  # far below the nominal 64 GiB
    lb.delete_isvc("local", "meas")

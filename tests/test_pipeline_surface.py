"""The benchmark pipeline's outer surface on CPU: `kvmini bench` end to end against the mock
server, the three sweeps and the quantisation sweep on the fake cluster backend (CSV headers,
sentinel rows, --resume), the report generators, the reproducible bundle, manifest snapshots.

Reference contracts: ``analyze.py:573-595`` (results.json keys), ``scripts/loadtest.py:577-590``
(requests.csv header), ``grid-sweep.sh:98``, ``sweeps/autoscale-sweep.sh:113-115``,
``sweeps/mig-sweep.sh:70`` (CSV headers), ``tools/bundle_run.sh:333`` (reproducible tar),
``runners/backends/vllm/deploy.sh`` / ``tgi/deploy.sh`` (InferenceService shape).
"""

import csv
import hashlib
import json
import os

import pytest

from kvmini.utils import load_json

REQUESTS_HEADER = ["id", "scheduled_ms", "start_ms", "ttfb_ms", "tllt_ms", "latency_ms", "status", "prompt_tokens", "completion_tokens", "total_tokens", "error", "trace_id"]
EXTRA_COLUMNS = ["ttft_ms", "chunks"]
CANONICAL_KEYS = {
    "p50_ms", "p95_ms", "p99_ms", "throughput_rps", "tokens_per_sec", "error_rate", "cold_start_count", "time_to_first_token_ms", "ttft_p50_ms", "ttft_p95_ms",
    "gpu_util_avg", "gpu_mem_used_avg", "gpu_power_watts_avg", "cpu_util_avg", "mem_used_avg",
    "cold_count", "cold_p50_ms", "cold_p95_ms", "cold_p99_ms", "warm_count", "warm_success_count", "warm_p50_ms", "warm_p95_ms", "warm_p99_ms", "warm_ttft_p50_ms", "warm_ttft_p95_ms", "warm_error_rate",
    "ttfb_histogram", "tllt_histogram", "generation_time_histogram", "per_token_time_histogram", "ttfb_p50_ms", "ttfb_p95_ms", "ttfb_p99_ms", "tllt_p50_ms", "tllt_p95_ms", "tllt_p99_ms",
    "generation_p50_ms", "generation_p95_ms", "per_token_p50_ms", "per_token_p95_ms", "avg_tokens_per_response",
    "window", "requests", "network_rtt_p50_ms", "network_rtt_p95_ms", "s3_avg_MBps", "cache_hit_ratio",
    "cost_per_request", "cost_per_1k_tokens", "cost_breakdown",
}


@pytest.fixture()
def mock_url():
    from kvmini.server.mock import MockConfig, start_mock_in_thread

    _m, url, stop = start_mock_in_thread(MockConfig(ttft_ms=1.0, tpot_ms=0.05))
    yield url
    stop()


@pytest.fixture()
def dead_url():
    return "http://127.0.0.1:9"


def _bench(argv):
    from kvmini.harness import bench

    return bench.main(argv)


# ---------------------------------------------------------------- (a) + (f): kvmini bench
def test_bench_cpu_smoke_profile_writes_the_canonical_artifacts(in_tmp, mock_url, monkeypatch):
    monkeypatch.setenv("DISABLE_IO_PROBE", "1")
    rc = _bench(["--profile", "cpu-smoke", "--url", mock_url, "--model", "mock-llm", "--concurrency", "5", "--run-dir", "runs/smoke", "--report", "--no-energy"])
    assert rc == 0
    res = load_json("runs/smoke/results.json")
    missing = CANONICAL_KEYS - set(res)
    assert not missing, sorted(missing)
    assert res["requests"] == {"total": 50, "success": 50} and res["error_rate"] == 0.0
    assert set(res["window"]) == {"start", "end", "seconds"} and res["window"]["seconds"] > 0
    assert set(res["cost_breakdown"]) >= {"total_cost", "gpu_seconds", "cpu_core_seconds", "mem_gib_seconds", "gpu_hourly", "cpu_hourly_per_core", "mem_hourly_per_gib", "overhead_fraction", "gpu_product_label"}
    for h in ("ttfb_histogram", "per_token_time_histogram"):
        assert {"bins", "counts", "bin_edges"} <= set(res[h])
    # the other two dialects read the same file (aliases)
    assert res["p95_total_ms"] == res["p95_ms"] and res["p95_latency_ms"] == res["p95_ms"] and res["throughput_req_per_sec"] == res["throughput_rps"]
    with open("runs/smoke/requests.csv", newline="") as f:
        rows = list(csv.reader(f))
    # the reference's twelve columns first, in its order; this build appends two of its own
    assert rows[0][:12] == REQUESTS_HEADER and rows[0][12:] == EXTRA_COLUMNS and len(rows) == 51
    with open("runs/smoke/requests_classified.csv", newline="") as f:
        assert next(csv.reader(f)) == REQUESTS_HEADER + EXTRA_COLUMNS + ["is_cold_start"]
    meta = load_json("runs/smoke/meta.json")
    assert meta["requests"] == 50 and meta["concurrency"] == 5 and meta["max_tokens"] == 32 and meta["pattern"] == "steady"
    assert os.path.exists("runs/smoke/report.html") and os.path.exists("runs/smoke/traces/traces.json")
    assert os.path.islink("runs/latest") or os.path.exists("runs/latest")


def test_bench_returns_4_when_no_request_succeeded(in_tmp, dead_url, monkeypatch, capsys):
    from kvmini.harness.bench import EXIT_NO_SUCCESS

    monkeypatch.setenv("DISABLE_IO_PROBE", "1")
    rc = _bench(["--url", dead_url, "--requests", "6", "--concurrency", "3", "--max-tokens", "4", "--run-dir", "runs/dead", "--no-energy"])
    assert rc == EXIT_NO_SUCCESS == 4
    out = capsys.readouterr()
    assert "=== FAILED" in out.out and "=== DONE ===" not in out.out and "0 of 6 requests succeeded" in out.err
    res = load_json("runs/dead/results.json")
    assert res["error_rate"] == 1.0 and res["p95_ms"] is None  # artifacts still describe the failure


def test_bench_dry_run_and_bad_profile(in_tmp, capsys):
    assert _bench(["--profile", "standard", "--dry-run"]) == 0
    assert "--dry-run enabled" in capsys.readouterr().out
    assert _bench(["--profile", "no-such-profile", "--dry-run"]) == 1
    assert _bench(["--list-profiles"]) == 0
    assert "cpu-smoke" in capsys.readouterr().out


# ---------------------------------------------------------------- (b): sweeps
def test_grid_sweep_header_rows_resume_and_error_sentinels(in_tmp, mock_url, dead_url, monkeypatch, capsys):
    from kvmini.harness import sweeps

    monkeypatch.setenv("DISABLE_IO_PROBE", "1")
    common = ["--requests", "8", "--concurrency", "2,4", "--max-tokens", "4", "--patterns", "steady", "--model", "mock-llm"]
    assert sweeps.main_grid(common + ["--url", mock_url, "--output", "grid.csv", "--resume", "--report"]) == 0
    with open("grid.csv", newline="") as f:
        rows = list(csv.reader(f))
    # reference grid-sweep.sh:98
    assert rows[0] == "concurrency,max_tokens,pattern,p50_ms,p95_ms,p99_ms,throughput_rps,tokens_per_sec,error_rate,ttft_p50_ms,ttft_p95_ms,cost_per_request,cost_per_1k_tokens,gpu_util_avg,run_dir".split(",")
    assert [r[0] for r in rows[1:]] == ["2", "4"] and all(float(r[4]) > 0 for r in rows[1:])
    assert os.path.exists("grid_report.html")
    first = [r[:] for r in rows]
    mtimes = {r[-1]: os.path.getmtime(os.path.join(r[-1], "results.json")) for r in rows[1:]}
    # --resume against a DEAD endpoint: finished cells are not re-run, their rows are rebuilt
    capsys.readouterr()
    assert sweeps.main_grid(common + ["--url", dead_url, "--output", "grid.csv", "--resume"]) == 0
    assert "2 cells resumed" in capsys.readouterr().out
    with open("grid.csv", newline="") as f:
        assert list(csv.reader(f)) == first
    assert {d: os.path.getmtime(os.path.join(d, "results.json")) for d in mtimes} == mtimes
    # a fresh sweep against the dead endpoint: ERROR rows, and nothing that --resume would take for finished
    assert sweeps.main_grid(common + ["--url", dead_url, "--output", "dead.csv", "--runs-root", "runs_dead", "--resume"]) == 0
    with open("dead.csv", newline="") as f:
        rows = list(csv.reader(f))
    assert all(r[3:14] == ["ERROR"] * 11 for r in rows[1:]) and len(rows) == 3
    assert not any(os.path.exists(os.path.join(r[-1], "results.json")) for r in rows[1:])
    assert all(os.path.exists(os.path.join(r[-1], "results.json.failed")) for r in rows[1:])


def test_autoscale_sweep_on_the_fake_backend(in_tmp, mock_url, dead_url, monkeypatch):
    from kvmini.harness import sweeps

    monkeypatch.setenv("DISABLE_IO_PROBE", "1")
    args = ["--requests", "6", "--container-concurrency", "1,4", "--initial-scale", "0", "--scale-to-zero-grace", "30s", "--stable-window", "60s", "--panic-window", "6s",
            "--max-cells", "2", "--settle-seconds", "0", "--cluster-backend", "fake", "--backend", "native", "--resume"]
    assert sweeps.main_autoscale(args + ["--url", mock_url, "--output", "as.csv"]) == 0
    with open("as.csv", newline="") as f:
        rows = list(csv.reader(f))
    # reference sweeps/autoscale-sweep.sh:113-115
    assert rows[0] == ("container_concurrency,initial_scale,scale_to_zero_grace_period,stable_window,panic_window,p50_ms,p95_ms,p99_ms,p50_warm_ms,p95_warm_ms,p50_cold_ms,p95_cold_ms,"
                       "cold_start_count,cold_multiplier,throughput_rps,tokens_per_sec,error_rate,cost_per_request,cost_per_1k_tokens,cold_cost_per_request,warm_cost_per_request,"
                       "cold_cost_per_1k_tokens,warm_cost_per_1k_tokens,energy_wh_per_1k_tokens,gpu_util_avg,run_dir,deploy_time_sec,notes").split(",")
    assert len(rows) == 3 and all(r[-1] == "success" and len(r) == len(rows[0]) for r in rows[1:])
    summary = load_json("as_summary.json")
    assert summary["successful_runs"] == 2 and summary["scale_to_zero"]["runs"] == 2
    # --resume against a dead endpoint: same rows (deploy time read back from results.json)
    assert sweeps.main_autoscale(args + ["--url", dead_url, "--output", "as2.csv"]) == 0
    with open("as2.csv", newline="") as f:
        rows2 = list(csv.reader(f))
    assert [r[:5] + r[-3:] for r in rows2] == [r[:5] + r[-3:] for r in rows] and rows2[1][6] == rows[1][6]
    # a fresh failing sweep: benchmark_failed sentinel rows of full width
    assert sweeps.main_autoscale(args + ["--url", dead_url, "--output", "as3.csv", "--runs-root", "runs_dead"]) == 0
    with open("as3.csv", newline="") as f:
        rows3 = list(csv.reader(f))
    assert all(r[-1] == "benchmark_failed" and len(r) == len(rows3[0]) for r in rows3[1:])


def test_partition_sweep_on_the_fake_backend(in_tmp, mock_url, dead_url, monkeypatch):
    from kvmini.harness import sweeps

    monkeypatch.setenv("DISABLE_IO_PROBE", "1")
    args = ["--profiles", "cpx,spx", "--requests", "6", "--concurrency", "2", "--max-tokens", "4", "--cluster-backend", "fake", "--backend", "native", "--resume"]
    assert sweeps.main_partition(args + ["--url", mock_url, "--output", "part.csv", "--report"]) == 0
    with open("part.csv", newline="") as f:
        rows = list(csv.reader(f))
    assert rows[0] == "profile,p50_ms,p95_ms,throughput_rps,Wh_per_1k_tokens,cost_per_1k_tokens,error_rate".split(",")  # sweeps/mig-sweep.sh:70
    assert [r[0] for r in rows[1:]] == ["cpx", "spx"] and all(float(r[2]) > 0 for r in rows[1:])
    assert os.path.exists("part_report.html")
    assert sweeps.main_partition(args + ["--url", dead_url, "--output", "part2.csv"]) == 0  # resumed: nothing is re-run
    with open("part2.csv", newline="") as f:
        assert list(csv.reader(f)) == rows
    assert sweeps.main_partition(args + ["--url", dead_url, "--output", "part3.csv", "--runs-root", "runs_dead"]) == 0
    with open("part3.csv", newline="") as f:
        rows3 = list(csv.reader(f))
    assert all(r[1:] == ["ERROR"] * 6 for r in rows3[1:])


def test_quant_sweep_resume_skips_finished_configs(in_tmp, mock_url, dead_url, monkeypatch):
    import yaml

    from kvmini.harness import quantization_sweep as qs

    monkeypatch.setenv("DISABLE_IO_PROBE", "1")
    base = {"max_model_len": 512, "gpu_memory_utilization": 0.5, "tensor_parallel_size": 1}
    cfg = {"benchmark": {"namespace": "bench", "service": "sweep", "requests": 6, "concurrency": 2, "model": "mock-llm"},
           "sweep_configurations": [{"name": "baseline", "quantization": "none", "kv_cache_dtype": "auto", **base}, {"name": "fp8", "quantization": "fp8", "kv_cache_dtype": "fp8", **base}]}
    with open("q.yaml", "w") as f:
        yaml.safe_dump(cfg, f)
    common = ["--config", "q.yaml", "--output-dir", "qs", "--cluster-backend", "fake", "--no-quality"]
    assert qs.main(common + ["--url", mock_url]) == 0
    first = list(csv.DictReader(open("qs/sweep_results_intermediate.csv")))
    assert [r["config_name"] for r in first] == ["baseline", "fp8"] and all(not r["error"] for r in first)
    # resumed against a dead endpoint: both configurations are taken from the first pass
    assert qs.main(common + ["--url", dead_url, "--resume"]) == 0
    again = list(csv.DictReader(open("qs/sweep_results_intermediate.csv")))
    assert [(r["config_name"], r["p95_ms"]) for r in again] == [(r["config_name"], r["p95_ms"]) for r in first]


def test_quant_sweep_agreement_quality_axis(in_tmp, mock_url, monkeypatch):
    """``quality_eval.mode: agreement``: the reference configuration scores 100 and leaves its
    greedy continuations behind; the others are scored against them (the mock answers every
    configuration alike, so they agree fully) and the score reaches the CSV and the report."""
    import yaml

    from kvmini.harness import quantization_sweep as qs

    monkeypatch.setenv("DISABLE_IO_PROBE", "1")
    base = {"max_model_len": 512, "gpu_memory_utilization": 0.5, "tensor_parallel_size": 1}
    cfg = {"benchmark": {"namespace": "bench", "service": "sweep", "requests": 6, "concurrency": 2, "model": "mock-llm"},
           "quality_eval": {"enabled": True, "mode": "agreement", "prompts": 5, "max_tokens": 8},
           "sweep_configurations": [{"name": "fp8", "quantization": "fp8", "kv_cache_dtype": "fp8", **base}, {"name": "bf16", "reference": True, "quantization": "none", "kv_cache_dtype": "auto", **base},
                                    {"name": "fp8-again", "quantization": "fp8", "kv_cache_dtype": "auto", **base}]}
    with open("q.yaml", "w") as f:
        yaml.safe_dump(cfg, f)
    assert qs.main(["--config", "q.yaml", "--output-dir", "qs", "--cluster-backend", "fake", "--url", mock_url]) == 0
    rows = {r["config_name"]: r for r in csv.DictReader(open("qs/sweep_results_intermediate.csv"))}
    assert rows["fp8"]["quality_score"] == ""  # ran before the reference: no score, not an error
    assert float(rows["bf16"]["quality_score"]) == 100.0 and float(rows["fp8-again"]["quality_score"]) == 100.0
    ref = load_json("qs/agreement_reference.json")
    assert ref["config_name"] == "bf16" and len(ref["continuations"]) == 5 and all(ref["continuations"])
    assert "quality" in open("qs/sweep_report.html").read()


# ---------------------------------------------------------------- (c): reports
def test_single_report_links_the_p95_trace_and_writes_the_viewer(in_tmp, mock_url, monkeypatch):
    from kvmini.report.generator import generate_single_report

    monkeypatch.setenv("DISABLE_IO_PROBE", "1")
    assert _bench(["--url", mock_url, "--requests", "20", "--concurrency", "4", "--max-tokens", "8", "--model", "mock-llm", "--run-dir", "runs/r", "--no-energy"]) == 0
    out = generate_single_report("runs/r/results.json", "runs/r/report.html", cost_file="cost.yaml")
    html = open(out).read()
    rows = list(csv.DictReader(open("runs/r/requests.csv")))
    p95 = load_json("runs/r/results.json")["p95_ms"]
    want = min(rows, key=lambda r: abs(float(r["latency_ms"]) - p95))
    assert f"traces/view.html?trace={want['trace_id']}" in html  # deep link to the request closest to p95
    assert os.path.exists("runs/r/traces/view.html") and "traces.json" in open("runs/r/traces/view.html").read()
    for needle in ("P95 latency", "Recommendations", "<svg"):
        assert needle in html
    assert "Schema-valid answers" not in html  # no request of this run carried a schema
    from kvmini.utils import write_json

    write_json("runs/r/results.json", {**load_json("runs/r/results.json"), "structured_output": {"checked": 20, "valid_json": 20, "schema_valid": 19, "schema_valid_rate": 0.95, "first_failures": []}})
    html = open(generate_single_report("runs/r/results.json", "runs/r/report2.html", cost_file="cost.yaml")).read()
    assert "Schema-valid answers" in html and "19 / 20" in html


def test_grid_and_partition_reports_from_fixtures(tmp_path):
    from kvmini.harness.sweeps import GRID_HEADER, PARTITION_HEADER
    from kvmini.report.generator import generate_grid_report, generate_partition_report

    g = tmp_path / "grid.csv"
    with open(g, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(GRID_HEADER)
        for c in (1, 8):
            for t in (32, 128):
                w.writerow([c, t, "steady", 10.0 * c, 20.0 * c + t, 30.0 * c, 5.0 * c, 100.0 * c, 0.0, 3.0, 5.0, 0.001, 0.002 / c, 50, f"runs/x_{c}_{t}"])
        w.writerow([16, 32, "steady"] + ["ERROR"] * 11 + ["runs/bad"])
    html = open(generate_grid_report(str(g), str(tmp_path / "grid.html"))).read()
    assert "<svg" in html and "ERROR" not in html.split("<table")[0]  # failed cells do not reach the charts
    assert "runs/x_8_128" in html and "runs/bad" not in html
    p = tmp_path / "part.csv"
    with open(p, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(PARTITION_HEADER)
        w.writerow(["cpx", 100, 150, 4.0, 0.5, 0.01, 0.0])
        w.writerow(["spx", 40, 60, 20.0, 0.2, 0.004, 0.0])
        w.writerow(["qpx"] + ["ERROR"] * 6)
    html = open(generate_partition_report(str(p), str(tmp_path / "part.html"))).read()
    assert "cpx" in html and "spx" in html and "<svg" in html


# ---------------------------------------------------------------- (d): bundle
def test_bundle_is_byte_reproducible_and_complete(in_tmp, mock_url, monkeypatch):
    import tarfile
    import time

    from kvmini.cluster import FakeBackend
    from kvmini.harness.bundle import bundle_run

    monkeypatch.setenv("DISABLE_IO_PROBE", "1")
    assert _bench(["--url", mock_url, "--requests", "10", "--concurrency", "2", "--max-tokens", "4", "--model", "mock-llm", "--run-dir", "runs/b", "--no-energy"]) == 0
    be = FakeBackend()
    be.add_node()
    be.add_pod(name="demo-llm-predictor-0", namespace="ml-prod", service="demo-llm", image="rocm/vllm@sha256:" + "a" * 64)
    a = bundle_run("runs/b", "ml-prod", "demo-llm", output="artifacts/one.tar.gz", backend=be, quiet=True)
    time.sleep(1.1)  # a wall-clock field or an mtime in the archive would differ now
    b = bundle_run("runs/b", "ml-prod", "demo-llm", output="artifacts/two/one.tar.gz", backend=be, quiet=True)
    sha = [hashlib.sha256(open(p, "rb").read()).hexdigest() for p in (a, b)]
    assert sha[0] == sha[1]
    with tarfile.open(a) as t:
        names = set(t.getnames())
        assert all(m.mtime == t.getmembers()[0].mtime and m.uid == 0 and m.uname == "" for m in t.getmembers())
    for f in ("results.json", "requests.csv", "meta.json", "provenance.json", "cluster_facts.json", "SUMMARY.md", "supply-chain/images.txt", "configs/cost.yaml", "traces/traces.json", "grafana/README.md"):
        assert f"one/{f}" in names, f
    with tarfile.open(a) as t:
        assert "rocm/vllm@sha256:" in t.extractfile("one/supply-chain/images.txt").read().decode()
    assert os.path.exists(a + ".sha256") and open(a + ".sha256").read().split()[0] == sha[0]


# ---------------------------------------------------------------- (e): manifests
def test_manifest_snapshots_for_the_three_runtimes():
    from kvmini.deploy import manifests as mf

    v = mf.vllm_rocm_isvc("demo-llm", "ml-prod", "s3://models/llama-3-8b")
    t = mf.tgi_rocm_isvc("demo-llm", "ml-prod")
    n = mf.native_isvc("demo-llm", "ml-prod")
    for m, backend in ((v, "vllm-rocm"), (t, "tgi-rocm"), (n, "native")):
        assert m["apiVersion"] == "serving.kserve.io/v1beta1" and m["kind"] == "InferenceService"
        assert m["metadata"]["name"] == "demo-llm" and m["metadata"]["namespace"] == "ml-prod"
        assert backend in json.dumps(m["metadata"])
        w = mf._workload(m)
        assert w["resources"]["limits"]["amd.com/gpu"] == w["resources"]["requests"]["amd.com/gpu"] == "1" or w["resources"]["limits"]["amd.com/gpu"] == 1
        assert "nvidia.com/gpu" not in json.dumps(m)
        assert mf.to_yaml(m).startswith("apiVersion: serving.kserve.io/v1beta1")
    assert "s3://models/llama-3-8b" in json.dumps(v)
    assert any("kvmini" in str(x) for x in (mf._workload(n).get("command") or []) + (mf._workload(n).get("args") or []))

    # partition profile (both schemas): resource name, selectors / tolerations and the label move together
    prof = {"profile": "cpx", "limits": {"resource_key": "amd.com/gpu-cpx", "count": 1}, "nodeSelector": {"amd.com/gpu.compute-partition": "CPX"},
            "tolerations": [{"key": "amd.com/partitioned", "operator": "Exists", "effect": "NoSchedule"}]}
    before = json.dumps(v, sort_keys=True)
    pv = mf.apply_partition_profile(v, prof)
    assert json.dumps(v, sort_keys=True) == before  # the input manifest is not edited in place
    w = mf._workload(pv)
    assert w["resources"]["limits"]["amd.com/gpu-cpx"] == w["resources"]["requests"]["amd.com/gpu-cpx"] == "1" and "amd.com/gpu" not in w["resources"]["limits"]
    assert pv["spec"]["predictor"]["nodeSelector"] == {"amd.com/gpu.compute-partition": "CPX"} and pv["spec"]["predictor"]["tolerations"][0]["key"] == "amd.com/partitioned"
    assert pv["metadata"]["labels"]["kvmini/partition"] == "cpx"
    pv2 = mf.apply_partition_profile(t, {"profile": "vf", "resources": {"limits": {"amd.com/gpu-vf": 2}, "requests": {"amd.com/gpu-vf": 2}}, "affinity": {"nodeAffinity": {}}})
    assert mf._workload(pv2)["resources"]["limits"]["amd.com/gpu-vf"] == "2" and "affinity" in pv2["spec"]["predictor"]
    # every partition profile shipped in the tree applies cleanly
    from kvmini.config.profiles import resolve_profile_path
    from kvmini.utils import load_yaml

    for name in ("cpx", "qpx", "dpx", "spx"):
        pm = mf.apply_partition_profile(n, load_yaml(resolve_profile_path(name)))
        assert any(k.startswith("amd.com/gpu") for k in mf._workload(pm)["resources"]["limits"])

    # autoscaling: Knative annotations (sweeps/autoscale-sweep.sh:147-170)
    av = mf.apply_autoscaling(n, target=4, initial_scale=0, min_scale=0, max_scale=4, scale_to_zero_grace="30s", stable_window="60s", panic_window_pct="10.0")
    ann = av["metadata"]["annotations"]
    assert {k: ann[k] for k in ann if k.startswith("autoscaling.knative.dev/") and k.rsplit("/", 1)[1] not in ("class", "metric")} == {
        "autoscaling.knative.dev/target": "4", "autoscaling.knative.dev/initialScale": "0", "autoscaling.knative.dev/minScale": "0", "autoscaling.knative.dev/maxScale": "4",
        "autoscaling.knative.dev/scaleToZeroGracePeriod": "30s", "autoscaling.knative.dev/window": "60s", "autoscaling.knative.dev/panicWindowPercentage": "10.0"}
    assert ann.get("autoscaling.knative.dev/metric", "concurrency") == "concurrency"
    assert n["metadata"]["annotations"]["autoscaling.knative.dev/target"] == "10"  # the builder's default, untouched by the copy above


def test_deploy_cli_writes_manifest_and_applies_to_fake_backend(in_tmp, capsys):
    from kvmini.deploy import cli

    assert cli.main(["--backend", "vllm-rocm", "--service", "svc-a", "--namespace", "ns", "--model-uri", "pvc://m/llama", "--dry-run"]) == 0
    out = capsys.readouterr().out
    assert "kind: InferenceService" in out and "svc-a" in out and "pvc://m/llama" in out


def test_preflight_dashboards_and_repo_checks(in_tmp, capsys):
    from kvmini.cluster import FakeBackend
    from kvmini.deploy.preflight import run_preflight
    from kvmini.gates import repo_checks
    from kvmini.report import dashboards

    be = FakeBackend()
    be.add_node()
    assert run_preflight(be, quiet=True) == 0
    # the kubectl path against a stub kubectl: context, KServe CRD and GPU capacity are all asked for
    from kvmini.cluster import KubectlBackend

    stub = in_tmp / "kubectl"
    stub.write_text("#!/bin/sh\necho \"$@\" >> kubectl_calls.txt\ncase \"$*\" in *current-context*) echo test-ctx;; *jsonpath*) echo 8;; *) echo ok;; esac\n")
    stub.chmod(0o755)
    assert run_preflight(KubectlBackend(kubectl=str(stub)), quiet=False) == 0
    calls = open("kubectl_calls.txt").read()
    assert "current-context" in calls and "inferenceservices.serving.kserve.io" in calls and "amd\\.com/gpu" in calls
    assert "kubectl context: test-ctx" in capsys.readouterr().out
    stub.write_text("#!/bin/sh\ncase \"$*\" in *crd*) exit 1;; *) echo ok;; esac\n")
    assert run_preflight(KubectlBackend(kubectl=str(stub)), quiet=True) == 2  # no KServe CRD: hard failure
    files = dashboards.write_all("dash_out")
    assert len(files) >= 4
    for p in files:
        d = load_json(p)
        assert d["panels"] and d["uid"] and all("targets" in pn for pn in d["panels"])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rc, sections = repo_checks.run_all(root)
    assert rc == 0, sections
    assert repo_checks.validate_schema({"a": 1}, {"type": "object", "required": ["b"]}) and not repo_checks.validate_schema({"b": "x"}, {"type": "object", "required": ["b"], "properties": {"b": {"type": "string"}}})
    assert repo_checks.deep_merge({"a": {"x": 1, "y": 2}, "k": 1}, {"a": {"y": 3}}) == {"a": {"x": 1, "y": 3}, "k": 1}


def test_docs_site_builds_every_page_with_working_sidebar_links(tmp_path):
    import re

    from kvmini.report import docsite

    files = docsite.build_site(str(tmp_path / "site"))
    names = {os.path.basename(f) for f in files}
    assert {"index.html", "readme.html", "design.html", "roadmap.html", "docs-api.html", "docs-kernels.html", "docs-troubleshooting.html", "docs-features.html"} <= names
    home = open(tmp_path / "site" / "index.html").read()
    for href in set(re.findall(r"href='([a-z0-9-]+\.html)'", home)):
        assert href in names, href  # every sidebar entry points at a page that was written
    assert home.count("class='card'") == len(docsite.FEATURES) and home.count("<svg class='diagram'") == 3
    kernels = open(tmp_path / "site" / "docs-kernels.html").read()
    assert "<table" in kernels and "<pre>" not in kernels.split("<main>")[0] and "const IDX=" in kernels
    # the built-in converter (no markdown-it on the box) handles what the documents use
    body = docsite._fallback_markdown("# T\n\ntext with `code` and **bold** and [a link](docs/API.md)\n\n| a | b |\n|---|---|\n| 1 | 2 |\n\n```bash\nmake docs\n```\n\n* one\n  continued\n* two\n")
    assert "<h1 id='t'>T</h1>" in body and "<code>code</code>" in body and "<strong>bold</strong>" in body and '<a href="docs/API.md">' in body
    assert "<table><tr><th>a</th><th>b</th></tr><tr><td>1</td><td>2</td></tr></table>" in body and "<pre><code>make docs</code></pre>" in body
    assert "<li>one continued</li><li>two</li>" in body
    assert docsite.main(["--out", str(tmp_path / "site2")]) == 0

"""CPU tests for the harness tools, driven against the in-tree mock server and
the fake / local cluster backends (no cluster, no GPU)."""

import argparse
import csv
import json
import os

import numpy as np
import pytest

from kvmini.utils import load_json

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------- fairness
def test_fairness_guard_logic():
    from kvmini.harness.fairness import Guard

    g = Guard(p95_budget_ms=100.0, window=5, cooldown_sec=0.2)
    for _ in range(5):
        g.observe(10.0)
    assert not g.should_throttle_b() and g.trigger_count == 0
    g.observe(500.0)  # floor-index p95 of [10,10,10,10,500] = element 3 -> still 10
    assert g.trigger_count == 0
    g.observe(500.0)  # [10,10,10,500,500] -> idx int(.95*4)=3 -> 500
    assert g.trigger_count == 1 and g.activations == 1 and g.should_throttle_b()
    g.observe(600.0)
    assert g.trigger_count == 2 and g.activations == 1  # still inside the cooldown: same activation
    assert Guard(None).should_throttle_b() is False


def test_fairness_end_to_end(mock_server, in_tmp):
    from kvmini.harness import fairness

    _mock, url = mock_server
    run_dir = str(in_tmp / "fair")
    slo = in_tmp / "fair_slo.json"
    slo.write_text(json.dumps({"fairness": {"a_p95_ms_max": 5000, "b_p95_ms_max": 5000}}))
    rc = fairness.main(["--url", url, "--model", "mock-llm", "--tenant-a-requests", "30", "--tenant-b-requests", "30", "--tenant-a-concurrency", "4", "--tenant-b-concurrency", "4",
                        "--max-tokens", "4", "--p95-budget-ms", "0.001", "--guard-cooldown", "0.05", "--run-dir", run_dir, "--slo", str(slo)])
    assert rc == 0
    doc = load_json(os.path.join(run_dir, "fairness_summary.json"))
    assert doc["summary"]["A"]["success"] == 30  # tenant A is never throttled
    assert doc["guard_triggers"] > 0 and doc["guard_activations"] >= 1
    assert doc["summary"]["B"]["throttled"] > 0  # an impossible budget throttles B
    assert doc["slo_gate"] == "pass" and "budgets" in doc
    rows = list(csv.DictReader(open(os.path.join(run_dir, "requests.csv"))))
    assert set(rows[0]) == {"id", "tenant", "start_ms", "latency_ms", "status", "error", "guard_action"}
    assert any(r["guard_action"] == "throttled" and r["status"] == "429" for r in rows)
    assert os.path.exists(os.path.join(run_dir, "fairness_report.html"))
    sh = doc["summary"]["throughput_share"]
    assert abs(sh["A"] + sh["B"] - 1.0) < 1e-9


# ---------------------------------------------------------------- parity probe
def test_parity_probe_full_and_reduced(in_tmp):
    from kvmini.harness import parity_probe
    from kvmini.server.mock import MockConfig, start_mock_in_thread

    _m, url, stop = start_mock_in_thread(MockConfig(ttft_ms=1.0, tpot_ms=0.1))
    try:
        rc = parity_probe.main(["--url", url, "--model", "mock-llm", "--out", "cap.json", "--html", "cap.html"])
        assert rc == 0
        res = load_json("cap.json")
        assert res["summary"] == {"supported_count": 5, "total": 5}, res
        assert set(res["capabilities"]) == {"tools_function_calling", "parallel_tool_calls", "json_mode", "logprobs", "streaming"}
        assert "Capability" in open("cap.html").read() or "capability" in open("cap.html").read()
    finally:
        stop()
    _m, url, stop = start_mock_in_thread(MockConfig(ttft_ms=1.0, tpot_ms=0.1, supports_tools=False, supports_logprobs=False, supports_json_mode=False))
    try:
        parity_probe.main(["--url", url, "--model", "mock-llm", "--out", "cap2.json"])
        res = load_json("cap2.json")
        caps = res["capabilities"]
        assert not caps["tools_function_calling"]["supported"] and not caps["parallel_tool_calls"]["supported"]
        assert not caps["logprobs"]["supported"] and not caps["json_mode"]["supported"]
        assert caps["streaming"]["supported"] and res["summary"]["supported_count"] == 1
    finally:
        stop()
    # dead endpoint: every probe reports unsupported instead of raising
    parity_probe.main(["--url", "http://127.0.0.1:9", "--model", "x", "--out", "cap3.json"])
    assert load_json("cap3.json")["summary"]["supported_count"] == 0


# ---------------------------------------------------------------- cache probe
def test_cache_probe_prompt_sets_are_deterministic():
    from kvmini.harness.cache_probe import analyse, generate_prompt_sets

    r1, u1 = generate_prompt_sets(100)
    r2, u2 = generate_prompt_sets(100)
    assert r1 == r2 and u1 == u2
    assert len(r1) == 100 and len(set(r1)) <= 20 and len(set(u1)) == 100
    an = analyse({"mean_ttft": 20.0, "std_ttft": 2.0, "count": 100}, {"mean_ttft": 50.0, "std_ttft": 2.0, "count": 100})
    assert an["inferred_hit_ratio"] == pytest.approx(min(1.0, 0.6 / 0.8))
    assert an["statistical_significance"] == "significant" and an["interpretation"]["cache_effectiveness"] == "high"
    an0 = analyse({"mean_ttft": 50.0, "std_ttft": 0.0, "count": 10}, {"mean_ttft": 50.0, "std_ttft": 0.0, "count": 10})
    assert an0["inferred_hit_ratio"] == 0.0 and an0["t_statistic"] == 0


def test_cache_probe_end_to_end(in_tmp):
    from kvmini.harness import cache_probe
    from kvmini.server.mock import MockConfig, start_mock_in_thread

    _m, url, stop = start_mock_in_thread(MockConfig(ttft_ms=30.0, tpot_ms=0.05, jitter=0.02, prefix_cache=True, prefix_cache_speedup=5.0))
    try:
        rc = cache_probe.main(["--url", url, "--base-requests", "40", "--max-tokens", "2", "--run-dir", "cache"])
    finally:
        stop()
    assert rc == 0
    doc = load_json("cache/cache_analysis.json")["cache_analysis"]
    assert doc["method"] == "ttft_heuristic"
    rep, uni = doc["experiments"]["repeat_80"], doc["experiments"]["unique_100"]
    assert rep["mean_ttft"] < uni["mean_ttft"]
    assert rep["measured_hit_ratio"] > 0.5 and uni["measured_hit_ratio"] < 0.2
    assert doc["analysis"]["inferred_hit_ratio"] > 0.3
    assert os.path.exists("cache/repeat_80/requests.csv") and os.path.exists("cache/unique_100_prompts.json")


# ---------------------------------------------------------------- quality + pareto
def test_quality_scoring_with_oracle_and_pareto(tmp_path):
    from kvmini.harness.quality import COMPLETION_ITEMS, MATH_ITEMS, YESNO_ITEMS, ParetoAnalyzer, QualityEvaluator

    def oracle(prompt, max_tokens):
        for it in COMPLETION_ITEMS:
            if it["context"] in prompt:
                return "ABCD"[it["correct"]], 0.01, 1
        for it in YESNO_ITEMS:
            if it["question"] in prompt:
                return ("Yes" if it["answer"] else "No"), 0.01, 1
        for it in MATH_ITEMS:
            if it["problem"] in prompt:
                return f"The answer is {it['answer']}", 0.01, 3
        return "", 0.01, 0

    q = QualityEvaluator("http://x", "m", call_fn=oracle).evaluate()
    assert q["quality_score"] == 100.0 and q["avg_quality_latency_ms"] == 10.0
    assert {k: v["samples"] for k, v in q["task_results"].items()} == {"hellaswag": 3, "boolq": 3, "math": 4}
    bad = QualityEvaluator("http://x", "m", call_fn=lambda p, n: ("Z", 0.0, 1)).evaluate()
    assert bad["quality_score"] == 0.0
    only = QualityEvaluator("http://x", "m", tasks=["boolq"], call_fn=lambda p, n: ("no", 0.0, 1)).evaluate()
    assert only["quality_score"] == pytest.approx(200 / 3, abs=0.01)

    pts = [
        {"name": "a", "quality_score": 90, "p95_ms": 100, "cost_per_1k_tokens": 1.0},
        {"name": "b", "quality_score": 80, "p95_ms": 120, "cost_per_1k_tokens": 1.2},  # dominated by a
        {"name": "c", "quality_score": 70, "p95_ms": 50, "cost_per_1k_tokens": 2.0},   # fastest
        {"name": "d", "quality_score": 60, "p95_ms": 200, "cost_per_1k_tokens": 0.5},  # cheapest
    ]
    out = ParetoAnalyzer.classify_pareto_bucket(pts)
    assert [o["pareto_bucket"] for o in out] == ["on-pareto", "off-pareto", "on-pareto", "on-pareto"]
    assert ParetoAnalyzer.frontier_mask(np.zeros((0, 3))).shape == (0,)


def test_quality_against_mock_and_results_merge(mock_server, tmp_path):
    from kvmini.harness import quality

    _m, url = mock_server
    rf = tmp_path / "results.json"
    rf.write_text(json.dumps({"p95_ms": 10.0}))
    assert quality.main(["--endpoint", url, "--model", "mock-llm", "--results-file", str(rf)]) == 0
    res = load_json(str(rf))
    assert res["p95_ms"] == 10.0 and 0.0 <= res["quality_score"] <= 100.0 and "task_results" in res


# ---------------------------------------------------------------- chaos
def test_chaos_against_mock_admin(mock_server, in_tmp):
    from kvmini.harness import chaos

    mock, url = mock_server
    rc = chaos.main(["--service", "mock-llm", "--url", url, "--model", "mock-llm", "--admin-url", url, "--requests", "20", "--concurrency", "4", "--max-tokens", "4",
                     "--faults", "pod-delete", "network-degrade", "--runs-root", str(in_tmp / "runs"), "--out", "resilience_table.json", "--ready-timeout", "5"])
    assert rc == 0
    doc = load_json("resilience_table.json")
    rows = {r["fault"]: r for r in doc["results"]}
    assert list(rows) == ["baseline", "pod-delete", "network-degrade"]
    for r in rows.values():
        assert set(r) >= {"fault", "mttr_s", "p95_ms", "slo_gate", "run_dir"}
        assert r["p95_ms"] is not None and os.path.exists(os.path.join(r["run_dir"], "results.json"))
    assert rows["network-degrade"]["p95_ms"] > rows["baseline"]["p95_ms"] + 30  # +50 ms injected
    assert rows["network-degrade"]["error_rate"] > 0
    assert mock.extra_latency_ms == 0.0 and mock.cfg.error_rate == 0.0  # cleared afterwards
    assert "chaos_mock-llm_pod_delete_" in rows["pod-delete"]["run_dir"]


def test_chaos_admin_mode_mttr_is_the_time_the_server_stays_away(mock_server, in_tmp):
    """A restart fault keeps the mock away for a second (503 on /health and on the API): the
    recorded MTTR is that second, not the round trip of one /health request; a server that does
    not come back within --ready-timeout has no MTTR (reference: ``chaos/chaos-harness.sh:70-80``
    waits for the pod to be Ready again)."""
    from kvmini.harness import chaos

    _mock, url = mock_server
    rc = chaos.main(["--service", "mock-llm", "--url", url, "--model", "mock-llm", "--admin-url", url, "--admin-down-s", "1.0", "--requests", "8", "--concurrency", "2", "--max-tokens", "2",
                     "--faults", "pod-delete", "--skip-baseline", "--runs-root", str(in_tmp / "runs"), "--out", "t.json", "--ready-timeout", "10"])
    assert rc == 0
    row = load_json("t.json")["results"][-1]
    assert row["fault"] == "pod-delete" and 0.8 <= row["mttr_s"] <= 3.0, row
    assert row["error_rate"] == 0  # the benchmark ran after recovery
    rc = chaos.main(["--service", "mock-llm", "--url", url, "--model", "mock-llm", "--admin-url", url, "--admin-down-s", "3.0", "--requests", "4", "--concurrency", "2", "--max-tokens", "2",
                     "--faults", "container-kill", "--skip-baseline", "--runs-root", str(in_tmp / "runs"), "--out", "t2.json", "--ready-timeout", "0.5"])
    assert load_json("t2.json")["results"][-1]["mttr_s"] is None
    import time

    time.sleep(2.7)  # (the shared mock comes back before the next test)


def test_chaos_uninjectable_fault_is_recorded(in_tmp, mock_server):
    from kvmini.harness import chaos

    _m, url = mock_server
    # the fake backend has no pods for this service: injection fails, the row says so
    chaos.main(["--service", "nope", "--url", url, "--skip-baseline", "--faults", "pod-delete", "--cluster-backend", "kubectl", "--out", "t.json", "--requests", "5"]) if False else None
    from kvmini.cluster import get_backend

    class A:  # minimal namespace
        pass

    a = chaos.main.__globals__["argparse"].Namespace(namespace="ns", service="svc", url=url, model=None, requests=5, concurrency=2, max_tokens=2, profile=None, slo=None, cost_file="cost.yaml", api_key=None,
                                                      faults=["bogus"], skip_baseline=True, ready_timeout=1.0, cluster_backend="fake", admin_url=None, runs_root=str(in_tmp / "runs"), out="t.json", fail_on_gate=False)
    r = chaos.ChaosRunner(a)
    r.backend.inject_fault = lambda *x: (_ for _ in ()).throw(RuntimeError("no pod"))
    r.run()
    row = load_json("t.json")["results"][0]
    assert row["mttr_s"] is None and "not injected" in row["note"]
    assert get_backend("fake").name == "fake"


# ---------------------------------------------------------------- quant sweep
def test_pareto_mask_direction_and_missing_values():
    from kvmini.harness.quantization_sweep import pareto_mask

    rows = [
        {"p95_ms": 100, "cost_per_1k_tokens": 1.0, "throughput_rps": 10, "quality_score": 90},
        {"p95_ms": 110, "cost_per_1k_tokens": 1.1, "throughput_rps": 9, "quality_score": 85},  # dominated
        {"p95_ms": 90, "cost_per_1k_tokens": 2.0, "throughput_rps": 12, "quality_score": None},
    ]
    assert pareto_mask(rows) == [True, False, True]
    assert pareto_mask([]) == []
    assert pareto_mask([{"x": 1}]) == [True]


def test_quant_sweep_config_and_analysis(in_tmp):
    from kvmini.harness import quantization_sweep as qs

    assert qs.main(["--create-config", "--config", "q.yaml"]) == 0
    import yaml

    cfg = yaml.safe_load(open("q.yaml"))
    assert len(cfg["sweep_configurations"]) == 6 and cfg["benchmark"]["requests"] == 100
    # no two rows measure the same (weights, KV, context, decoding) point
    pts = [(c["quantization"], c["kv_cache_dtype"], c["max_model_len"], json.dumps(c["decoding"], sort_keys=True)) for c in cfg["sweep_configurations"]]
    assert len(set(pts)) == 6 and len({(q, kv) for q, kv, _c, _d in pts}) == 6
    assert qs.vllm_args_for(cfg["sweep_configurations"][3])[:4] == ["--quantization", "fp8", "--kv-cache-dtype", "fp8"]
    lt = qs.loadtest_args_for(cfg["sweep_configurations"][3]["decoding"], str(in_tmp), "fp8-best-of-4")
    assert "--extra-openai-json" in lt and json.load(open(lt[-1]))["use_beam_search"] is True
    id1 = qs.config_run_id(cfg["sweep_configurations"][0]).rsplit("_", 1)[0]
    id2 = qs.config_run_id(dict(cfg["sweep_configurations"][0])).rsplit("_", 1)[0]
    assert id1 == id2 and id1.startswith("quant_sweep_")

    runner = qs.QuantizationSweepRunner(cfg, str(in_tmp / "out"), cluster_backend="fake")
    m = runner.manifest_for(cfg["sweep_configurations"][2])
    assert m["metadata"]["name"] == "llama3-8b-sweep-fp8-weights"
    assert "--quantization" in m["spec"]["predictor"]["model"]["args"]
    assert "amd.com/gpu" in json.dumps(m) and "nvidia" not in json.dumps(m)
    runner.results = [
        {"config_name": "baseline", "quantization": "none", "kv_cache_dtype": "auto", "p95_ms": 300.0, "throughput_rps": 40.0, "cost_per_1k_tokens": 0.003, "quality_score": 80.0, "error": ""},
        {"config_name": "fp8", "quantization": "fp8", "kv_cache_dtype": "fp8", "p95_ms": 200.0, "throughput_rps": 60.0, "cost_per_1k_tokens": 0.002, "quality_score": 79.0, "error": ""},
        {"config_name": "slow", "quantization": "awq", "kv_cache_dtype": "auto", "p95_ms": 400.0, "throughput_rps": 30.0, "cost_per_1k_tokens": 0.004, "quality_score": 70.0, "error": ""},
        runner.error_result({"name": "broken", "quantization": "gptq"}, "timeout"),
    ]
    runner.save_intermediate()
    report = runner.generate_analysis()
    assert report and "Pareto" in open(report).read()
    pareto = list(csv.DictReader(open(in_tmp / "out" / "pareto_optimal_configs.csv")))
    assert [r["config_name"] for r in pareto] == ["fp8", "baseline"]  # sorted by cost; 'slow' is dominated
    assert os.path.exists(in_tmp / "out" / "heatmap_p95_ms.svg")
    # the analysis can be rebuilt from the intermediate CSV alone
    assert qs.main(["--config", "q.yaml", "--output-dir", str(in_tmp / "out2"), "--analyze-only", str(in_tmp / "out" / "sweep_results_intermediate.csv"), "--cluster-backend", "fake"]) == 0
    assert os.path.exists(in_tmp / "out2" / "sweep_report.html")


# ---------------------------------------------------------------- A/B + backend comparison
def test_ab_compare_endpoints(in_tmp):
    from kvmini.harness import ab_compare
    from kvmini.server.mock import MockConfig, start_mock_in_thread

    _a, url_a, stop_a = start_mock_in_thread(MockConfig(ttft_ms=2.0, tpot_ms=0.1))
    _b, url_b, stop_b = start_mock_in_thread(MockConfig(ttft_ms=25.0, tpot_ms=0.1))
    try:
        rc = ab_compare.main(["--backends", "fast,slow", "--model", "mock-llm", "--profile", "standard", "--requests", "20", "--concurrency", "4", "--toggle-streaming", "--wait-between", "0",
                              "--endpoints", f"fast={url_a}", f"slow={url_b}", "--run-dir", "ab"])
    finally:
        stop_a()
        stop_b()
    assert rc == 0
    rows = list(csv.DictReader(open("ab/ab_comparison.csv")))
    assert [(r["backend"], r["streaming"]) for r in rows] == [("fast", "true"), ("fast", "false"), ("slow", "true"), ("slow", "false")]
    assert list(rows[0]) == ab_compare.CSV_COLS
    rep = load_json("ab/comparison_report.json")
    assert rep["winners"]["fastest_ttft"]["backend"] == "fast"
    assert set(rep["streaming_vs_non_streaming"]) == {"fast", "slow"}
    assert rep["comparison_metadata"]["streaming_tested"] is True
    # measured per-token time (the reference's tpot_ms column was always empty); no zero placeholders
    assert all(float(r["mean_tpot_ms"]) > 0 for r in rows)
    assert all(r["mean_ttft_ms"] == "" for r in rows if r["streaming"] == "false")  # a non-streamed answer has no first token
    assert rep["winners"]["fastest_ttft"]["streaming"] == "true" and rep["winners"]["fastest_tpot"] is not None
    assert all(d["ttft_advantage_ms"] > 0 for d in rep["streaming_vs_non_streaming"].values())
    page = open("ab/ab_report.html").read()
    assert "<svg" in page and "Time per output token" in page and "fast" in page


def test_ab_compare_applies_partition_profile_to_deployed_manifests(in_tmp):
    from kvmini.cluster import FakeBackend
    from kvmini.harness import ab_compare

    ns = argparse.Namespace(backends="vllm-rocm", model="llama-3-8b", model_uri=None, namespace="ml", endpoints=None, cluster_backend="fake", ready_timeout=1.0,
                            partition_profile=os.path.join(ROOT, "profiles", "partition", "mi355x-cpx-nps4.yaml"), no_cleanup=True)
    runner = ab_compare.ABRunner(ns, {"requests": 1, "concurrency": 1, "max_tokens": 1, "pattern": "steady"})
    assert isinstance(runner.cluster, FakeBackend)
    applied = []
    runner.cluster.apply = lambda m: applied.append(m)
    runner.cluster.wait_ready = lambda *a, **k: None
    runner.cluster.get_isvc_url = lambda *a, **k: "http://x"
    assert runner.deploy("vllm-rocm") == "http://x"
    res = applied[0]["spec"]["predictor"]["model"]["resources"] if "model" in applied[0]["spec"]["predictor"] else applied[0]["spec"]["predictor"]["containers"][0]["resources"]
    gpu_keys = [k for k in res["limits"] if k.startswith("amd.com/")]
    assert gpu_keys and all("cpx" in k for k in gpu_keys), res


def test_compare_backends_reports(in_tmp, mock_server):
    from kvmini.harness import compare_backends as cb

    _m, url = mock_server
    prof = in_tmp / "p.yaml"
    prof.write_text("pattern: steady\nrequests: 20\nconcurrency: 4\nmax_tokens: 4\n")
    rc = cb.main(["--model", "mock-llm", "--profile", str(prof), "--backends", "native", "tgi", "--endpoints", f"native={url}", "--output-dir", "cmp", "--ready-timeout", "1", "--cluster-backend", "fake"])
    assert rc == 0
    doc = load_json("cmp/backend_comparison.json")
    by = {r["backend"]: r for r in doc["results"]}
    assert by["native"]["success"] and by["native"]["total_requests"] == 20 and by["native"]["p95_latency_ms"] > 0
    assert by["native"]["throughput_rps"] > 0
    assert doc["winners"]["p95_latency_ms"] == "native"
    assert os.path.exists("cmp/backend_comparison.html") and os.path.exists("cmp/backend_comparison.csv")
    s = cb.ComparisonSummary("m", "p", "t", [cb.BackendResult("a", "r", "m", "p", True, p95_latency_ms=5, throughput_rps=1), cb.BackendResult("b", "r", "m", "p", True, p95_latency_ms=3, throughput_rps=2), cb.BackendResult("c", "r", "m", "p", False)])
    assert s.get_winner("p95_latency_ms") == "b" and s.get_winner("throughput_rps") == "b" and s.get_winner("nonsense") is None


# ---------------------------------------------------------------- reference matrix / build-vs-perf
def test_reference_matrix_dry_run_validation_and_bom(in_tmp, capsys):
    from kvmini.harness import reference_matrix as rm
    from kvmini.utils import load_yaml

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg_path = os.path.join(root, "reference-matrix.yaml")
    assert rm.main(["--config", cfg_path, "--dry-run", "--gpu", "mi355x", "--output-dir", "art"]) == 0
    out = capsys.readouterr().out
    assert "Would run 4 configurations" in out and "mi355x + llama-3-8b + steady" in out
    r = rm.ReferenceRunner(load_yaml(cfg_path), "art", cluster_backend="fake")
    assert len(r.cells()) == 12
    assert r.validate_results({"error_rate": 0.0, "throughput_rps": 50, "cold_p95_ms": 200, "warm_p95_ms": 100}) == []
    probs = r.validate_results({"error_rate": 0.5, "throughput_rps": 1, "cold_p95_ms": 900, "warm_p95_ms": 100})
    assert len(probs) == 3
    run = in_tmp / "art" / "ref_x"
    run.mkdir(parents=True)
    (run / "results.json").write_text("{}")
    c1 = rm.directory_checksum(str(run))
    m = r.config["matrix"]
    bom = r.create_bom(str(run), m["gpus"][0], m["models"][0], m["traffic_patterns"][0], "unsigned (test)")
    txt = open(bom).read()
    assert c1 in txt and "mi355x" in txt and "kvmini bench" in txt
    assert rm.directory_checksum(str(run)) == c1  # BOM.md itself is excluded
    r.generate_summary([str(run)], 1)
    assert load_json("art/matrix_summary.json")["total_runs"] == 1 and load_json("art/results.json")["has_results"] is True


def test_build_vs_perf_helpers(tmp_path):
    from kvmini.harness import build_vs_perf as bvp
    from kvmini.utils import load_yaml

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prof = load_yaml(os.path.join(root, "profiles", "engine", "llama-3-8b.yaml"))
    args = bvp.serve_args_from_profile(prof)
    assert args[:2] == ["--max-num-seqs", "64"] and "--max-model-len" in args and "--no-graphs" not in args
    assert bvp.builder_time(prof, None) == 0.0
    dt = bvp.builder_time(prof, "python -c \"print('{model_name}', '{quantization_arg}', {pack_weights})\"")
    assert dt > 0
    with pytest.raises(KeyError):
        bvp.builder_time(prof, "echo {missing_key}")
    out = tmp_path / "t.csv"
    row = {c: 1 for c in bvp.COLUMNS}
    bvp.write_row(str(out), row)
    bvp.write_row(str(out), row)
    lines = open(out).read().strip().splitlines()
    assert len(lines) == 3 and lines[0].split(",") == bvp.COLUMNS


# ---------------------------------------------------------------- telemetry daemon
def test_telemetry_daemon_single_rank(tmp_path):
    from kvmini.parallel import telemetry_daemon as td

    out = tmp_path / "power.json"
    assert td.main(["--fake", "--duration", "0.3", "--gather-every", "0.1", "--out", str(out)]) == 0
    doc = load_json(str(out))
    assert doc["world_size"] == 1 and len(doc["samples"]) == 3
    assert doc["samples"][0]["watts"] > 0
    table = np.full((2, 16), np.nan)
    table[:, 0] = [0, 1]
    table[:, 3] = [500.0, 510.0]
    table[:, 5] = 2 * (1 << 20)
    txt = td._prom_text(table)
    assert 'gpu_package_power{gpu_id="1"} 510.000' in txt and 'gpu_used_vram{gpu_id="0"} 2.000' in txt


def test_every_cli_subcommand_resolves_and_prints_help(capsys):
    import importlib

    from kvmini.cli import COMMANDS, main

    for name, (mod, fn, _help) in COMMANDS.items():
        assert callable(getattr(importlib.import_module(mod), fn)), name
    with pytest.raises(SystemExit) as e:
        main(["fairness", "--help"])
    assert e.value.code == 0
    assert "--tenant-a-requests" in capsys.readouterr().out


def test_spec_delta_report_with_self_speculation_on_cpu(in_tmp):
    """`kvmini spec-delta`: serves the tiny model twice on CPU (plain, then early-exit
    self-speculation on damped weights), writes the delta report with the acceptance rate."""
    from kvmini.harness import spec_delta

    rc = spec_delta.main(["--model", "tiny", "--draft", "self:1", "--tail-damping", "0.02", "--num-speculative-tokens", "3", "--requests", "12", "--concurrency", "3", "--max-tokens", "10",
                          "--port", "8471", "--serve-args", "--device cpu --dtype fp32 --max-model-len 128", "--output-dir", "spec"])
    assert rc == 0
    doc = load_json("spec/spec_delta.json")
    assert doc["draft"] == "self:1" and doc["baseline"]["spec_proposed"] == 0
    assert doc["speculative"]["spec_proposed"] > 0 and doc["speculative"]["acceptance_rate"] > 0.5
    assert doc["baseline"]["error_rate"] == 0.0 and doc["speculative"]["error_rate"] == 0.0
    keys = {d["metric"] for d in doc["deltas"]}
    assert {"time_to_first_token_ms", "itl_p50_ms", "p95_ms", "tokens_per_sec"} <= keys
    page = open("spec/spec_delta.html").read()
    assert "acceptance rate" in page.lower() and "self:1" in page


def test_router_fails_over_when_a_replica_goes_away():
    """Router over two running servers: requests spread over both; after one stops, the request
    that finds it dead is retried on the other (no error reaches the client), the replica is
    marked down, and /health stays 200 while one replica is left."""
    import asyncio
    import json as _json
    import urllib.error
    import urllib.request

    from kvmini.server.httpd import run_server_in_thread
    from kvmini.server.mock import start_mock_in_thread
    from kvmini.server.router import Router

    _m1, u1, stop1 = start_mock_in_thread()
    _m2, u2, stop2 = start_mock_in_thread()
    router = Router([u1, u2])
    _srv, port, stop_router = run_server_in_thread(router.build_server)
    url = f"http://127.0.0.1:{port}"

    def chat():
        req = urllib.request.Request(url + "/v1/chat/completions", data=_json.dumps({"model": "mock-llm", "messages": [{"role": "user", "content": "hi"}], "max_tokens": 4}).encode(),
                                     headers={"Content-Type": "application/json"})
        try:
            with urllib.request.urlopen(req, timeout=10) as r:
                return r.status, r.headers.get("X-Kvmini-Replica")
        except urllib.error.HTTPError as e:
            return e.code, None

    try:
        seen = {chat()[1] for _ in range(6)}
        assert seen == {"0", "1"}
        stop2()
        codes = [chat() for _ in range(6)]
        assert all(c == 200 and rep == "0" for c, rep in codes), codes
        assert router.replicas[1].healthy is False and router.replicas[1].errors >= 1
        assert urllib.request.urlopen(url + "/health", timeout=5).status == 200
        metrics = urllib.request.urlopen(url + "/metrics", timeout=5).read().decode()
        assert 'kvmini_router_replica_healthy{replica="1"} 0' in metrics
    finally:
        stop_router()
        stop1()


def test_prefix_agreement_and_agreement_evaluator():
    """Fidelity score of the quantisation sweep: words that match the reference's greedy
    continuation before the first difference."""
    from kvmini.harness.quality import AgreementEvaluator, agreement_prompts, prefix_agreement

    assert prefix_agreement(["a b c d"], ["a b c d"]) == 1.0
    assert prefix_agreement(["a b c d"], ["a b x d"]) == 0.5  # what follows the first difference does not count
    assert prefix_agreement(["a b", ""], ["a b c", ""]) == 1.0
    assert prefix_agreement(["a b c d", "e f"], ["z", "e f"]) == 0.5
    ps = agreement_prompts(96)
    assert len(ps) == len(set(ps)) == 96
    ps = agreement_prompts(24)

    ref_texts = {p: f"w{i} x y z" for i, p in enumerate(ps)}
    ref = AgreementEvaluator("http://unused", "m", 24, 8, call_fn=lambda p, n: (ref_texts[p], 0.001, 4)).evaluate(None)
    assert ref["quality_score"] == 100.0 and len(ref["continuations"]) == 24
    half = AgreementEvaluator("http://unused", "m", 24, 8, call_fn=lambda p, n: (ref_texts[p].replace("y", "q"), 0.001, 4)).evaluate(ref["continuations"])
    assert half["quality_score"] == 100.0 and half["prefix_agreement"] == 50.0 and half["quality_metric"] == "greedy_top1_agreement"
    none = AgreementEvaluator("http://unused", "m", 24, 8, call_fn=lambda p, n: ("other " + ref_texts[p], 0.001, 4)).evaluate(ref["continuations"])
    assert none["quality_score"] == 0.0 and none["prefix_agreement"] == 0.0

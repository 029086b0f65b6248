"""Quantiles, histograms, cold/warm split and the analyzer, pinned to the golden
vectors of the reference's own CI fixture (.github/workflows/lint-test.yml:93-122)."""

import json
import math
import os
import shutil

import pytest

from kvmini.analysis.coldwarm import classify_requests_cold_warm, compute_cold_warm_metrics
from kvmini.analysis.histograms import compute_histogram, compute_histograms
from kvmini.analysis.quantiles import RollingQuantile, percentile, percentiles
from kvmini.analysis.reduce import add_alias_keys, analyze_rows, analyze_run, read_requests_csv
from kvmini.cluster import FakeBackend

DATA = os.path.join(os.path.dirname(__file__), "data")


def test_percentile_golden():
    data = [10, 20, 30, 40, 50, 60, 70, 80, 90, 100]
    assert abs(percentile(data, 0.5) - 55) < 1e-9
    assert abs(percentile(data, 0.95) - 95.5) < 1e-9
    assert percentile([7.0], 0.99) == 7.0
    assert math.isnan(percentile([], 0.5))
    assert percentile([1.0, float("nan"), 3.0], 0.5) == 2.0
    assert percentile(data, 1.5) == 100 and percentile(data, -1) == 10


@pytest.mark.parametrize("engine", ["python", "numpy", "auto"])
def test_percentiles_engines_agree(engine):
    import random

    rng = random.Random(1)
    a = [rng.lognormvariate(3, 1) for _ in range(5001)] + [float("nan")]
    want = [percentile(a, p) for p in (0.5, 0.95, 0.99)]
    got = percentiles(a, (0.5, 0.95, 0.99), engine=engine)
    assert got == pytest.approx(want, rel=0, abs=0)


def test_quantile_methods():
    a = list(range(1, 101))
    assert percentile(a, 0.95, "floor") == a[int(0.95 * 99)]  # rolling-p95 definition
    assert percentile(a, 0.95, "nearest_rank") == a[95]  # probe definition
    rq = RollingQuantile(window=50, p=0.95)
    assert rq.value() is None
    for x in range(200):
        rq.add(x)
    assert len(rq) == 50 and rq.value() == sorted(range(150, 200))[int(0.95 * 49)]


def test_histogram_semantics():
    h = compute_histograms([1, 2, 3, 4, 5] * 10, num_bins=5)
    assert len(h["bins"]) == 5 and h["total_samples"] == 50 and h["counts"] == [10] * 5
    assert compute_histogram([]) == {"bins": [], "counts": [], "bin_edges": []}
    d = compute_histogram([4.0, 4.0, float("nan")])
    assert d == {"bins": [4.0], "counts": [2], "bin_edges": [4.0, 4.0]}
    h = compute_histogram([0.0, 10.0], 20)
    assert h["counts"][0] == 1 and h["counts"][-1] == 1  # max lands in the last (closed) bin


def test_read_requests_csv_and_golden_results():
    rows = read_requests_csv(os.path.join(DATA, "ci_requests.csv"))
    assert len(rows) == 5 and rows[0]["status"] == 200
    r = analyze_rows(rows)
    assert r["p50_ms"] == pytest.approx(165.0)
    assert r["p95_ms"] == pytest.approx(214.0)
    assert r["p99_ms"] == pytest.approx(218.8)
    assert r["throughput_rps"] == pytest.approx(5.0)
    assert r["tokens_per_sec"] == pytest.approx(76.25)
    assert r["error_rate"] == pytest.approx(0.2)
    assert r["window"]["seconds"] == pytest.approx(0.8)
    assert r["requests"] == {"total": 5, "success": 4}
    assert r["per_token_p50_ms"] == pytest.approx(2.6389, abs=1e-4)
    # empty cold subset keeps the reference's short shape
    assert r["cold_count"] == 0 and r["cold_p95_ms"] is None and "cold_error_rate" not in r
    assert r["warm_count"] == 5 and r["warm_error_rate"] == pytest.approx(0.2)


def test_cold_warm_classification_matches_bruteforce():
    import random

    rng = random.Random(3)
    rows = [{"start_ms": rng.uniform(0, 200_000), "status": 200.0, "latency_ms": 10.0} for _ in range(500)]
    colds = [rng.uniform(0, 200) for _ in range(7)]
    classify_requests_cold_warm(rows, colds, 30.0)
    for r in rows:
        t = r["start_ms"] / 1000
        assert r["is_cold_start"] == any(c <= t <= c + 30.0 for c in colds)
    m = compute_cold_warm_metrics(rows)
    assert m["cold_count"] + m["warm_count"] == 500


def test_analyze_run_with_fake_cluster(tmp_path):
    run = tmp_path / "run"
    run.mkdir()
    shutil.copy(os.path.join(DATA, "ci_requests.csv"), run / "requests.csv")
    fb = FakeBackend()
    fb.add_pod(name="demo-llm-predictor-0", service="demo-llm", namespace="ns", started_at=1.2)  # inside [1.001, 1.801]
    (run / "io_probe.json").write_text(json.dumps({"rtt": {"p50_ms": 1.5, "p95_ms": 2.5}, "s3": {"avg_MBps": 90.0}}))
    (run / "power.json").write_text(json.dumps({"samples": [{"ts_s": 1.0 + 0.1 * i, "watts": 700.0, "gfx_util": 88.0, "vram_used": 2e10} for i in range(10)]}))
    res = analyze_run(str(run), "ns", "demo-llm", backend=fb, quiet=True)
    # 3 requests start at/after 1.2 s (the reference's predicate) + 1 that was in flight across it
    assert res["cold_start_count"] == 1 and res["cold_count"] == 4
    from kvmini.analysis.coldwarm import classify_requests_cold_warm
    from kvmini.analysis.reduce import read_requests_csv

    rows = read_requests_csv(str(run / "requests.csv"))
    assert sum(r["is_cold_start"] for r in classify_requests_cold_warm(rows, [1.2], spanning=False)) == 3
    assert res["network_rtt_p95_ms"] == 2.5 and res["s3_avg_MBps"] == 90.0
    assert res["gpu_power_watts_avg"] == pytest.approx(700.0) and res["gpu_util_avg"] == pytest.approx(88.0)
    assert res["energy_wh"] == pytest.approx(700.0 * 0.8 / 3600)
    assert res["energy_wh_per_1k_tokens"] == pytest.approx(700.0 * 0.8 / 3600 / 102 * 1000)
    assert (run / "requests_classified.csv").read_text().splitlines()[0].endswith("is_cold_start")
    # merge-update: a later stage's keys survive a re-run of analyze
    p = run / "results.json"
    d = json.loads(p.read_text())
    d["cost_per_request"] = 0.5
    p.write_text(json.dumps(d))
    res2 = analyze_run(str(run), "ns", "demo-llm", backend=fb, quiet=True)
    assert res2["cost_per_request"] == 0.5
    # alias dialects
    assert res2["p95_latency_ms"] == res2["p95_ms"] and res2["throughput_req_per_sec"] == res2["throughput_rps"]
    assert res2["total_requests"] == 5 and res2["successful_requests"] == 4


def test_alias_keys_do_not_clobber_with_none():
    r = add_alias_keys({"p95_ms": 10.0, "requests": {"total": 1, "success": 1}, "avg_ttft_ms": 3.0})
    assert r["avg_ttft_ms"] == 3.0 and r["p95_latency_ms"] == 10.0


def test_rolling_quantile_incremental_equals_resorting_the_window():
    """The windowed quantile kept by insertion / removal equals sorting the last `window`
    observations on every read (what the reference's RollingP95 does,
    scripts/fairness_dual_tenant.py:46-65), duplicates and all three definitions included."""
    import random

    from kvmini.analysis.quantiles import RollingQuantile, percentile

    rng = random.Random(3)
    for method in ("floor", "linear", "nearest_rank"):
        for window in (1, 7, 50):
            rq = RollingQuantile(window=window, p=0.95, method=method)
            seen = []
            assert rq.value() is None and len(rq) == 0
            for i in range(400):
                x = float(rng.choice([1, 2, 2, 3, 5, 8, 13])) if i % 3 else rng.random() * 100
                rq.add(x)
                seen.append(x)
                assert rq.value() == percentile(seen[-window:], 0.95, method)
                assert len(rq) == min(window, len(seen))

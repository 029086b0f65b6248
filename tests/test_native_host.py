"""C++ host routines (csrc/host_routines.cpp via kvmini.analysis.native_host) against the
Python / numpy implementations that define their semantics."""

import math
import random

import numpy as np
import pytest

from kvmini.analysis import coldwarm, native_host
from kvmini.analysis.quantiles import RollingQuantile, percentiles, rolling_quantile
from kvmini.harness import quantization_sweep
from kvmini.harness.quality import ParetoAnalyzer

pytestmark = pytest.mark.skipif(not native_host.available(), reason="libkvmini_rt.so not built")


def _numpy_frontier(p):
    le = (p[None, :, :] <= p[:, None, :]).all(-1)
    lt = (p[None, :, :] < p[:, None, :]).any(-1)
    return ~(le & lt).any(1)


@pytest.mark.parametrize("n,d", [(1, 3), (7, 2), (200, 3), (513, 4)])
def test_pareto_mask_equals_numpy(n, d):
    rng = np.random.default_rng(n)
    p = np.round(rng.random((n, d)) * 8) / 8  # coarse grid: ties and duplicates
    assert (native_host.pareto_mask(p) == _numpy_frontier(p)).all()
    assert (ParetoAnalyzer.frontier_mask(p) == _numpy_frontier(p)).all()  # n >= 64 goes native


def test_pareto_mask_nan_neutral_equals_sweep_rule():
    rng = np.random.default_rng(3)
    rows = []
    for _ in range(150):
        r = {"p95_ms": float(rng.integers(1, 20)), "cost_per_1k_tokens": float(rng.integers(1, 20)), "throughput_rps": float(rng.integers(1, 20))}
        if rng.random() < 0.3:
            r["energy_per_1k_tokens_wh"] = float(rng.integers(1, 9))
        rows.append(r)
    fast = quantization_sweep.pareto_mask(rows)  # 150 rows: native
    saved = native_host.available
    try:
        native_host.available = lambda: False
        slow = quantization_sweep.pareto_mask(rows)
    finally:
        native_host.available = saved
    assert fast == slow and any(fast) and not all(fast)


def test_classify_cold_equals_python(monkeypatch):
    rnd = random.Random(5)
    rows = [{"start_ms": rnd.uniform(0, 600_000), "latency_ms": rnd.choice([0, "", None, rnd.uniform(10, 40_000)]), "status": 200} for _ in range(3000)]
    cold = [rnd.uniform(0, 600) for _ in range(5)]
    for spanning in (True, False):
        native = [r["is_cold_start"] for r in coldwarm.classify_requests_cold_warm([dict(r) for r in rows], cold, 30.0, spanning)]
        monkeypatch.setattr(coldwarm, "_NATIVE_MIN_ROWS", 10**9)
        python = [r["is_cold_start"] for r in coldwarm.classify_requests_cold_warm([dict(r) for r in rows], cold, 30.0, spanning)]
        monkeypatch.setattr(coldwarm, "_NATIVE_MIN_ROWS", 2048)
        assert native == python and any(native) and not all(native)


@pytest.mark.parametrize("method", ["linear", "floor", "nearest_rank"])
def test_quantiles_equal_oracle(method):
    rnd = random.Random(11)
    vals = [rnd.choice([float("nan"), rnd.gauss(100, 30)]) if rnd.random() < 0.05 else rnd.gauss(100, 30) for _ in range(5001)]
    ps = (0.0, 0.5, 0.95, 0.99, 1.0)
    got = native_host.quantiles(vals, ps, method)
    want = percentiles(vals, ps, method, engine="python")
    assert got == want
    assert math.isnan(native_host.quantiles([], (0.5,))[0])


@pytest.mark.parametrize("window,method", [(1, "floor"), (50, "floor"), (600, "linear"), (64, "nearest_rank")])
def test_rolling_quantile_equals_python(window, method):
    rnd = random.Random(window)
    a = RollingQuantile(window, 0.95, method)
    b = native_host.NativeRollingQuantile(window, 0.95, method)
    assert b.value() is None and len(b) == 0
    for i in range(2000):
        x = float(rnd.randint(0, 40))
        a.add(x)
        assert b.add(x) == a.value()
        assert len(a) == len(b)
    xs = [float(rnd.randint(0, 40)) for _ in range(500)]
    out = b.add_many(xs)
    for x in xs:
        a.add(x)
    assert out[-1] == a.value() == b.value()
    b.close()


def test_rolling_quantile_factory_picks_engine():
    assert isinstance(rolling_quantile(50), RollingQuantile)
    assert isinstance(rolling_quantile(1024), native_host.NativeRollingQuantile)
    assert isinstance(rolling_quantile(1024, engine="python"), RollingQuantile)

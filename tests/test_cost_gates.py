"""Cost estimator, pricing, planner, energy integration, SLO gate, canary, validator."""

import json
import os
import shutil
import subprocess
import sys
import tarfile

import pytest

from kvmini.cluster import FakeBackend, make_pod
from kvmini.cost import calculator, estimator, planner
from kvmini.cost.pricing import load_pricing, pricing_from_dict
from kvmini.gates import canary, slo
from kvmini.gates.validate_config import ConfigValidator
from kvmini.telemetry.energy import PowerSample, trapezoidal_wh
from kvmini.utils import dns1123_label, parse_k8s_quantity, redact_cmd, stable_hash

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "tests", "data")


def test_parse_k8s_quantity():
    assert parse_k8s_quantity("500m") == pytest.approx(0.5)
    assert parse_k8s_quantity("2Gi") == 2147483648
    assert parse_k8s_quantity("1") == 1.0
    assert parse_k8s_quantity("1.5G") == 1.5e9
    assert parse_k8s_quantity("1e3") == 1000.0
    assert parse_k8s_quantity(None) == 0.0 and parse_k8s_quantity("junk") == 0.0


def test_utils_misc():
    assert redact_cmd(["x", "--api-key", "secret", "HF_TOKEN=abc", "--n", "1"]) == ["x", "--api-key", "***", "HF_TOKEN=***", "--n", "1"]
    assert dns1123_label("Llama_3.1 8B!!") == "llama-3-1-8b"
    assert stable_hash({"a": 1, "b": 2}) == stable_hash({"b": 2, "a": 1})


def test_pricing_both_schemas_and_fuzzy_match():
    p = load_pricing(os.path.join(ROOT, "cost.yaml"))
    assert p.gpu_hourly("AMD_Instinct_MI355X") == 3.45
    assert p.gpu_hourly("AMD Instinct MI355X OAM") == 3.45  # amd-smi market name
    assert p.gpu_hourly("AMD-Instinct-MI300X") == 1.99
    assert p.gpu_hourly(None) == p.gpu_default and p.gpu_hourly("Unknown GPU") == p.gpu_default
    q = pricing_from_dict({"gpus": {"mi355x": {"hourly": 4.0}}, "cpu_per_hour": 0.05, "memory_per_gb_hour": 0.006})
    assert q.gpu_hourly("mi355x") == 4.0 and q.cpu_hourly_per_core == 0.05 and q.mem_hourly_per_gib == 0.006


def _run_dir(tmp_path):
    run = tmp_path / "run"
    run.mkdir()
    shutil.copy(os.path.join(DATA, "ci_requests.csv"), run / "requests.csv")
    return run


def test_cost_estimator_resource_seconds(tmp_path):
    run = _run_dir(tmp_path)
    fb = FakeBackend()
    fb.add_node("node-0")
    fb.add_pod(name="p0", service="svc", namespace="ns", started_at=0.0, node="node-0", gpus=1, cpu="4", memory="16Gi", sidecar=True)
    f = estimator.estimate_cost(str(run), "ns", "svc", os.path.join(ROOT, "cost.yaml"), fb, quiet=True)
    b = f["cost_breakdown"]
    assert b["gpu_seconds"] == pytest.approx(0.8) and b["cpu_core_seconds"] == pytest.approx(3.2) and b["mem_gib_seconds"] == pytest.approx(12.8)
    total = (0.8 * 3.45 + 3.2 * 0.04 + 12.8 * 0.005) / 3600 * 1.10
    assert b["total_cost"] == pytest.approx(total) and b["gpu_product_label"] == "AMD_Instinct_MI355X"
    assert f["cost_per_request"] == pytest.approx(total / 4) and f["cost_per_1k_tokens"] == pytest.approx(total / 102 * 1000)
    assert f["warm_requests"] == 4 and f["cold_requests"] == 0
    res = json.loads((run / "results.json").read_text())
    assert res["cost_breakdown"]["gpu_hourly"] == 3.45
    # no pods: no crash, null costs, a warning
    g = estimator.estimate_cost(str(run), "ns", "missing", os.path.join(ROOT, "cost.yaml"), fb, quiet=True)
    assert g["cost_per_request"] is None and g["cost_warnings"]


def test_cost_partition_fraction_and_pod_window():
    p = pricing_from_dict({})
    assert estimator.gpu_units({"amd.com/gpu": "2"}, p) == 2.0
    assert estimator.gpu_units({"amd.com/gpu-cpx": "1"}, p) == 0.125
    assert estimator.gpu_units({"amd.com/gpu-dpx": "1", "cpu": "4"}, p) == 0.5
    pods = [make_pod("a", "s", started_at=10.0, finished_at=20.0, gpus=1), make_pod("b", "s", started_at=15.0, gpus=1)]
    rs = estimator.sum_resource_seconds(pods, 12.0, 30.0, p)
    assert rs["gpu_seconds"] == pytest.approx(8.0 + 15.0)


def test_legacy_calculator(tmp_path, capsys):
    f = tmp_path / "r.txt"
    f.write_text("100 200\n300 200\n999 500\n")
    assert calculator.main([str(f), "3.6", "--requests-per-1k-tokens", "10"]) == 0
    assert "Cost per 1K tokens: $0.002000" in capsys.readouterr().out
    assert calculator.main([str(tmp_path / "missing"), "1"]) == 1


def test_planner(tmp_path):
    pl = planner.CapacityPlanner(os.path.join(ROOT, "cost.yaml"))
    plan = pl.plan_capacity(100.0, 500.0)
    recs = plan["recommendations"]
    assert [r["rank"] for r in recs] == list(range(1, len(recs) + 1)) and recs == sorted(recs, key=lambda r: r["score"])
    mi355 = next(o for o in plan["capacity_requirements"]["options"] if o["gpu_type"] == "AMD_Instinct_MI355X")
    assert mi355["headroom_factor"] == pytest.approx(1.8) and mi355["cpu_cores"] == mi355["gpu_count"] * 4
    csvf = tmp_path / "sweep.csv"
    csvf.write_text("throughput_rps,p95_ms,tensor_parallel_size\n80,300,1\n160,320,2\n40,280,0\n")
    pl.calibrate_from_sweep_csv(str(csvf))
    assert pl.calibrated_baseline == {"rps_per_gpu": 80.0, "p95_ms": 310.0}
    out = tmp_path / "plan.json"
    assert planner.main(["--target-rps", "50", "--p95-budget", "400", "--json", str(out), "--output", str(tmp_path / "r.md"), "--cost-file", os.path.join(ROOT, "cost.yaml")]) == 0
    assert "Recommendations" in (tmp_path / "r.md").read_text() and json.loads(out.read_text())["planning_inputs"]["target_rps"] == 50


def test_trapezoid_energy():
    s = [PowerSample(float(t), 360.0) for t in range(0, 11)]
    assert trapezoidal_wh(s, 0, 10) == pytest.approx(1.0)
    s[5] = PowerSample(5.0, None)
    assert trapezoidal_wh(s, 0, 10) == pytest.approx(1.0)  # missing sample bridged
    assert trapezoidal_wh(s, 2.5, 7.5) == pytest.approx(360.0 * 4 / 3600)
    assert trapezoidal_wh(s[:1], 0, 10) == 0.0 and trapezoidal_wh(s, 5, 5) == 0.0


def test_energy_integrate_cli(tmp_path):
    from kvmini.telemetry import energy

    run = _run_dir(tmp_path)
    (run / "power.json").write_text(json.dumps({"samples": [{"ts_s": 0.5 + 0.1 * i, "watts": 720.0, "per_gpu": [{"index": 0, "watts": 720.0}]} for i in range(20)]}))
    assert energy.main(["integrate", "--run-dir", str(run), "--idle-tax", "series", "--merge-results"]) == 0
    e = json.loads((run / "energy.json").read_text())
    assert e["Wh_active"] == pytest.approx(720.0 * 0.7 / 3600, rel=1e-6)  # samples inside [1.001, 1.801]: 1.1 .. 1.8
    assert e["Wh_per_1k_tokens_active"] == pytest.approx(e["Wh_active"] / 102 * 1000)
    assert e["Wh_idle_tax"] > 0 and e["Wh_active_per_gpu"]["0"] == pytest.approx(e["Wh_active"])
    assert json.loads((run / "results.json").read_text())["energy_wh_per_1k_tokens"] == pytest.approx(e["Wh_per_1k_tokens_active"])
    # collect with the synthetic source
    out = tmp_path / "power2.json"
    assert energy.main(["collect", "--out", str(out), "--interval", "0.05", "--duration", "0.2", "--fake"]) == 0
    assert len(json.loads(out.read_text())["samples"]) >= 3


def test_slo_gate_exit_codes():
    good, bad = os.path.join(DATA, "results_good.json"), os.path.join(DATA, "results_bad.json")
    sl = os.path.join(ROOT, "slo.json")
    assert slo.main(["--results", good, "--energy", os.path.join(DATA, "energy_good.json"), "--slo", sl]) == 0
    assert slo.main(["--results", bad, "--slo", sl]) == 3
    summary, failures = slo.evaluate(json.load(open(sl)), json.load(open(bad)))
    assert set(failures) == {"p95_ms", "error_rate", "$per_1k_tokens", "cold_multiplier", "Wh_per_1k_tokens"}
    # missing metrics are skipped unless --strict
    assert slo.evaluate({"p95_ms": 10}, {})[1] == [] and slo.evaluate({"p95_ms": 10}, {}, strict=True)[1]
    fair = {"summary": {"A": {"p95_ms": 100}, "B": {"p95_ms": 900}, "throughput_share": {"A": 0.8, "B": 0.2}}, "guard_triggers": 7}
    _s, f = slo.evaluate({"fairness": {"tenant_p95_ms_max": 500, "throughput_share_diff_max": 0.3, "guard_triggers_max": 5}}, {}, fair=fair)
    assert set(f) == {"fairness_tenantB_p95_ms", "fairness_share_diff", "fairness_guard_triggers"}


def test_canary_dirs_and_bundles(tmp_path):
    good, bad = json.load(open(os.path.join(DATA, "results_good.json"))), json.load(open(os.path.join(DATA, "results_bad.json")))
    d, reg = canary.compare(good, bad)
    assert reg and d["p95_ms"]["regression"] and d["throughput_rps"]["regression"]
    assert not canary.compare(good, good)[1]
    assert canary.compare({"error_rate": 0.0}, {"error_rate": 0.01})[1]
    a, b = tmp_path / "a", tmp_path / "b"
    a.mkdir()
    b.mkdir()
    (a / "results.json").write_text(json.dumps(good))
    (b / "results.json").write_text(json.dumps(bad))
    tgz = tmp_path / "b.tar.gz"
    with tarfile.open(tgz, "w:gz") as t:
        t.add(b / "results.json", arcname="bundle/results.json")
    assert canary.main(["--baseline", str(a), "--candidate", str(tgz), "--out", str(tmp_path / "c.html")]) == 2
    assert canary.main(["--baseline", str(a), "--candidate", str(a), "--out", str(tmp_path / "c2.html")]) == 0
    assert "REGRESS" in (tmp_path / "c.html").read_text() and json.loads((tmp_path / "c.json").read_text())["p95_ms"]["regression"]


def test_validator_rules():
    v = ConfigValidator()
    assert not v.validate_profile({"vllm_features": {"quantization": "awq"}, "model_requirements": {"compatible_formats": ["gptq"]}, "concurrency": 4})
    assert any("not compatible" in e for e in v.errors) and any("Suggestion" in w and "Docs" in w for w in v.warnings)
    assert not v.validate_profile({"_gpu_memory_gb": 40, "validation_hints": {"model_size_hint": "70b"}, "concurrency": 4})
    assert any("insufficient" in e.lower() for e in v.errors)
    assert v.validate_profile({"_gpu_memory_gb": 288, "validation_hints": {"model_size_hint": "70b"}, "concurrency": 4})  # fits one MI355X
    assert v.validate_profile({"_gpu_memory_gb": 48, "validation_hints": {"model_size_hint": "70b"}, "vllm_features": {"quantization": "awq"}, "model_requirements": {"compatible_formats": ["awq"]}, "concurrency": 4})
    assert not v.validate_profile({"vllm_features": {"num_scheduler_steps": 4}})
    assert v.validate_profile({"vllm_features": {"num_scheduler_steps": 4}, "max_tokens": 64, "concurrency": 8})
    v.validate_profile({"vllm_features": {"device": "cpu"}, "concurrency": 1, "requests": 0, "max_tokens": 4096})
    assert len(v.warnings) >= 3
    # fp8 is native on gfx950: no warning there, a warning on hardware without FP8 matrix cores
    v.validate_profile({"vllm_features": {"quantization": "fp8"}, "_gpu_name": "AMD Instinct MI355X", "concurrency": 8})
    assert not any("FP8" in w for w in v.warnings)
    v.validate_profile({"vllm_features": {"quantization": "fp8"}, "_gpu_name": "AMD Instinct MI250X", "concurrency": 8})
    assert any("FP8" in w for w in v.warnings)


def test_validator_autodetect_via_amd_smi(monkeypatch, tmp_path):
    """GPU memory auto-detection through the amd-smi CLI (stubbed): a 70b hint on a
    40 GiB GPU must fail validation."""
    from kvmini.gates import validate_config as vc

    class StubProc:
        returncode = 0
        stdout = json.dumps([{"gpu": 0, "asic": {"market_name": "AMD Instinct MI210"}, "vram": {"size": {"value": 40960, "unit": "MB"}}}, {"gpu": 1, "asic": {"market_name": "AMD Instinct MI210"}, "vram": {"size": {"value": 65536, "unit": "MB"}}}])

    monkeypatch.setattr(subprocess, "run", lambda *a, **k: StubProc())
    assert vc.detect_gpu() == {"memory_gb": 40, "name": "AMD Instinct MI210"}
    prof = tmp_path / "p.yaml"
    prof.write_text("requests: 10\nconcurrency: 4\nmax_tokens: 32\nvalidation_hints:\n  model_size_hint: 70b\n")
    assert vc.main(["--profile", str(prof)]) == 1
    assert vc.main(["--profile", str(prof), "--gpu-memory-gb", "288"]) == 0


def test_every_shipped_profile_validates_and_loads():
    from kvmini.config.profiles import list_profiles, load_profile, resolve_load_settings

    groups = list_profiles()
    assert len(groups["runners/profiles"]) >= 19 and len(groups["profiles/partition"]) == 5 and len(groups["profiles/engine"]) >= 6
    for rel in groups["runners/profiles"]:
        if "/partition/" in rel:
            continue
        p = load_profile(os.path.join(ROOT, rel))
        cfg = dict(p.raw)
        cfg["_gpu_memory_gb"] = 288
        assert ConfigValidator().validate_profile(cfg), rel
    p = load_profile("speculative-decoding")
    assert "--speculative-model" in p.vllm_args() and "--use-v2-block-manager" in p.vllm_args()
    s = resolve_load_settings({"requests": None, "concurrency": 3, "max_tokens": None, "pattern": None}, p)
    assert s["concurrency"] == 3 and s["requests"] == 200 and s["max_tokens"] == 128  # CLI > profile > default
    assert load_profile("fp8").name == "fp8" and load_profile("tool-calling").openai_tools()[0]["function"]["name"] == "get_weather"
    assert load_profile("structured-output").test_schema["required"] == ["name", "age"]


def test_native_runtime_bindings_cpu_side():
    """The parts of libkvmini_rt.so that need no GPU: record grouping, trapezoid."""
    import numpy as np

    from kvmini.telemetry import native_rt as rt

    rec = np.array([[1.0, 0, 100, 50, 1e9, 60, 2000, 5], [1.0, 1, 110, 70, 2e9, 62, 2100, 6], [2.0, 0, 120, float("nan"), 1e9, 61, 2000, 7], [2.0, 1, 130, 75, 2e9, 63, 2100, 8]])
    s = rt.records_to_samples(rec)
    assert len(s) == 2 and s[0]["watts"] == 210.0 and s[0]["gfx_util"] == 60.0 and s[1]["gfx_util"] == 75.0
    assert [g["index"] for g in s[1]["per_gpu"]] == [0, 1] and s[1]["per_gpu"][0]["gfx_util"] is None
    # 100 W for 3 s with a NaN hole = 300 J
    assert rt.trapz_wh([0, 1, 2, 3], [100, 100, float("nan"), 100]) == pytest.approx(300 / 3600)
    sampler = rt.make_sampler(0.05)
    assert hasattr(sampler, "start") and hasattr(sampler, "stop") and hasattr(sampler, "to_power_json")


def test_long_name_alias_package_resolves_to_kvmini():
    import kserve_vllm_mini_amd as alias
    import kvmini.models.llama as real_llama
    from kserve_vllm_mini_amd.models import llama
    from kserve_vllm_mini_amd.parallel import telemetry_gather

    import kvmini.parallel.telemetry_gather as real_tg

    assert llama is real_llama and telemetry_gather is real_tg
    assert alias.ops.gpu_available is __import__("kvmini.ops").ops.gpu_available
    assert alias.utils.parse_k8s_quantity("500m") == 0.5

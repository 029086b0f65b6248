"""Capacity & budget planner (``kvmini plan``).

Same model as the reference planner (``planner.py:59-351``): per GPU type a
baseline (RPS per GPU at a p95), latency factor ``min(budget / p95, 2)``, GPU
count = ceil(target / effective RPS) x cold-start headroom x burst headroom,
4 CPU cores and 32 GB memory per GPU, monthly cost over 720 h with a regional
multiplier and 100 GB storage per GPU, warm-pool sizing, and a score
``cost/1000 + 1/utilisation`` that ranks the options.  Differences:

* baselines are AMD Instinct parts; the MI355X baseline is the measured headline
  of this repository's native engine (and is replaced by run history /
  ``--calibrate-csv`` when given — the reference loads run history and then
  discards it, ``planner.py:215-216``);
* pricing comes from the one ``cost.yaml`` schema shared with the estimator (the
  reference planner cannot read the reference's own cost.yaml:
  ``planner.py:125,130`` vs ``cost.yaml:2-16``);
* calibration reads CSVs with the stdlib (no pandas needed).
"""

from __future__ import annotations

import argparse
import csv
import datetime as _dt
import json
import math
import statistics
import sys
from pathlib import Path
from typing import Any, Dict, List, Optional

from .pricing import UnitPricing, load_pricing

# RPS per GPU / p95 at the headline mix (8B model, 64 output tokens).  MI355X: measured
# with the built-in engine (bench.py); the others scale with HBM bandwidth (decode is
# bandwidth-bound): MI325X 6.0, MI300X 5.3, MI250X 3.2 TB/s vs 8.0 TB/s.
GPU_BASELINES: Dict[str, Dict[str, float]] = {
    "AMD_Instinct_MI355X": {"rps_per_gpu": 60.0, "p95_ms": 350.0},
    "AMD_Instinct_MI325X": {"rps_per_gpu": 45.0, "p95_ms": 470.0},
    "AMD_Instinct_MI300X": {"rps_per_gpu": 40.0, "p95_ms": 530.0},
    "AMD_Instinct_MI250X": {"rps_per_gpu": 24.0, "p95_ms": 880.0},
}

DEFAULT_MIX: Dict[str, Any] = {
    "cold_start_multiplier": 1.2,
    "burst_multiplier": 1.5,
    "cpu_per_gpu": 4,
    "memory_gb_per_gpu": 32,
    "cold_start_time_s": 45,
    "cold_start_frequency": 0.1,
    "target_cold_time_s": 10,
}


class CapacityPlanner:
    def __init__(self, cost_config_path: Optional[str] = "cost.yaml", pricing: Optional[UnitPricing] = None):
        self.pricing = pricing or load_pricing(cost_config_path)
        self.calibrated_baseline: Optional[Dict[str, float]] = None
        self.calibration_source: Optional[str] = None

    # ---- calibration ------------------------------------------------------
    def calibrate_from_runs(self, run_dirs: List[str], gpus_per_replica: int = 1) -> None:
        """Median throughput / p95 of historical ``results.json`` files."""
        rps, p95 = [], []
        for d in run_dirs or []:
            f = Path(d) / "results.json"
            if not f.exists():
                continue
            try:
                r = json.loads(f.read_text())
            except ValueError:
                continue
            if isinstance(r.get("throughput_rps"), (int, float)) and isinstance(r.get("p95_ms"), (int, float)):
                rps.append(float(r["throughput_rps"]) / max(1, gpus_per_replica))
                p95.append(float(r["p95_ms"]))
        if rps:
            self.calibrated_baseline = {"rps_per_gpu": statistics.median(rps), "p95_ms": statistics.median(p95)}
            self.calibration_source = f"{len(rps)} run(s)"

    def calibrate_from_sweep_csv(self, csv_path: str) -> None:
        """Median per-GPU RPS and p95 of a sweep CSV with columns ``throughput_rps``,
        ``p95_ms`` and ``tensor_parallel_size`` (GPUs per replica)."""
        try:
            with open(csv_path, newline="") as f:
                rows = list(csv.DictReader(f))
        except OSError:
            return
        rps, p95 = [], []
        for r in rows:
            try:
                tp = float(r.get("tensor_parallel_size") or 0)
                if tp <= 0:
                    continue
                rps.append(float(r["throughput_rps"]) / tp)
                p95.append(float(r["p95_ms"]))
            except (KeyError, TypeError, ValueError):
                continue
        if rps:
            self.calibrated_baseline = {"rps_per_gpu": statistics.median(rps), "p95_ms": statistics.median(p95)}
            self.calibration_source = csv_path

    # ---- model ------------------------------------------------------------
    def _gpu_types(self) -> List[str]:
        names = [n for n in GPU_BASELINES if self.pricing.gpu_hourly(n) is not None]
        return names

    def _calculate_base_capacity(self, target_rps: float, p95_budget_ms: float, mix: Dict[str, Any]) -> Dict[str, Any]:
        options = []
        for gpu_type in self._gpu_types():
            base = dict(GPU_BASELINES[gpu_type])
            if self.calibrated_baseline and gpu_type == "AMD_Instinct_MI355X":
                base.update(self.calibrated_baseline)
            latency_factor = min(p95_budget_ms / base["p95_ms"], 2.0)
            effective = base["rps_per_gpu"] * latency_factor
            base_gpus = math.ceil(target_rps / effective)
            headroom = mix.get("cold_start_multiplier", 1.2) * mix.get("burst_multiplier", 1.5)
            gpus = math.ceil(base_gpus * headroom)
            options.append(
                {
                    "gpu_type": gpu_type,
                    "gpu_count": gpus,
                    "cpu_cores": gpus * mix.get("cpu_per_gpu", 4),
                    "memory_gb": gpus * mix.get("memory_gb_per_gpu", 32),
                    "effective_rps": effective,
                    "headroom_factor": headroom,
                }
            )
        return {"options": options}

    def _calculate_costs(self, capacity: Dict[str, Any], region: str) -> Dict[str, Any]:
        mult = self.pricing.regions.get(region, 1.0)
        hours = 24 * 30
        out = []
        for o in capacity["options"]:
            gpu = self.pricing.gpu_hourly(o["gpu_type"]) * o["gpu_count"] * hours * mult
            cpu = self.pricing.cpu_hourly_per_core * o["cpu_cores"] * hours * mult
            mem = self.pricing.mem_hourly_per_gib * o["memory_gb"] * hours * mult
            sto = self.pricing.storage_hourly_per_gib * o["gpu_count"] * 100 * hours * mult
            out.append(
                {
                    "gpu_type": o["gpu_type"],
                    "gpu_count": o["gpu_count"],
                    "costs": {
                        "gpu_monthly": round(gpu, 2),
                        "cpu_monthly": round(cpu, 2),
                        "memory_monthly": round(mem, 2),
                        "storage_monthly": round(sto, 2),
                        "total_monthly": round(gpu + cpu + mem + sto, 2),
                    },
                    "region": region,
                    "region_multiplier": mult,
                }
            )
        return {"cost_breakdown": out}

    @staticmethod
    def _calculate_warm_pool_sizing(target_rps: float, mix: Dict[str, Any]) -> Dict[str, Any]:
        cold_s = mix.get("cold_start_time_s", 45)
        freq = mix.get("cold_start_frequency", 0.1)
        per_min = target_rps * 60 * freq
        pool = max(math.ceil(per_min * (cold_s / 60)), max(1, math.ceil(target_rps * 0.1)))
        return {"warm_pool_size": pool, "cold_start_time_s": cold_s, "target_cold_time_s": mix.get("target_cold_time_s", 10), "expected_cold_requests_per_min": round(per_min, 1)}

    @staticmethod
    def _rationale(gpu_type: str, monthly: float, util: float) -> str:
        eff = "High efficiency" if util > 0.8 else "Good efficiency" if util > 0.6 else "Moderate efficiency" if util > 0.4 else "Low efficiency"
        tier = "Budget-friendly" if monthly < 1000 else "Mid-range cost" if monthly < 5000 else "Premium option"
        return f"{tier} with {gpu_type.replace('_', ' ')} GPUs. {eff} ({util * 100:.0f}% utilization)."

    def plan_capacity(self, target_rps: float, p95_budget_ms: float, mix_profile: Optional[Dict[str, Any]] = None, region: str = "us-west-2", run_history: Optional[List[str]] = None) -> Dict[str, Any]:
        mix = dict(DEFAULT_MIX)
        mix.update(mix_profile or {})
        if run_history:
            self.calibrate_from_runs(run_history)
        cap = self._calculate_base_capacity(target_rps, p95_budget_ms, mix)
        costs = self._calculate_costs(cap, region)
        warm = self._calculate_warm_pool_sizing(target_rps, mix)
        recs = []
        for o, c in zip(cap["options"], costs["cost_breakdown"]):
            monthly = c["costs"]["total_monthly"]
            util = target_rps / (o["gpu_count"] * o["effective_rps"])
            recs.append(
                {
                    "rank": 0,
                    "gpu_type": o["gpu_type"],
                    "gpu_count": o["gpu_count"],
                    "monthly_cost": monthly,
                    "cost_per_rps": round(monthly / target_rps, 2),
                    "gpu_utilization": round(util * 100, 1),
                    "warm_pool_size": warm["warm_pool_size"],
                    "score": round(monthly / 1000 + 1 / max(util, 0.1), 2),
                    "rationale": self._rationale(o["gpu_type"], monthly, util),
                }
            )
        recs.sort(key=lambda r: r["score"])
        for i, r in enumerate(recs):
            r["rank"] = i + 1
        return {
            "planning_inputs": {"target_rps": target_rps, "p95_budget_ms": p95_budget_ms, "region": region, "mix_profile": mix, "calibration": self.calibrated_baseline, "calibration_source": self.calibration_source},
            "capacity_requirements": cap,
            "cost_analysis": costs,
            "warm_pool_sizing": warm,
            "recommendations": recs,
            "generated_at": _dt.datetime.now().isoformat(),
        }

    @staticmethod
    def generate_report(plan: Dict[str, Any], output_file: Optional[str] = None) -> str:
        i = plan["planning_inputs"]
        lines = [
            "# Capacity & Budget Planning Report",
            "",
            "## Requirements",
            f"- **Target RPS**: {i['target_rps']}",
            f"- **P95 Latency Budget**: {i['p95_budget_ms']}ms",
            f"- **Region**: {i['region']}",
        ]
        if i.get("calibration"):
            lines.append(f"- **Calibrated from**: {i.get('calibration_source')} → {i['calibration']['rps_per_gpu']:.1f} RPS/GPU @ p95 {i['calibration']['p95_ms']:.0f} ms")
        lines += ["", "## Recommendations (Ranked by Cost-Efficiency)", ""]
        for r in plan["recommendations"]:
            lines += [
                f"### #{r['rank']}: {r['gpu_type'].replace('_', ' ')}",
                f"- **GPUs**: {r['gpu_count']} units",
                f"- **Monthly Cost**: ${r['monthly_cost']:,}",
                f"- **Cost per RPS**: ${r['cost_per_rps']}/RPS/month",
                f"- **GPU Utilization**: {r['gpu_utilization']}%",
                f"- **Warm Pool Size**: {r['warm_pool_size']} replicas",
                f"- **Rationale**: {r['rationale']}",
                "",
            ]
        w = plan["warm_pool_sizing"]
        lines += [
            "## Cold Start Mitigation",
            f"- **Recommended Warm Pool**: {w['warm_pool_size']} replicas",
            f"- **Cold Start Time**: {w['cold_start_time_s']}s",
            f"- **Expected Cold Requests**: {w['expected_cold_requests_per_min']}/min",
            "",
            "## Next Steps",
            "1. **Validate assumptions** by running benchmarks with your actual workload",
            "2. **Start with the #1 recommendation** and monitor performance",
            "3. **Adjust warm pool size** based on observed cold start patterns",
            "4. **Consider autoscaling** between min=warm_pool_size and max=gpu_count",
            "",
            f"*Generated at: {plan.get('generated_at', 'unknown')}*",
            "",
        ]
        text = "\n".join(lines)
        if output_file:
            Path(output_file).write_text(text)
            print(f"Planning report saved to {output_file}")
        return text


def load_mix_profile(path: Optional[str]) -> Dict[str, Any]:
    if path and Path(path).exists():
        from kvmini.utils import load_yaml

        data = load_yaml(path)
        if isinstance(data, dict):
            return data
    return dict(DEFAULT_MIX)


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(prog="kvmini plan", description="Capacity & Budget Planner")
    ap.add_argument("--target-rps", type=float, required=True)
    ap.add_argument("--p95-budget", type=float, required=True, help="P95 latency budget in milliseconds")
    ap.add_argument("--mix", default="profile.yaml", help="Traffic mix profile YAML")
    ap.add_argument("--region", default="us-west-2")
    ap.add_argument("--cost-file", default="cost.yaml")
    ap.add_argument("--runs", nargs="+", help="Historical run directories for calibration")
    ap.add_argument("--output", help="Output markdown report file")
    ap.add_argument("--calibrate-csv", help="Sweep CSV to calibrate baselines (throughput/p95)")
    ap.add_argument("--json", help="Output JSON plan file")
    a = ap.parse_args(argv)
    planner = CapacityPlanner(a.cost_file)
    if a.calibrate_csv:
        planner.calibrate_from_sweep_csv(a.calibrate_csv)
    plan = planner.plan_capacity(a.target_rps, a.p95_budget, load_mix_profile(a.mix), a.region, a.runs)
    json_file = a.json or "capacity_plan.json"
    Path(json_file).write_text(json.dumps(plan, indent=2))
    print(f"Capacity plan generated: {json_file}")
    planner.generate_report(plan, a.output or "capacity_report.md")
    rec = plan["recommendations"][0]
    print(f"\nTop Recommendation: {rec['gpu_count']}x {rec['gpu_type']}")
    print(f"Monthly Cost: ${rec['monthly_cost']:,}")
    print(f"Efficiency: {rec['gpu_utilization']}% GPU utilization")
    print(f"Cost per RPS: ${rec['cost_per_rps']}/RPS/month")
    return 0


if __name__ == "__main__":
    sys.exit(main())

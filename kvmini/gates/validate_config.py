"""Profile / configuration validator (``kvmini validate`` / stage 0 of ``bench``).

Four rule groups, as in the reference (``scripts/validate_config.py:23-138``):

1. multi-step scheduling (``num_scheduler_steps > 1``, chunked prefill, v2 block
   manager) without ``max_tokens`` — error;
2. quantisation: pre-quantised-weights warning for AWQ/GPTQ, format
   compatibility against ``model_requirements.compatible_formats`` — error; FP8 on
   ROCm: gfx950 (MI350X/MI355X) and gfx942 have native FP8 matrix cores, so the
   reference's "H100 only" warning becomes a hardware check — FP8 only warns when
   the detected/declared GPU has no FP8 MFMA (MI250X and older), and notes that
   gfx950 uses OCP e4m3 while gfx942 uses the FNUZ encoding;
3. CPU-only deployment warning;
4. resource sanity: concurrency > requests, max_tokens > 2048, concurrency <= 1,
   and the GPU-memory heuristic (fp16 weights 7b/13b/34b/70b -> 14/26/65/140 GB,
   /4 for int4 formats, x1.2 head-room) against the available GPU memory, which is
   auto-detected through amd-smi (min over visible GPUs; 288 GB on MI355X) when
   not given.
"""

from __future__ import annotations

import argparse
import sys
from typing import Any, Dict, List, Optional

FP8_NATIVE = ("mi355", "mi350", "mi325", "mi300", "mi308", "gfx950", "gfx942")
MODEL_FP16_GB = {"1b": 2, "3b": 6, "7b": 14, "8b": 16, "13b": 26, "34b": 65, "70b": 140, "405b": 810}


class ConfigValidator:
    def __init__(self) -> None:
        self.errors: List[str] = []
        self.warnings: List[str] = []

    def validate_multistep_scheduling(self, config: Dict[str, Any]) -> None:
        vf = config.get("vllm_features") or {}
        steps = vf.get("num_scheduler_steps")
        multistep = (isinstance(steps, int) and steps > 1) or bool(vf.get("enable_chunked_prefill")) or bool(vf.get("use_v2_block_manager"))
        if multistep and not config.get("max_tokens"):
            self.errors.append(
                "Multi-step scheduling detected without max_tokens. This will cause the KServe vLLM runtime to crash. "
                "Add 'max_tokens: <value>' to your profile or use the --max-tokens flag."
            )

    def validate_quantization_compatibility(self, config: Dict[str, Any]) -> None:
        vf = config.get("vllm_features") or {}
        quant = vf.get("quantization")
        kv_dtype = vf.get("kv_cache_dtype")
        gpu_name = str(config.get("_gpu_name") or (config.get("validation_hints") or {}).get("gpu") or "").lower()
        if quant == "fp8" or (isinstance(kv_dtype, str) and kv_dtype.startswith("fp8")):
            if gpu_name and not any(tag in gpu_name for tag in FP8_NATIVE):
                self.warnings.append(
                    f"FP8 requested but GPU '{gpu_name}' has no FP8 matrix cores; it will be emulated or rejected. "
                    "Suggestion: use MI300X/MI325X (gfx942) or MI350X/MI355X (gfx950). Docs: docs/FEATURES.md"
                )
            elif "gfx942" in gpu_name or "mi300" in gpu_name or "mi325" in gpu_name:
                self.warnings.append("FP8 on gfx942 uses the FNUZ e4m3 encoding; checkpoints quantised for OCP e4m3 (gfx950, H100) need re-scaling. Docs: docs/FEATURES.md")
        if quant in ("awq", "gptq"):
            self.warnings.append(
                f"{quant.upper()} quantization requires pre-quantized weights. "
                "Suggestion: use a model repo with built-in AWQ/GPTQ weights or provide quantization_param_path. Docs: docs/FEATURES.md"
            )
        compatible = (config.get("model_requirements") or {}).get("compatible_formats")
        if isinstance(compatible, str):
            compatible = [compatible]
        if quant and compatible and quant not in set(compatible):
            self.errors.append(
                f"Quantization method '{quant}' is not compatible with model formats {compatible}. "
                "Suggestion: choose one of the compatible formats or switch profile. Docs: docs/FEATURES.md"
            )

    def validate_cpu_deployment(self, config: Dict[str, Any]) -> None:
        if (config.get("vllm_features") or {}).get("device") == "cpu":
            self.warnings.append("CPU-only deployment has limited vLLM support. Many features (quantization, speculative decoding) are GPU-only. Docs: docs/FEATURES.md")

    def validate_resource_requirements(self, config: Dict[str, Any]) -> None:
        requests = config.get("requests", 0) or 0
        concurrency = config.get("concurrency", 1) or 1
        max_tokens = config.get("max_tokens", 0) or 0
        if requests > 0 and concurrency > requests:
            self.warnings.append(f"Concurrency ({concurrency}) > requests ({requests}). Some workers will be idle.")
        if max_tokens > 2048:
            self.warnings.append(f"Large max_tokens ({max_tokens}) may cause memory issues. Monitor GPU memory utilization during benchmarks.")
        gpu_mem = config.get("_gpu_memory_gb")
        hints = config.get("validation_hints") or {}
        model_hint = str(hints.get("model_size_hint") or config.get("model_size_hint") or "").lower()
        quant = (config.get("vllm_features") or {}).get("quantization")
        base = MODEL_FP16_GB.get(model_hint)
        if gpu_mem is not None and base is not None:
            if quant in ("awq", "gptq", "int4"):
                base = max(4, base // 4)
            elif quant in ("fp8", "int8"):
                base = max(4, base // 2)
            tp = int((config.get("vllm_features") or {}).get("tensor_parallel_size") or 1)
            need = int(base * 1.2 / max(1, tp))
            if need > gpu_mem:
                self.errors.append(
                    f"Estimated GPU memory insufficient: need ~{need} GiB per GPU, have {gpu_mem} GiB. "
                    "Suggestion: enable INT4 (AWQ/GPTQ) or FP8, raise tensor_parallel_size, reduce max_tokens/concurrency, or use a smaller model. Docs: docs/FEATURES.md"
                )
        min_hint = hints.get("gpu_memory_gb_min")
        if gpu_mem is not None and isinstance(min_hint, (int, float)) and gpu_mem < min_hint:
            self.warnings.append(f"Profile recommends at least {min_hint} GiB of GPU memory; detected {gpu_mem} GiB. Suggestion: pick a smaller model or a whole-GPU (SPX) partition.")
        if concurrency <= 1:
            self.warnings.append("Low concurrency (<=1) may underutilize GPU. Suggestion: try 4–16 for throughput baselines.")

    def validate_profile(self, config: Dict[str, Any]) -> bool:
        self.errors.clear()
        self.warnings.clear()
        self.validate_multistep_scheduling(config)
        self.validate_quantization_compatibility(config)
        self.validate_cpu_deployment(config)
        self.validate_resource_requirements(config)
        return not self.errors


def detect_gpu() -> Dict[str, Any]:
    """{'memory_gb': min VRAM over visible GPUs, 'name': product} through the amd-smi
    CLI (short timeout), then the amdsmi library; empty when neither answers."""
    import json as _json
    import subprocess

    try:
        res = subprocess.run(["amd-smi", "static", "--vram", "--asic", "--json"], capture_output=True, text=True, timeout=5)
        if res.returncode == 0 and res.stdout.strip():
            data = _json.loads(res.stdout)
            gpus = data if isinstance(data, list) else data.get("gpu_data") or data.get("gpus") or []
            mems, name = [], None
            for g in gpus:
                vram = (g.get("vram") or {}).get("size")
                if isinstance(vram, dict):
                    val, unit = vram.get("value"), str(vram.get("unit", "MB")).upper()
                    if isinstance(val, (int, float)):
                        mems.append(val / 1024 if unit.startswith("M") else val)
                elif isinstance(vram, (int, float)):
                    mems.append(vram / 1024)
                name = name or (g.get("asic") or {}).get("market_name")
            if mems:
                return {"memory_gb": int(min(mems)), "name": name}
    except Exception:
        pass
    try:
        from kvmini.telemetry.amdsmi_sampler import gpu_inventory

        inv = [g for g in gpu_inventory() if g.get("vram_total_bytes")]
        if inv:
            return {"memory_gb": int(min(g["vram_total_bytes"] for g in inv) / (1 << 30)), "name": inv[0].get("product")}
    except Exception:
        pass
    return {}


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(prog="kvmini validate", description="Validate benchmark configurations for known issues")
    ap.add_argument("--profile", type=str)
    ap.add_argument("--max-tokens", type=int)
    ap.add_argument("--concurrency", type=int)
    ap.add_argument("--requests", type=int)
    ap.add_argument("--vllm-args", type=str, help="Additional vLLM arguments to validate")
    ap.add_argument("--gpu-memory-gb", type=int, help="Available GPU memory (GiB) for heuristic checks")
    ap.add_argument("--gpu-name", type=str, help="GPU product (default: detected)")
    ap.add_argument("--no-detect", action="store_true", help="do not query amd-smi")
    ap.add_argument("--all", action="store_true", help="validate every config in the repository (profiles, pricing, SLO, matrix, chart values, dashboards)")
    ap.add_argument("--yaml-tree", default=None, metavar="DIR", help="parse every YAML file under DIR")
    ap.add_argument("--exclude", nargs="*", default=["charts/*/templates", "charts/*/templates/*"])
    a = ap.parse_args(argv)
    if a.all or a.yaml_tree:
        from kvmini.gates import repo_checks
        from kvmini.utils import repo_root

        if a.all:
            n, sections = repo_checks.run_all(repo_root())
        else:
            errs = repo_checks.check_yaml_tree(a.yaml_tree, a.exclude)
            n, sections = len(errs), {"yaml": errs}
        repo_checks.report(sections)
        return 1 if n else 0
    config: Dict[str, Any] = {}
    if a.profile:
        try:
            from kvmini.utils import load_yaml

            config = load_yaml(a.profile) or {}
        except Exception as e:
            print(f"Error loading profile {a.profile}: {e}", file=sys.stderr)
            return 1
    if a.max_tokens:
        config["max_tokens"] = a.max_tokens
    if a.concurrency:
        config["concurrency"] = a.concurrency
    if a.requests:
        config["requests"] = a.requests
    if a.vllm_args:
        vf = config.setdefault("vllm_features", {})
        toks = a.vllm_args.replace("=", " ").split()
        for i, t in enumerate(toks):
            if t == "--num-scheduler-steps":
                try:
                    vf["num_scheduler_steps"] = int(toks[i + 1])
                except (IndexError, ValueError):
                    vf["num_scheduler_steps"] = 2
            elif t == "--enable-chunked-prefill":
                vf["enable_chunked_prefill"] = True
            elif t == "--quantization" and i + 1 < len(toks):
                vf["quantization"] = toks[i + 1]
            elif t == "--kv-cache-dtype" and i + 1 < len(toks):
                vf["kv_cache_dtype"] = toks[i + 1]
            elif t == "--tensor-parallel-size" and i + 1 < len(toks):
                try:
                    vf["tensor_parallel_size"] = int(toks[i + 1])
                except ValueError:
                    pass
    if a.gpu_memory_gb:
        config["_gpu_memory_gb"] = a.gpu_memory_gb
    if a.gpu_name:
        config["_gpu_name"] = a.gpu_name
    if ("_gpu_memory_gb" not in config or "_gpu_name" not in config) and not a.no_detect:
        det = detect_gpu()
        if det.get("memory_gb") and "_gpu_memory_gb" not in config:
            config["_gpu_memory_gb"] = det["memory_gb"]
        if det.get("name") and "_gpu_name" not in config:
            config["_gpu_name"] = det["name"]

    v = ConfigValidator()
    v.validate_profile(config)
    if v.warnings:
        print("WARNINGS (actionable suggestions):", file=sys.stderr)
        for w in v.warnings:
            print(f"  - {w}", file=sys.stderr)
        print("", file=sys.stderr)
    if v.errors:
        print("ERRORS:", file=sys.stderr)
        for e in v.errors:
            print(f"  - {e}", file=sys.stderr)
        print("\nConfiguration validation FAILED. See suggestions above.", file=sys.stderr)
        print("Further reading:\n  - docs/FEATURES.md\n  - docs/PARTITIONS.md", file=sys.stderr)
        return 1
    print("Configuration validation PASSED.")
    if v.warnings:
        print("Note: warnings were reported above.")
    return 0


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python3
"""2048-token prompts through the engine (Llama-3-8B, random weights) per weight format: bf16,
FP8 weights with bf16 activations (KVMINI_FP8_ACT=0), FP8 weights x FP8 activations on the scaled
MFMA, MXFP4.  One process per configuration (the activation switch is read at import).  JSON lines.

    python scripts/prefill_quant_bench.py [--budget 2048] [--configs bf16,fp8-w8a16,fp8,mxfp4]
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(quant: str, budget: int, prompt: int) -> None:
    sys.path.insert(0, ROOT)
    import torch

    from kvmini.engine.engine import EngineConfig, LLMEngine
    from kvmini.engine.scheduler import SamplingParams

    eng = LLMEngine(EngineConfig(model="llama-3-8b", device="cuda:0", max_num_seqs=32, max_batched_tokens=budget, max_model_len=4096, kernels="hip", kv_memory_fraction=0.2,
                                 quantization=None if quant == "bf16" else quant))
    eng.warmup()
    times = []
    for rep in range(5):
        seq = eng.make_sequence([1] + [(7 * i + rep) % 1000 + 10 for i in range(prompt - 1)], SamplingParams(max_tokens=1, ignore_eos=True))
        eng.scheduler.add(seq)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        while eng.scheduler.has_work():
            eng.step()
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    best = min(times[1:])
    print(json.dumps({"bench": "prefill_quant", "quantization": quant, "fp8_activations": os.environ.get("KVMINI_FP8_ACT", "1") == "1", "prompt_tokens": prompt, "step_budget": budget,
                      "ms_per_prompt": round(best, 2), "prompt_tokens_per_s": round(prompt / best * 1e3)}), flush=True)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--budget", type=int, default=2048)
    ap.add_argument("--prompt", type=int, default=2048)
    ap.add_argument("--configs", default="bf16,fp8-w8a16,fp8,mxfp4")
    ap.add_argument("--child", default=None)
    a = ap.parse_args()
    if a.child:
        child(a.child, a.budget, a.prompt)
        return 0
    for cfg in a.configs.split(","):
        env = dict(os.environ)
        quant = cfg
        if cfg == "fp8-w8a16":
            quant, env["KVMINI_FP8_ACT"] = "fp8", "0"
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", quant, "--budget", str(a.budget), "--prompt", str(a.prompt)], env=env, capture_output=True, text=True, timeout=500)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            print(json.dumps({"bench": "prefill_quant", "config": cfg, "error": (r.stderr or r.stdout)[-600:]}), flush=True)
            return 1
        rec = json.loads(lines[-1])
        rec["config"] = cfg
        print(json.dumps(rec), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())

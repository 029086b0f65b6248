#!/usr/bin/env python3
"""Prompt steps are padded to a token bucket: time to the first token of single prompts of several
lengths under two bucket lists (powers of two, and with the 1.5 x steps in between)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from kvmini.engine import runner as R  # noqa: E402
from kvmini.engine.engine import EngineConfig, LLMEngine  # noqa: E402
from kvmini.engine.scheduler import SamplingParams  # noqa: E402

LISTS = {"powers_of_two": (16, 32, 48, 64, 128, 256, 512, 1024, 2048), "with_1.5x_steps": (16, 32, 48, 64, 128, 256, 384, 512, 768, 1024, 1536, 2048),
         "with_1.25x_steps": (16, 32, 48, 64, 128, 192, 256, 320, 384, 512, 640, 768, 896, 1024, 1280, 1536, 1792, 2048)}
if len(sys.argv) > 1:
    LISTS = {k: v for k, v in LISTS.items() if k in sys.argv[1:]}
for name, buckets in LISTS.items():
    R.DEFAULT_BUCKETS = buckets
    eng = LLMEngine(EngineConfig(model="llama-3-8b", device="cuda:0", max_num_seqs=4, max_batched_tokens=2048, max_model_len=4096, kernels="hip", kv_memory_fraction=0.2, enable_prefix_caching=False))
    t0 = time.perf_counter()
    eng.warmup()
    warm = time.perf_counter() - t0
    for plen in (150, 300, 600, 700, 850, 1100, 1500, 1700, 2000):
        best = 1e9
        for rep in range(4):
            ids = [1] + [(7 * i + rep) % 5000 + 10 for i in range(plen - 1)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.generate([ids], SamplingParams(max_tokens=1, ignore_eos=True))
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print(json.dumps({"buckets": name, "prompt_tokens": plen, "bucket": eng.runner.bucket_for(plen), "first_token_ms": round(best * 1e3, 2), "warmup_s": round(warm, 2), "graphs": len(eng.runner.graphs)}), flush=True)
    del eng
    torch.cuda.empty_cache()

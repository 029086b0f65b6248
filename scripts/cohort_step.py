#!/usr/bin/env python3
"""The step the closed-loop headline workload produces once per cycle: 18 prompts of 14 tokens
admitted together next to 2 decoding sequences (254 tokens: tiled-GEMM path).  Prints its
time, for rocprofv3 --kernel-trace --stats.  GPU box."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from kvmini.engine.engine import EngineConfig, LLMEngine
from kvmini.engine.scheduler import SamplingParams

eng = LLMEngine(EngineConfig(model="llama-3-8b", device="cuda:0", max_num_seqs=32, max_batched_tokens=256, max_model_len=2048, kernels="hip", kv_memory_fraction=0.2, async_scheduling=False))
eng.warmup()
sp = SamplingParams(max_tokens=200, ignore_eos=True)
for i in range(2):
    eng.scheduler.add(eng.make_sequence([1] + list(range(50 + i, 63 + i)), sp))
for _ in range(4):
    eng.step()
ts = []
for rep in range(6):
    seqs = [eng.make_sequence([1] + list(range(100 + 20 * rep + i, 113 + 20 * rep + i)), SamplingParams(max_tokens=3, ignore_eos=True)) for i in range(int(os.environ.get("COHORT", "18")))]
    for s in seqs:
        eng.scheduler.add(s)
    if rep == 5:  # the last cohort step sits between two `arange` launches: scripts/trace_between_markers.py cuts a kernel trace there
        torch.arange(0, 4099, device="cuda:0")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.step()
    if rep == 5:
        torch.arange(0, 4099, device="cuda:0")
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
    for _ in range(4):
        eng.step()
print(f"cohort step ({os.environ.get('COHORT', '18')} x 14 prompt tokens + 2 decodes):", " ".join(f"{t:.2f}" for t in ts), "ms")

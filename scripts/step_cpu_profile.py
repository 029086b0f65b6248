#!/usr/bin/env python3
"""cProfile of the engine's host-side work per decode step (GPU box)."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from kvmini.engine.engine import EngineConfig, LLMEngine
from kvmini.engine.scheduler import SamplingParams

eng = LLMEngine(EngineConfig(model="llama-3-8b", device="cuda:0", max_num_seqs=32, max_batched_tokens=256, max_model_len=2048, kernels="hip", kv_memory_fraction=0.2))
eng.warmup()
prompts = [[1] + list(range(100 + i, 113 + i)) for i in range(20)]
for rep in range(2):
    seqs = [eng.make_sequence(p, SamplingParams(max_tokens=64, ignore_eos=True)) for p in prompts]
    for s in seqs:
        eng.scheduler.add(s)
    eng.step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    n = 0
    if rep == 1:
        pr.enable()
    while eng.scheduler.has_work():
        eng.step()
        n += 1
    if rep == 1:
        pr.disable()
    torch.cuda.synchronize()
    print("ms/step", (time.perf_counter() - t0) / n * 1e3, "steps", n)
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)

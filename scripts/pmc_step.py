#!/usr/bin/env python3
"""A few eager (no HIP graph) decode steps of the engine, for rocprofv3 --pmc passes:
every kernel is an individual dispatch, so per-kernel counters can be attributed."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from kvmini.engine.engine import EngineConfig, LLMEngine
from kvmini.engine.scheduler import SamplingParams

model = sys.argv[1] if len(sys.argv) > 1 else "llama-3-8b"
eng = LLMEngine(EngineConfig(model=model, device="cuda:0", max_num_seqs=32, max_batched_tokens=64, max_model_len=512, kernels="hip", kv_memory_fraction=0.1, use_graphs=False))
prompts = [[1] + list(range(100 + i, 113 + i)) for i in range(20)]
seqs = [eng.make_sequence(p, SamplingParams(max_tokens=8, ignore_eos=True)) for p in prompts[:18]]
# two sampled requests: the threshold-search sampling kernel shows up in the counters too
seqs += [eng.make_sequence(p, SamplingParams(max_tokens=8, temperature=0.8, top_p=0.9, top_k=50, ignore_eos=True)) for p in prompts[18:]]
for s in seqs:
    eng.scheduler.add(s)
n = 0
while eng.scheduler.has_work():
    eng.step()
    n += 1
torch.cuda.synchronize()
print("steps", n)

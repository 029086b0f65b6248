#!/usr/bin/env python3
"""A 2048-token prompt through the engine (Llama-3-8B, step budget = argv[1], default 1024), four
times: the workload for `rocprofv3 --kernel-trace --stats` of the prefill path.  Prints ms per
prompt.  The last prompt sits between two `arange` launches (the only ones of the process), so a
kernel trace can be cut to exactly one prompt (scripts/trace_between_markers.py)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from kvmini.engine.engine import EngineConfig, LLMEngine
from kvmini.engine.scheduler import SamplingParams

budget = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
eng = LLMEngine(EngineConfig(model="llama-3-8b", device="cuda:0", max_num_seqs=32, max_batched_tokens=budget, max_model_len=4096, kernels="hip", kv_memory_fraction=0.2))
eng.warmup()
for rep in range(4):
    seq = eng.make_sequence([1] + [(7 * i + rep) % 1000 + 10 for i in range(2047)], SamplingParams(max_tokens=1, ignore_eos=True))
    eng.scheduler.add(seq)
    if rep == 3:
        torch.arange(0, 4099, device="cuda:0")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while eng.scheduler.has_work():
        eng.step()
    if rep == 3:
        torch.arange(0, 4099, device="cuda:0")
    torch.cuda.synchronize()
    print(f"prompt {rep}: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)

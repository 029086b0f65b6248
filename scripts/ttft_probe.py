#!/usr/bin/env python3
"""Engine-side time to first token under load, without HTTP: 19 long-running sequences keep the
engine in steady decode; probe requests (14 prompt tokens) are submitted one at a time at random
phases; for each, the time from submit() to its first token event and the number of steps the
engine launched in between.  GPU box."""
import os
import random
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from kvmini.engine.engine import EngineConfig, LLMEngine
from kvmini.engine.scheduler import SamplingParams

margin = float(sys.argv[1]) if len(sys.argv) > 1 else 0.7
eng = LLMEngine(EngineConfig(model="llama-3-8b", device="cuda:0", max_num_seqs=32, max_batched_tokens=256, max_model_len=2048, kernels="hip", kv_memory_fraction=0.2, launch_margin_ms=margin))
eng.warmup()
first = {}
lock = threading.Lock()


def on_tokens(events):
    now = time.perf_counter()
    with lock:
        for rid, toks, done, reason, seq in events:
            if rid not in first and toks:
                first[rid] = (now, eng.stats.steps)


eng.on_tokens = on_tokens
eng.start()
for i in range(19):
    eng.submit([1] + list(range(100 + i, 113 + i)), SamplingParams(max_tokens=1500, ignore_eos=True))
time.sleep(1.0)
res = []
for k in range(40):
    time.sleep(0.05 + random.random() * 0.01)
    t0, s0 = time.perf_counter(), eng.stats.steps
    rid = eng.submit([1] + list(range(300 + k, 313 + k)), SamplingParams(max_tokens=2, ignore_eos=True))
    while True:
        with lock:
            if rid in first:
                break
        time.sleep(0.0002)
    t1, s1 = first[rid]
    res.append(((t1 - t0) * 1e3, s1 - s0))
eng.stop()
ms = sorted(r[0] for r in res)
print(f"margin {margin} ms: first token after submit: p50 {ms[len(ms) // 2]:.2f} ms, min {ms[0]:.2f}, max {ms[-1]:.2f}; steps launched in between: {sorted(r[1] for r in res)}")
print(f"step ms ewma {eng.stats.step_ms_ewma:.3f}, pipelined steps {eng.pipelined_steps}")

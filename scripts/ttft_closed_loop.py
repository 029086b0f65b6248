#!/usr/bin/env python3
"""Closed-loop emulation without HTTP: 20 sequences of 64 tokens; when one finishes, a new one is
submitted DELAY ms later (the client's turnaround).  Prints the engine-side time from submit to
first token and the steps launched in between.  GPU box.  Usage: ttft_closed_loop.py [delay_ms] [margin_ms]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from kvmini.engine.engine import EngineConfig, LLMEngine
from kvmini.engine.scheduler import SamplingParams

delay = float(sys.argv[1]) * 1e-3 if len(sys.argv) > 1 else 1.5e-3
margin = float(sys.argv[2]) if len(sys.argv) > 2 else 0.7
eng = LLMEngine(EngineConfig(model="llama-3-8b", device="cuda:0", max_num_seqs=32, max_batched_tokens=256, max_model_len=2048, kernels="hip", kv_memory_fraction=0.2, launch_margin_ms=margin))
eng.warmup()
sub = {}
res = []
todo = []
lock = threading.Lock()
N_TOTAL = 400
count = [0]


def submit(k):
    t0, s0 = time.perf_counter(), eng.stats.steps
    rid = eng.submit([1] + list(range(300 + k % 500, 313 + k % 500)), SamplingParams(max_tokens=64, ignore_eos=True))
    with lock:
        sub[rid] = (t0, s0, False)


def on_tokens(events):
    now = time.perf_counter()
    with lock:
        for rid, toks, done, reason, seq in events:
            ent = sub.get(rid)
            if ent and not ent[2] and toks:
                res.append(((now - ent[0]) * 1e3, eng.stats.steps - ent[1]))
                sub[rid] = (ent[0], ent[1], True)
            if done and count[0] < N_TOTAL:
                count[0] += 1
                todo.append((now + delay, count[0]))


eng.on_tokens = on_tokens
eng.start()
for i in range(20):
    submit(i)
    time.sleep(0.011)  # spread the phases
t_end = time.time() + 60
while time.time() < t_end:
    with lock:
        due = [x for x in todo if x[0] <= time.perf_counter()]
        for x in due:
            todo.remove(x)
        finished = count[0] >= N_TOTAL and not todo and all(v[2] for v in sub.values())
    for _, k in due:
        submit(100 + k)
    if finished:
        break
    time.sleep(0.0001)
eng.stop()
ms = sorted(r[0] for r in res[20:])
steps = [r[1] for r in res[20:]]
print(f"delay {delay * 1e3:.1f} ms, margin {margin}: submit -> first token p50 {ms[len(ms) // 2]:.2f} ms (min {ms[0]:.2f}, p95 {ms[int(len(ms) * 0.95)]:.2f}); steps in between: " + ", ".join(f"{s}: {steps.count(s)}" for s in sorted(set(steps))))

#!/usr/bin/env python3
"""Host-side cost of one engine step (GPU box), one-step-at-a-time against pipelined stepping
(the next step launched before the previous one's ids are read back).  Prints ms per step of a
20-sequence decode for both, and for the synchronous mode the mean microseconds per phase:
schedule, staging up to the graph launch, wait for the GPU, commit."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from kvmini.engine.engine import EngineConfig, LLMEngine
from kvmini.engine.scheduler import SamplingParams

eng = LLMEngine(EngineConfig(model="llama-3-8b", device="cuda:0", max_num_seqs=32, max_batched_tokens=256, max_model_len=2048, kernels="hip", kv_memory_fraction=0.2))
eng.warmup()
T = {"schedule": 0.0, "stage+launch": 0.0, "gpu wait": 0.0, "commit": 0.0}
sched, runner = eng.scheduler, eng.runner
_schedule, _commit, _run = sched.schedule, eng._commit, runner._run_staged
_finalize = eng._finalize_pending
sync = torch.cuda.Event.synchronize
marks = {}


def schedule(*a, **k):
    t = time.perf_counter(); r = _schedule(*a, **k); T["schedule"] += time.perf_counter() - t; marks["exec0"] = time.perf_counter(); return r


def run_staged(*a, **k):
    r = _run(*a, **k); T["stage+launch"] += time.perf_counter() - marks["exec0"]; return r


def synchronize(self):
    t = time.perf_counter(); sync(self); T["gpu wait"] += time.perf_counter() - t


def commit(*a, **k):
    t = time.perf_counter(); r = _commit(*a, **k); T["commit"] += time.perf_counter() - t; return r


def finalize(*a, **k):
    t = time.perf_counter(); w0 = T["gpu wait"]; r = _finalize(*a, **k); T["commit"] += time.perf_counter() - t - (T["gpu wait"] - w0); return r


sched.schedule, eng._commit, runner._run_staged, eng._finalize_pending = schedule, commit, run_staged, finalize
torch.cuda.Event.synchronize = synchronize
prompts = [[1] + list(range(100 + i, 113 + i)) for i in range(20)]
for mode in (False, True):
    eng.ecfg.async_scheduling = mode
    for rep in range(2):
        for p in prompts:
            sched.add(eng.make_sequence(p, SamplingParams(max_tokens=64, ignore_eos=True)))
        eng.step()
        eng.step()
        torch.cuda.synchronize()
        for k in T:
            T[k] = 0.0
        t0 = time.perf_counter()
        n = 0
        while sched.has_work():
            eng.step()
            n += 1
        wall = time.perf_counter() - t0
    print("pipelined" if mode else "synchronous", "steps", n, "ms/step", round(wall / n * 1e3, 4))
    for k, v in T.items():
        print(f"  {k:<14} {v / n * 1e6:8.1f} us/step")
    print(f"  {'other':<14} {(wall - sum(T.values())) / n * 1e6:8.1f} us/step")

# ---- the pieces of "stage+launch" in isolation
g, _ = runner.graphs[(32, True)]
sb = runner.stages[0]
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(200):
    runner.stage_dev.copy_(sb.cpu, non_blocking=True)
h2d = (time.perf_counter() - t) / 200
torch.cuda.synchronize()
rep = []
for _ in range(30):
    t = time.perf_counter()
    g.replay()
    rep.append(time.perf_counter() - t)
    torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(200):
    sb.out_cpu[:20].copy_(runner.out_ids[:20], non_blocking=True)
d2h = (time.perf_counter() - t) / 200
torch.cuda.synchronize()
print(f"  h2d enqueue {h2d * 1e6:.1f} us, graph replay call {sorted(rep)[len(rep) // 2] * 1e6:.1f} us, d2h enqueue {d2h * 1e6:.1f} us, stage ints {sb.cpu.numel()}")

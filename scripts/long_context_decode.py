#!/usr/bin/env python3
"""Decode-step time against context length: 20 sequences with CTX-token prompts (Llama-3-8B), 24
decode steps each.  The KV cache read per step grows with the context (20 x CTX x 128 KiB), next
to the 16 GB of weights.  GPU box.  Usage: long_context_decode.py [ctx ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from kvmini.engine.engine import EngineConfig, LLMEngine
from kvmini.engine.scheduler import SamplingParams

ctxs = [int(a) for a in sys.argv[1:]] or [128, 1024, 4096, 8000]
eng = LLMEngine(EngineConfig(model="llama-3-8b", device="cuda:0", max_num_seqs=32, max_batched_tokens=2048, max_model_len=8192, kernels="hip", kv_memory_fraction=0.3, enable_prefix_caching=False))
eng.warmup()
for ctx in ctxs:
    seqs = [eng.make_sequence([1] + [(13 * i + 7 * j) % 5000 + 10 for j in range(ctx - 1)], SamplingParams(max_tokens=25, ignore_eos=True)) for i in range(20)]
    for s in seqs:
        eng.scheduler.add(s)
    t0 = time.perf_counter()
    while any(s.is_prefilling or not s.output_ids for s in seqs):
        eng.step()
    torch.cuda.synchronize()
    t_prefill = time.perf_counter() - t0
    n = 0
    t0 = time.perf_counter()
    while eng.scheduler.has_work():
        eng.step()
        n += 1
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / max(1, n)
    kv_gb = 20 * ctx * 131072 / 1e9
    print(f"ctx {ctx}: prefill of 20 x {ctx} tokens {t_prefill * 1e3:.0f} ms ({20 * ctx / t_prefill / 1e3:.1f} k tok/s); decode {dt * 1e3:.3f} ms/step, KV read {kv_gb:.2f} GB/step, (weights + KV) / time = {(16.06 + kv_gb) / dt / 1e3:.2f} TB/s", flush=True)
